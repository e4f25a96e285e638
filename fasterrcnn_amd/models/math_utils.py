"""
Host-side math helper with the contract of pytorch/FasterRCNN/models/math_utils.py:13-37.

Only `intersection_over_union` lives here: it is what the (CPU, per-image) mAP bookkeeping of
statistics.py calls.  The box-delta decoders of math_utils.py:65-128 are not separate functions
in this package: the fp32 one is fused into the RPN proposal kernel (csrc/proposals.hip) and the
float64 one into the detections kernel (csrc/detect.hip).
"""
import numpy as np


def _areas(boxes):
    return (boxes[..., 2] - boxes[..., 0]) * (boxes[..., 3] - boxes[..., 1])


def intersection_over_union(boxes1, boxes2):
    """
    IoU matrix (N, M) of boxes1 (N, 4) against boxes2 (M, 4), corners (y1, x1, y2, x2).
    Pairs whose overlap rectangle has no positive extent in either axis count as disjoint; the
    denominator carries the reference's 1e-7 guard, so the quotients (and every threshold
    comparison statistics.py makes on them) are the reference's.
    """
    a = np.asarray(boxes1)[:, None, :]
    b = np.asarray(boxes2)[None, :, :]
    extent = np.minimum(a[..., 2:4], b[..., 2:4]) - np.maximum(a[..., 0:2], b[..., 0:2])      # (N, M, 2): height, width of the overlap
    overlap = np.where((extent > 0).all(axis=-1), extent[..., 0] * extent[..., 1], 0)
    union = _areas(a) + _areas(b) - overlap
    return overlap / (union + 1e-7)
