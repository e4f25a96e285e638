"""
Host-side math helper kept from pytorch/FasterRCNN/models/math_utils.py:13-37.

Only `intersection_over_union` lives here: it is what the (CPU, per-image) mAP bookkeeping of
statistics.py calls.  The box-delta decoders of math_utils.py:65-128 are not separate functions
in this package: the fp32 one is fused into the RPN proposal kernel (csrc/proposals.hip) and the
float64 one into the detections kernel (csrc/detect.hip).
"""
import numpy as np


def intersection_over_union(boxes1, boxes2):
    """
    IoU of every pair of boxes1 (N,4) x boxes2 (M,4), boxes as (y1, x1, y2, x2) -> (N, M).
    Same arithmetic as the reference, including the 1e-7 epsilon in the denominator.
    """
    top_left_point = np.maximum(boxes1[:, None, 0:2], boxes2[:, 0:2])
    bottom_right_point = np.minimum(boxes1[:, None, 2:4], boxes2[:, 2:4])
    well_ordered_mask = np.all(top_left_point < bottom_right_point, axis=2)
    intersection_areas = well_ordered_mask * np.prod(bottom_right_point - top_left_point, axis=2)
    areas1 = np.prod(boxes1[:, 2:4] - boxes1[:, 0:2], axis=1)
    areas2 = np.prod(boxes2[:, 2:4] - boxes2[:, 0:2], axis=1)
    union_areas = areas1[:, None] + areas2 - intersection_areas
    epsilon = 1e-7
    return intersection_areas / (union_areas + epsilon)
