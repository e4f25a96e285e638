"""
Anchor generation, mirroring pytorch/FasterRCNN/models/anchors.py:25-135
and :137-262 (`generate_rpn_map`, the anchor <-> ground-truth IoU matching).

`generate_anchor_maps` keeps the reference signature and return types (numpy float32 maps) but the
maps are produced by the HIP kernel `frcnn_anchors` (bit-exact with the reference's float64 numpy
arithmetic); the model itself keeps them on the device (`device_anchor_maps`).
"""
import itertools
from math import sqrt

import numpy as np
import torch as t

from .. import _native as nv


def _compute_anchor_sizes():
    """(9,2) matrix of (height, width), k = area-major x aspect-minor (anchors.py:25-41)."""
    areas = [128 * 128, 256 * 256, 512 * 512]
    x_aspects = [0.5, 1.0, 2.0]
    heights = np.array([x_aspects[j] * sqrt(areas[i] / x_aspects[j]) for (i, j) in itertools.product(range(3), range(3))])
    widths = np.array([sqrt(areas[i] / x_aspects[j]) for (i, j) in itertools.product(range(3), range(3))])
    return np.vstack([heights, widths]).T


def device_anchor_maps(image_shape, feature_map_shape, feature_pixels, device=None):
    """
    Returns (anchor_map, anchor_valid_map) as CUDA float32 tensors shaped (H, W, 36) and (H, W, 9).
    """
    assert len(image_shape) == 3
    nv.require_gpu()
    device = t.device("cuda") if device is None else device
    height, width = int(feature_map_shape[-2]), int(feature_map_shape[-1])
    image_height, image_width = int(image_shape[1]), int(image_shape[2])
    with t.cuda.device(device):
        anchor_map = t.empty((height, width, 36), dtype=t.float32, device=device)
        valid_map = t.empty((height, width, 9), dtype=t.float32, device=device)
        nv.check(nv.lib().frcnn_anchors(image_height, image_width, height, width, int(feature_pixels),
                                        nv.ptr(anchor_map), nv.ptr(valid_map), nv.stream_ptr()), "frcnn_anchors")
    return anchor_map, valid_map


def generate_anchor_maps(image_shape, feature_map_shape, feature_pixels):
    """
    Same contract as the reference: two numpy float32 maps,
      1. (height, width, num_anchors*4): (center_y, center_x, anchor_height, anchor_width) per anchor
      2. (height, width, num_anchors):   1 where the anchor lies fully inside the image else 0
    """
    anchor_map, valid_map = device_anchor_maps(image_shape, feature_map_shape, feature_pixels)
    return anchor_map.cpu().numpy(), valid_map.cpu().numpy()


def generate_rpn_map(anchor_map, anchor_valid_map, gt_boxes, object_iou_threshold=0.7, background_iou_threshold=0.3):
    """
    Same contract as anchors.py:137-262, computed by `frcnn_rpn_targets` on the device.
      anchor_map (H,W,36), anchor_valid_map (H,W,9): numpy arrays or CUDA tensors
      gt_boxes: list of datasets.training_sample.Box (corners (y1,x1,y2,x2))
    Returns (rpn_map float32 (H,W,9,6) = (trainable, object, ty, tx, th, tw),
             object anchor indices (N,3) int64 rows (y,x,k), background anchor indices (M,3)).
    """
    nv.require_gpu()
    if len(gt_boxes) < 1:
        raise ValueError("generate_rpn_map needs at least one ground-truth box (as the reference does)")
    device = anchor_map.device if isinstance(anchor_map, t.Tensor) and anchor_map.is_cuda else t.device("cuda")
    from ..runtime import to_device_map
    am = to_device_map(anchor_map, device)
    vm = to_device_map(anchor_valid_map, device)
    height, width, num_anchors = int(vm.shape[0]), int(vm.shape[1]), int(vm.shape[2])
    a = height * width * num_anchors
    gt = t.from_numpy(np.stack([np.asarray(b.corners, dtype=np.float32) for b in gt_boxes])).to(device).contiguous()
    rpn_map = t.empty((a, 6), dtype=t.float32, device=device)
    obj = t.empty((a,), dtype=t.int32, device=device)
    bg = t.empty((a,), dtype=t.int32, device=device)
    counts = t.zeros((2,), dtype=t.int32, device=device)
    ws = t.empty((max(len(gt_boxes), 1),), dtype=t.int64, device=device)
    with t.cuda.device(device):
        nv.check(nv.lib().frcnn_rpn_targets(nv.ptr(am), nv.ptr(vm), a, nv.ptr(gt), len(gt_boxes),
                                            float(object_iou_threshold), float(background_iou_threshold),
                                            nv.ptr(rpn_map), nv.ptr(obj), nv.ptr(bg), nv.ptr(counts), nv.ptr(ws),
                                            nv.stream_ptr()), "frcnn_rpn_targets")
    n_obj, n_bg = (int(v) for v in counts.cpu().tolist())

    def coords(flat):
        flat = flat.cpu().numpy().astype(np.int64)
        return np.stack([flat // (width * num_anchors), (flat // num_anchors) % width, flat % num_anchors], axis=1)

    return (rpn_map.reshape(height, width, num_anchors, 6).cpu().numpy(), coords(obj[:n_obj]), coords(bg[:n_bg]))
