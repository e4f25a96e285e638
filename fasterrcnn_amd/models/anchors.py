"""
Anchor generation, mirroring pytorch/FasterRCNN/models/anchors.py:25-135
(`generate_rpn_map`, :137-262, is training-side ground-truth labelling and out of scope).

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
