"""
Image loading and preprocessing, mirroring pytorch/FasterRCNN/datasets/image.py (ChannelOrder,
PreprocessingParams :17-31, _compute_scale_factor :34-41, load_image :59-101).

JPEG/PNG decoding stays on the host (PIL; the reference uses imageio + PIL).  Everything after the
decode -- PIL's BILINEAR resize to the minimum side, the optional horizontal flip, channel
re-ordering, scaling and mean/std normalisation into the float32 (3, h, w) tensor -- runs on the
device in `frcnn_preprocess` (csrc/preprocess.hip), bit-exact with PIL's 8-bit resampler and the
reference's float32 arithmetic.
"""
import ctypes as C
from dataclasses import dataclass
from enum import Enum
from typing import List

import numpy as np
import torch as t

from .. import _native as nv


class ChannelOrder(Enum):
    RGB = "RGB"
    BGR = "BGR"


@dataclass
class PreprocessingParams:
    """Scaling is applied first, then (x - mean) / std per channel in `channel_order` order."""
    channel_order: ChannelOrder
    scaling: float
    means: List[float]
    stds: List[float]


def _compute_scale_factor(original_width, original_height, min_dimension_pixels):
    if not min_dimension_pixels:
        return 1.0
    if original_width > original_height:
        scale_factor = min_dimension_pixels / original_height
    else:
        scale_factor = min_dimension_pixels / original_width
    return scale_factor


def preprocess_image(rgb, preprocessing, min_dimension_pixels=None, horizontal_flip=False, return_resized=False):
    """
    rgb: uint8 (H, W, 3) RGB image (numpy array or CUDA tensor) as decoded by imageio/PIL.
    Returns (image_data CUDA float32 tensor (3, h, w), scale_factor, (3, H, W)); with
    return_resized also the resized uint8 (h, w, 3) CUDA tensor (what PIL's resize returns).
    Same arithmetic as image.py:92-100 + :43-57 of the reference.
    """
    nv.require_gpu()
    if isinstance(rgb, np.ndarray):
        rgb = t.from_numpy(np.ascontiguousarray(rgb))
    if rgb.dtype != t.uint8 or rgb.dim() != 3 or rgb.shape[2] != 3:
        raise ValueError("rgb must be a uint8 (H, W, 3) image")
    rgb = rgb.cuda().contiguous()
    h0, w0 = int(rgb.shape[0]), int(rgb.shape[1])
    if min_dimension_pixels is not None:
        scale_factor = _compute_scale_factor(original_width=w0, original_height=h0, min_dimension_pixels=min_dimension_pixels)
        width = int(w0 * scale_factor)            # image.py:94-95: int() truncation
        height = int(h0 * scale_factor)
    else:
        scale_factor, width, height = 1.0, w0, h0
    if preprocessing.channel_order not in (ChannelOrder.RGB, ChannelOrder.BGR):
        raise ValueError("Invalid ChannelOrder value: %s" % str(preprocessing.channel_order))
    out = t.empty((3, height, width), dtype=t.float32, device=rgb.device)
    out_u8 = t.empty((height, width, 3), dtype=t.uint8, device=rgb.device) if return_resized else None
    lib = nv.lib()
    ws_bytes = int(lib.frcnn_preprocess_workspace_bytes(h0, w0, height, width))
    ws = t.empty((ws_bytes,), dtype=t.uint8, device=rgb.device)
    means = (C.c_float * 3)(*[float(m) for m in preprocessing.means])
    stds = (C.c_float * 3)(*[float(v) for v in preprocessing.stds])
    with t.cuda.device(rgb.device):
        nv.check(lib.frcnn_preprocess(nv.ptr(rgb), h0, w0, height, width,
                                      1 if preprocessing.channel_order == ChannelOrder.BGR else 0,
                                      1 if horizontal_flip else 0, float(preprocessing.scaling), means, stds,
                                      nv.ptr(out), nv.ptr(out_u8), nv.ptr(ws), ws_bytes, nv.stream_ptr()), "frcnn_preprocess")
    if return_resized:
        return out, scale_factor, (3, h0, w0), out_u8
    return out, scale_factor, (3, h0, w0)


def load_image(url, preprocessing, min_dimension_pixels=None, horizontal_flip=False):
    """
    Same contract as the reference's load_image (image.py:59-101): returns
    (image_data np.float32 (3, h, w), PIL image (resized, for drawing), scale_factor, (3, H, W)).
    The pixels are resized and normalised on the device; use `preprocess_image` directly to keep
    the tensor on the GPU.
    """
    from PIL import Image
    with Image.open(url) as im:
        data = np.array(im.convert("RGB"))
    image_data, scale_factor, shape, resized = preprocess_image(data, preprocessing, min_dimension_pixels, horizontal_flip,
                                                               return_resized=True)
    image = Image.fromarray(resized.cpu().numpy(), mode="RGB")
    return image_data.cpu().numpy(), image, scale_factor, shape
