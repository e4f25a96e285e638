"""
Image preprocessing parameter types, mirroring pytorch/FasterRCNN/datasets/image.py:17-31.
Only the types the model surface needs (backbone.image_preprocessing_params); image loading and
resizing are outside the accelerated path.
"""
from dataclasses import dataclass
from enum import Enum
from typing import List


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
