"""
Backbone contract, mirroring pytorch/FasterRCNN/models/backbone.py:30-65.
"""
from ..datasets import image


class Backbone:
    """
    Backbone base class. When overriding, ensure all members and methods are defined.
    """
    def __init__(self):
        # Required properties
        self.feature_map_channels = 0   # feature map channels
        self.feature_pixels = 0         # each feature map cell corresponds to an NxN area of the image
        self.feature_vector_size = 0    # length of the pooled feature vector handed to the detector heads
        self.image_preprocessing_params = image.PreprocessingParams(
            channel_order=image.ChannelOrder.BGR, scaling=1.0, means=[103.939, 116.779, 123.680], stds=[1, 1, 1])

        # Required members
        self.feature_extractor = None       # nn.Module: image (1,C,H,W) -> feature map (1,feature_map_channels,h,w)
        self.pool_to_feature_vector = None  # nn.Module: RoIs (N,feature_map_channels,7,7) -> (N,feature_vector_size)

    def compute_feature_map_shape(self, image_shape):
        """
        (channels, height, width) of the feature extractor output for an input image shape
        (channels, height, width); only the last two dimensions of `image_shape` are used.
        """
        return image_shape[-3:]
