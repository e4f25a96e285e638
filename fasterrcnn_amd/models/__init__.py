"""
Host-side mirror of the reference's `pytorch/FasterRCNN/models` package for the inference path:
same module names, class names, constructor arguments and state_dict keys, with every tensor op
executed by libfrcnn_hip.so (see ../_native.py).
"""
from . import anchors, backbone, detector, faster_rcnn, math_utils, resnet, rpn, vgg16  # noqa: F401
