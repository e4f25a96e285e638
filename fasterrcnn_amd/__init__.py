"""
fasterrcnn_amd -- MI355X (gfx950) native Faster R-CNN inference hot path.

Keeps the Python class surface of trzy/FasterRCNN's PyTorch tree
(`pytorch/FasterRCNN/models/faster_rcnn.py`: FasterRCNNModel.forward / predict) and runs every
stage as hand-written HIP kernels behind the C ABI of `include/frcnn_hip.h`
(`fasterrcnn_amd/csrc/libfrcnn_hip.so`).  There is no CPU or eager-PyTorch fallback: if the
library is missing or no gfx950 device is present the model raises.
"""
__version__ = "0.1.0"
