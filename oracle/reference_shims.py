"""
oracle/reference_shims.py -- TEST INFRASTRUCTURE ONLY (build container only).

Makes the reference's PyTorch tree importable on a CPU-only box so oracle/make_golden.py can run
the REAL reference orchestration (models/faster_rcnn.py, rpn.py, detector.py, anchors.py,
math_utils.py, statistics.py) and capture golden vectors.  /root/reference is never copied: it is
imported in place.  What the reference needs but this image lacks is stubbed at import time:

  * torchvision (pytorch/requirements.txt:8, not installed, not vendored): a stub package whose
    ops.nms / ops.RoIPool forward to oracle/frcnn_oracle.py's restatements of torchvision's
    documented semantics (RoIPool's backward: oracle/train_oracle.py RoIPoolFunction) -- so the golden vectors pin the reference's OWN code around those two
    calls, not torchvision itself (parity unpinned there);
  * imageio (datasets/image.py:11): empty stub, never called;
  * `.cuda()` / device="cuda" (rpn.py:120-122, math_utils.py:125, detector.py:65,
    faster_rcnn.py:217-218): mapped to the CPU.
"""
import os
import sys
import tempfile
import types

import torch as t
from torch import nn

REFERENCE_ROOT = "/root/reference"


def install(oracle_module):
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("%s is not mounted (this script only runs in the build container)" % REFERENCE_ROOT)

    tv = types.ModuleType("torchvision")
    ops = types.ModuleType("torchvision.ops")
    models = types.ModuleType("torchvision.models")

    def nms(boxes, scores, iou_threshold):
        keep = oracle_module.nms(boxes.detach().numpy(), scores.detach().numpy(), iou_threshold)
        return t.from_numpy(keep)

    class RoIPool(nn.Module):
        def __init__(self, output_size, spatial_scale):
            super().__init__()
            self.output_size = output_size
            self.spatial_scale = spatial_scale

        def forward(self, input, rois):
            if input.requires_grad:       # train_step: forward + backward restated in oracle/train_oracle.py
                from oracle import train_oracle
                return train_oracle.RoIPoolFunction.apply(input, rois, self.output_size[0], self.spatial_scale)
            out = oracle_module.roi_pool(input.detach().numpy(), rois.detach().numpy(), self.output_size[0],
                                         self.spatial_scale)
            return t.from_numpy(out)

    ops.nms = nms
    ops.RoIPool = RoIPool
    models.vgg16 = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("torchvision.models stub"))
    from oracle import tv_resnet
    for arch in ("resnet50", "resnet101", "resnet152"):
        setattr(models, arch, getattr(tv_resnet, arch))
    for wname in ("ResNet50_Weights", "ResNet101_Weights", "ResNet152_Weights"):
        setattr(models, wname, types.SimpleNamespace(IMAGENET1K_V1=None))   # no network: weights come from load_state_dict
    tv.ops = ops
    tv.models = models
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.ops"] = ops
    sys.modules["torchvision.models"] = models
    sys.modules["imageio"] = types.ModuleType("imageio")

    # .cuda() -> identity, device="cuda" -> cpu
    t.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    for name in ("tensor", "empty", "zeros", "ones", "full"):
        orig = getattr(t, name)

        def wrapped(*a, __orig=orig, **k):
            if str(k.get("device", "")).startswith("cuda"):
                k["device"] = "cpu"
            return __orig(*a, **k)
        setattr(t, name, wrapped)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # the tree has no __init__.py files: namespace packages
    from pytorch.FasterRCNN.models import faster_rcnn, vgg16, resnet, anchors, math_utils   # noqa: E402
    from pytorch.FasterRCNN import statistics                                       # noqa: E402
    from pytorch.FasterRCNN.datasets import training_sample                         # noqa: E402
    return types.SimpleNamespace(faster_rcnn=faster_rcnn, vgg16=vgg16, resnet=resnet, anchors=anchors, math_utils=math_utils,
                                 statistics=statistics, training_sample=training_sample)
