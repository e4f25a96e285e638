"""
Detector stage, mirroring pytorch/FasterRCNN/models/detector.py:20-80 (losses at :83-155: csrc/train.hip
`frcnn_detector_loss`, driven by fasterrcnn_amd/training.py).  `_classifier` / `_regressor` keep their names and
initialisation; RoI pooling is csrc/roipool.hip, the heads one stacked GEMM (csrc/linear.hip)
followed by the softmax/split epilogue.
"""
import torch as t
from torch import nn

from .. import _native as nv
from .. import runtime as rt
from .rpn import pack_stack_rows
from .vgg16 import linear


class DetectorNetwork(nn.Module):
    def __init__(self, num_classes, backbone, pooling="pool", sampling_ratio=2):
        """pooling: "pool" = torchvision RoIPool 7x7 @ 1/16, what the reference uses (detector.py:27); "align" = RoIAlign with
        torchvision.ops.roi_align's semantics (aligned=False, `sampling_ratio` samples per bin and axis), beyond the reference:
        BASELINE.json's north_star names it."""
        super().__init__()
        if pooling not in nv.ROI_OPS:
            raise ValueError("pooling must be one of %s" % sorted(nv.ROI_OPS))
        if pooling == "align" and int(sampling_ratio) > 2:
            raise ValueError("sampling_ratio must be 1, 2 or <= 0 (adaptive)")
        self.pooling = pooling
        self.sampling_ratio = int(sampling_ratio)
        self._input_features = 7 * 7 * backbone.feature_map_channels
        self._num_classes = num_classes
        self._spatial_scale = 1.0 / backbone.feature_pixels
        # Define network
        self._pool_to_feature_vector = backbone.pool_to_feature_vector
        self._classifier = nn.Linear(in_features=backbone.feature_vector_size, out_features=num_classes)
        self._regressor = nn.Linear(in_features=backbone.feature_vector_size, out_features=(num_classes - 1) * 4)
        # Initialize weights (detector.py:33-36)
        self._classifier.weight.data.normal_(mean=0.0, std=0.01)
        self._classifier.bias.data.zero_()
        self._regressor.weight.data.normal_(mean=0.0, std=0.001)
        self._regressor.bias.data.zero_()
        self._packed_key = None
        self._packed = None

    def packed(self):
        params = [self._classifier.weight, self._classifier.bias, self._regressor.weight, self._regressor.bias]
        key = rt.param_key(params)
        if key != self._packed_key:
            n = 5 * self._num_classes - 4
            self._packed = pack_stack_rows(self._classifier, self._regressor, n_pad=(n + 127) // 128 * 128)
            self._packed_key = key
        return self._packed

    def roi_pool(self, feature_map, proposals):
        """RoIPool (or, pooling="align", RoIAlign) 7x7 of proposals (N,4) (y1,x1,y2,x2) over feature_map (1,C,H,W) -> (N, C, 7, 7)."""
        fm = rt.as_f32_cuda(feature_map, "feature_map")
        props = rt.as_f32_cuda(proposals, "proposals")
        c, fh, fw = int(fm.shape[1]), int(fm.shape[2]), int(fm.shape[3])
        n = int(props.shape[0])
        x = fm[0].permute(1, 2, 0).contiguous()
        out = t.empty((max(n, 1), 7, 7, c), dtype=t.float32, device=fm.device)
        if n > 0:
            cnt = t.tensor([n], dtype=t.int32, device=fm.device)
            with t.cuda.device(fm.device):
                if self.pooling == "align":
                    nv.check(nv.lib().frcnn_roi_align(nv.ptr(x), fh, fw, c, nv.ptr(props), nv.ptr(cnt), n, 7, float(self._spatial_scale),
                                                      self.sampling_ratio, 0, nv.ptr(out), nv.stream_ptr()), "frcnn_roi_align")
                else:
                    nv.check(nv.lib().frcnn_roi_pool(nv.ptr(x), fh, fw, c, nv.ptr(props), nv.ptr(cnt), n, 7,
                                                     float(self._spatial_scale), nv.ptr(out), nv.stream_ptr()), "frcnn_roi_pool")
        return out[:n].permute(0, 3, 1, 2)

    def forward(self, feature_map, proposals):
        """
        Same contract as detector.py:38-80: classes (N, num_classes) softmax, box deltas
        (N, 4*(num_classes-1)) as (ty, tx, th, tw) per class.
        """
        assert feature_map.shape[0] == 1, "Batch size must be 1"
        rois = self.roi_pool(feature_map, proposals)
        y = self._pool_to_feature_vector(rois=rois)
        n = int(y.shape[0])
        ncls = self._num_classes
        nd = (ncls - 1) * 4
        wh, bh = self.packed()
        logits = linear(y, wh, bh, ncls + nd, relu=False)
        classes = t.empty((n, ncls), dtype=t.float32, device=y.device)
        if n > 0:
            with t.cuda.device(y.device):
                nv.check(nv.lib().frcnn_softmax_rows(nv.ptr(logits), ncls + nd, nv.ptr(classes), n, ncls, nv.stream_ptr()),
                         "frcnn_softmax_rows")
        box_deltas = logits[:, ncls:].contiguous()
        return classes, box_deltas
