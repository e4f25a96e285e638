"""
Region proposal network, mirroring pytorch/FasterRCNN/models/rpn.py:29-173 (inference part;
the losses at :176-272 and their gradients are csrc/train.hip `frcnn_rpn_loss`, driven by fasterrcnn_amd/training.py).  Layer names `_rpn_conv1`,
`_rpn_class`, `_rpn_boxes` and their initialisation follow the reference; the arithmetic is
csrc/conv.hip (3x3 trunk), csrc/linear.hip (both 1x1 heads as ONE GEMM, output already NHWC)
and csrc/proposals.hip (sigmoid, decode, top-N, clip, size filter, NMS).
"""
import torch as t
from torch import nn

from .. import _native as nv
from .. import runtime as rt
from .vgg16 import conv3x3, pack_conv3x3


def pack_stack_rows(m1, m2, n_pad=128):
    """Stacks two heads' [n][k] weights (+biases) into one zero-padded [n_pad][k] matrix."""
    w1 = rt.as_f32_cuda(m1.weight.detach().reshape(m1.weight.shape[0], -1), "head weight")
    w2 = rt.as_f32_cuda(m2.weight.detach().reshape(m2.weight.shape[0], -1), "head weight")
    b1 = rt.as_f32_cuda(m1.bias.detach(), "head bias")
    b2 = rt.as_f32_cuda(m2.bias.detach(), "head bias")
    n1, n2, k = int(w1.shape[0]), int(w2.shape[0]), int(w1.shape[1])
    wo = t.empty((n_pad, k), dtype=t.float32, device=w1.device)
    bo = t.empty((n_pad,), dtype=t.float32, device=w1.device)
    with t.cuda.device(w1.device):
        nv.check(nv.lib().frcnn_pack_stack_rows(nv.ptr(w1), nv.ptr(b1), n1, nv.ptr(w2), nv.ptr(b2), n2, k, n_pad,
                                                nv.ptr(wo), nv.ptr(bo), nv.stream_ptr()), "frcnn_pack_stack_rows")
    return wo, bo


import collections

_scratch = collections.OrderedDict()
_SCRATCH_MAX = 8            # proposals-only contexts kept alive (LRU): programs that create streams dynamically must not leak 50 MB each


def scratch_context(device, h=1024, w=1024, rois=512):
    """Proposal scratch for the stage-level entry points (RegionProposalNetwork.forward, the train step): one proposals-only
    frcnn_ctx (~50 MB, frcnn_ctx_create_proposals) per (device, stream) -- a ctx is not re-entrant, and ctypes calls drop the GIL,
    so two streams / threads must never share one.  The fused model owns its own per-slot contexts.  At most _SCRATCH_MAX contexts
    are cached; the least recently used one is dropped (its frcnn_ctx is destroyed once the work enqueued on its stream has drained:
    hipFree synchronises)."""
    dev = t.device(device)
    key = (str(dev), int(t.cuda.current_stream(dev).cuda_stream))
    ctx = _scratch.get(key)
    if ctx is None or h > ctx.max_h or w > ctx.max_w:
        ctx = rt.Context(dev, max(h, 1024), max(w, 1024), 0, proposals_only=True)
        _scratch[key] = ctx
    _scratch.move_to_end(key)
    while len(_scratch) > _SCRATCH_MAX:
        _scratch.popitem(last=False)
    return ctx


class RegionProposalNetwork(nn.Module):
    def __init__(self, feature_map_channels, allow_edge_proposals=False):
        super().__init__()
        self._allow_edge_proposals = allow_edge_proposals
        num_anchors = 9
        channels = feature_map_channels
        self._rpn_conv1 = nn.Conv2d(in_channels=channels, out_channels=channels, kernel_size=(3, 3), stride=1, padding="same")
        self._rpn_class = nn.Conv2d(in_channels=channels, out_channels=num_anchors, kernel_size=(1, 1), stride=1, padding="same")
        self._rpn_boxes = nn.Conv2d(in_channels=channels, out_channels=num_anchors * 4, kernel_size=(1, 1), stride=1, padding="same")
        # Initialize weights (rpn.py:43-49)
        for m in (self._rpn_conv1, self._rpn_class, self._rpn_boxes):
            m.weight.data.normal_(mean=0.0, std=0.01)
            m.bias.data.zero_()
        self._packed_key = None
        self._packed = None
        self.math_mode = "f32"
        self.x6_trunk = False        # the 3x3 trunk as an x6 Winograd layer (csrc/wino_x6.hip) in the f32_winograd mode
        self.x3_trunk = False        # ... in the f32x3 arithmetic instead (csrc/wino_x3.hip)
        self.x3f_trunk = False       # ... as the ONE-launch f32x3 layer (csrc/wino_x3f.hip; FasterRCNNModel.alone_winograd_x3f_layers)

    def packed(self):
        params = [p for m in (self._rpn_conv1, self._rpn_class, self._rpn_boxes) for p in (m.weight, m.bias)]
        trunk_math = self.math_mode
        one_launch = self.math_mode == "f32_winograd" and self.x3f_trunk
        if one_launch:
            trunk_math = "f32_winograd_x3"
        elif self.math_mode == "f32_winograd" and self.x6_trunk:
            trunk_math = "f32_winograd_x3" if self.x3_trunk else "f32_winograd_x6"
        key = (trunk_math, one_launch) + rt.param_key(params)
        if key != self._packed_key:
            head_w, head_b = pack_stack_rows(self._rpn_class, self._rpn_boxes)
            self._packed = (pack_conv3x3(self._rpn_conv1, trunk_math, one_launch=one_launch), rt.as_f32_cuda(self._rpn_conv1.bias.detach(), "bias"),
                            head_w, head_b)
            self._packed_key = key
        return self._packed

    def packed_direct(self):
        """Fresh direct-kernel packs whatever the inference math mode is (the train step's masters)."""
        head_w, head_b = pack_stack_rows(self._rpn_class, self._rpn_boxes)
        return (pack_conv3x3(self._rpn_conv1, "f32"), rt.as_f32_cuda(self._rpn_conv1.bias.detach(), "bias").clone(), head_w, head_b)

    def forward(self, feature_map, image_shape, anchor_map, anchor_valid_map, max_proposals_pre_nms, max_proposals_post_nms):
        """
        Same contract as rpn.py:51-156.  feature_map (1, C, H, W) CUDA float32 ->
          objectness_score_map (1, H, W, 9), box_deltas_map (1, H, W, 36), proposals (N, 4) (y1, x1, y2, x2).
        """
        assert feature_map.shape[0] == 1
        fm = rt.as_f32_cuda(feature_map, "feature_map")
        dev = fm.device
        c, fh, fw = int(fm.shape[1]), int(fm.shape[2]), int(fm.shape[3])
        if c % 16 != 0 or c % 64 != 0:
            raise NotImplementedError("feature_map_channels must be a multiple of 64")
        x = fm[0].permute(1, 2, 0).contiguous()                       # NHWC plumbing
        wc, bc, wh, bh = self.packed()
        lib = nv.lib()
        ctx = scratch_context(dev, int(image_shape[1]), int(image_shape[2]))
        amap = rt.to_device_map(anchor_map, dev)
        vmap = rt.to_device_map(anchor_valid_map, dev)
        a = fh * fw * 9
        with t.cuda.device(dev):
            s = nv.stream_ptr()
            trunk = conv3x3(x, wc, bc, c, c, relu=True, pool=False, one_launch=self.math_mode == "f32_winograd" and self.x3f_trunk)
            head = t.zeros((fh * fw, 128), dtype=t.float32, device=dev)
            ws_bytes = int(lib.frcnn_linear_workspace_bytes(fh * fw, 45, c))
            ws = t.empty((max(ws_bytes, 4) // 4,), dtype=t.float32, device=dev)
            nv.check(lib.frcnn_linear(nv.ptr(trunk), c, nv.ptr(wh), nv.ptr(bh), nv.ptr(head), 128, fh * fw, 45, c, 0,
                                      nv.ptr(ws), ws_bytes, s), "frcnn_linear")
            scores = t.empty((a,), dtype=t.float32, device=dev)
            sorted_idx = t.empty((max_proposals_pre_nms,), dtype=t.int32, device=dev)
            props = t.empty((max_proposals_post_nms, 4), dtype=t.float32, device=dev)
            counts = t.zeros((4,), dtype=t.int32, device=dev)
            nv.check(lib.frcnn_rpn_proposals(ctx.handle, nv.ptr(head), 128, nv.ptr(amap),
                                             None if self._allow_edge_proposals else nv.ptr(vmap),
                                             fh, fw, int(image_shape[1]), int(image_shape[2]),
                                             int(max_proposals_pre_nms), int(max_proposals_post_nms), 0.7, 16.0,
                                             nv.ptr(scores), nv.ptr(sorted_idx), nv.ptr(props), nv.ptr(counts), s),
                     "frcnn_rpn_proposals")
        n = int(counts[2].item())
        self.last_sorted_indices = sorted_idx[: int(counts[0].item())]
        objectness_score_map = scores.reshape(1, fh, fw, 9)
        box_deltas_map = head[:, 9:45].reshape(1, fh, fw, 36).contiguous()
        return objectness_score_map, box_deltas_map, props[:n]
