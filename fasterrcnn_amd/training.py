"""
The train step of FasterRCNNModel (reference: models/faster_rcnn.py:228-362 `train_step`, the samplers at
:364-561, the losses at models/rpn.py:176-272 and models/detector.py:83-155, torch.optim.SGD built at
__main__.py:98-105) for the VGG-16 and ResNet-50/101/152 backbones -- SURVEY.md section 8 rows f2 + f3.

The reference leans on autograd; here the backward pass is written out operator by operator over the
C ABI (include/frcnn_hip.h, "Training path"): every gradient GEMM is `frcnn_gemm_tn` /
`frcnn_conv3x3_wgrad` on the exact-f32 matrix pipe, data gradients of the 3x3 layers reuse the forward
kernel with a flipped/transposed weight pack.  Design points:

  * master weights live in the PACKED layouts the kernels consume (tap-major conv weights, (7,7,C)-ordered
    fc1, stacked heads).  Weight gradients are produced directly in those layouts and SGD (momentum and
    weight decay are element-wise, so layout agnostic) updates them in place: no per-step re-packing.
    `TrainState.sync_to_parameters()` writes them back to the nn.Parameters (reference key names/layouts)
    lazily -- before state_dict(), predict() or forward();
  * what the reference trains is reproduced exactly: only parameters with "weight" in their name and
    requires_grad (VGG-16: blocks 1-2 frozen; ResNet: conv1, bn1, layer1 and every BatchNorm frozen, resnet.py:48-55,
    86, 123 -- biases are never updated).  A frozen BatchNorm is folded into its convolution: the kernels run the
    folded weight, SGD and weight decay act on the raw weight (gradient = BN scale x folded gradient);
  * host-side randomness is drawn exactly as the reference draws it (python `random.sample` for the anchor
    mini-batch, `torch.randperm` on the CPU generator for the proposal batch), so a seeded run selects the
    same samples;
  * there are two host synchronisations per step, as in the reference: the count/labels of the labelled
    proposals (the reference's `len()` / `t.where` calls) and the final read of the loss values.
"""
import random
import time

import numpy as np
import torch as t

from . import _native as nv
from . import runtime as rt
from .models import rpn as rpn_mod
from .models import vgg16

_TRAINABLE_CONVS = range(4, 13)          # block3_conv1 .. block5_conv3 (vgg16.py:49-58 freezes blocks 1-2)
_POOL_AFTER = {1, 3, 6, 9}


def _lib():
    return nv.lib()


def _ws(nbytes, device):
    return t.empty((max(int(nbytes), 4) // 4 + 1,), dtype=t.float32, device=device)


# Arithmetic of the step in progress (FasterRCNNModel.grad_math, set by train_step): "f32" = the exact-f32 matrix pipe, "bf16" = operands
# rounded to bfloat16, bf16 matrix pipe, f32 accumulation -- every gradient GEMM (csrc/gemm_tn.hip) and, round 4, the forward and
# data-gradient convolutions of the trainable ResNet blocks (csrc/conv_gather.hip: conv_gather_bf16_kernel).
_GRAD_MATH = 0

# Host clocks of the steps (tools/train_bench.py --host-clocks): a list, or None.  Per step one tuple of time.perf_counter() readings: entry,
# before / after host sync 1 (the labelled-proposal count), before / after host sync 2 (the losses) -- the two stretches between them are
# what the host needs to ENQUEUE the step's launches (the queues are deep: enqueueing does not wait for the chip).
HOST_CLOCKS = None


def gemm_tn(a, lda, b, ldb, m, n, r, out=None, ldc=None):
    """C[m][n] = sum_r A[r][m] B[r][n] (frcnn_gemm_tn_math)."""
    dev = a.device
    ldc = n if ldc is None else ldc
    c = t.empty((m, ldc), dtype=t.float32, device=dev) if out is None else out
    lib = _lib()
    wsb = int(lib.frcnn_gemm_tn_workspace_bytes(m, n, r))
    ws = _ws(wsb, dev) if wsb else None
    nv.check(lib.frcnn_gemm_tn_math(nv.ptr(a), lda, nv.ptr(b), ldb, nv.ptr(c), ldc, m, n, r, _GRAD_MATH, nv.ptr(ws), wsb,
                                    nv.stream_ptr()), "frcnn_gemm_tn_math")
    return c


def transpose(x, rows, cols, ldi):
    """[rows][ldi] -> [cols][round_up(rows, 4)] zero padded (frcnn_transpose)."""
    ldo = (rows + 3) // 4 * 4
    y = t.empty((cols, ldo), dtype=t.float32, device=x.device)
    nv.check(_lib().frcnn_transpose(nv.ptr(x), ldi, nv.ptr(y), ldo, rows, cols, nv.stream_ptr()), "frcnn_transpose")
    return y, ldo


def relu_backward(dy, y):
    nv.check(_lib().frcnn_relu_backward(nv.ptr(dy), nv.ptr(y), dy.numel(), nv.stream_ptr()), "frcnn_relu_backward")


def conv3x3_wgrad(x_hwc, dz_hwc, cin, cout):
    h, w = int(x_hwc.shape[0]), int(x_hwc.shape[1])
    lib = _lib()
    dwp = t.empty((9, cout, cin), dtype=t.float32, device=x_hwc.device)
    wsb = int(lib.frcnn_conv3x3_wgrad_workspace_bytes(h, w, cin, cout))
    ws = _ws(wsb, x_hwc.device) if wsb else None
    nv.check(lib.frcnn_conv3x3_wgrad_math(nv.ptr(x_hwc), nv.ptr(dz_hwc), nv.ptr(dwp), h, w, cin, cout, _GRAD_MATH, nv.ptr(ws), wsb,
                                          nv.stream_ptr()), "frcnn_conv3x3_wgrad_math")
    return dwp


def winograd_bank(wp, cout, cin, data_gradient=False, fused=True):
    """Direct pack [9][cout][cin] -> Winograd F(2x2,3x3) filter bank of the layer, or (data_gradient) of the convolution that maps
    the output gradient to the input gradient (rotated, channel-transposed filter).  fused: the one-launch kernel's flat layout
    (frcnn_pack_conv3x3_winograd_fused_taps), else round 1's [16][cout][cin] / [16][cin][cout] of the three-launch form.
    Rebuilt from the master weights at every use: ten 16 MB writes per step at most."""
    if fused:
        u = t.empty((16 * cin * cout,), dtype=t.float32, device=wp.device)
        nv.check(_lib().frcnn_pack_conv3x3_winograd_fused_taps(nv.ptr(wp), nv.ptr(u), cout, cin, 1 if data_gradient else 0,
                                                               nv.stream_ptr()), "frcnn_pack_conv3x3_winograd_fused_taps")
        return u
    u = t.empty((16, cin, cout) if data_gradient else (16, cout, cin), dtype=t.float32, device=wp.device)
    nv.check(_lib().frcnn_pack_conv3x3_winograd_taps(nv.ptr(wp), nv.ptr(u), cout, cin, 1 if data_gradient else 0, nv.stream_ptr()),
             "frcnn_pack_conv3x3_winograd_taps")
    return u


def conv3x3_forward(x_hwc, wp, b, cin, cout, winograd):
    """y = relu(conv3x3(x) + b) without pooling, on the direct kernel or (f32_winograd mode) as a one-launch Winograd layer."""
    if winograd and nv.uses_winograd_fused(cin, cout):
        wp = winograd_bank(wp, cout, cin)
    return vgg16.conv3x3(x_hwc, wp, b, cin, cout, relu=True, pool=False)


def conv3x3_dgrad(dz_hwc, wp, cin, cout, zero_bias, winograd=False):
    """Gradient with respect to the input of y = conv3x3(x, wp): a 3x3 conv of dz (cout channels) to cin channels."""
    if winograd and nv.uses_winograd_fused(cout, cin):
        return vgg16.conv3x3(dz_hwc, winograd_bank(wp, cout, cin, data_gradient=True), zero_bias, cout, cin, relu=False, pool=False)
    wd = t.empty((9, cin, cout), dtype=t.float32, device=dz_hwc.device)
    nv.check(_lib().frcnn_pack_conv3x3_dgrad(nv.ptr(wp), nv.ptr(wd), cout, cin, nv.stream_ptr()), "frcnn_pack_conv3x3_dgrad")
    return vgg16.conv3x3(dz_hwc, wd, zero_bias, cout, cin, relu=False, pool=False)


def maxpool2x2(x_hwc):
    h, w, c = (int(v) for v in x_hwc.shape)
    y = t.empty((h // 2, w // 2, c), dtype=t.float32, device=x_hwc.device)
    nv.check(_lib().frcnn_maxpool2x2_nhwc(nv.ptr(x_hwc), nv.ptr(y), h, w, c, nv.stream_ptr()), "frcnn_maxpool2x2_nhwc")
    return y


def maxpool2x2_backward(x_hwc, dy_hwc):
    h, w, c = (int(v) for v in x_hwc.shape)
    dx = t.empty_like(x_hwc)
    nv.check(_lib().frcnn_maxpool2x2_backward(nv.ptr(x_hwc), nv.ptr(dy_hwc), nv.ptr(dx), h, w, c, nv.stream_ptr()),
             "frcnn_maxpool2x2_backward")
    return dx


def sgd_hyper_parameters(optimizer):
    """(lr, momentum, weight_decay) of a torch.optim.SGD built as __main__.py:98-105 does, or of optim.SGD below."""
    groups = getattr(optimizer, "param_groups", None)
    if not groups:
        raise TypeError("optimizer must expose param_groups (torch.optim.SGD or fasterrcnn_amd.training.SGD)")
    vals = {(float(g["lr"]), float(g.get("momentum", 0.0)), float(g.get("weight_decay", 0.0))) for g in groups}
    if len(vals) != 1:
        raise NotImplementedError("per-parameter-group hyper-parameters are not supported: %s" % sorted(vals))
    for g in groups:
        if g.get("dampening", 0) or g.get("nesterov", False) or g.get("maximize", False):
            raise NotImplementedError("only plain SGD with momentum and weight decay (reference __main__.py:105)")
    return vals.pop()


class SGD:
    """Hyper-parameter holder with torch.optim.SGD's `param_groups` shape; the state lives in TrainState."""
    def __init__(self, lr, momentum=0.9, weight_decay=5e-4):
        self.param_groups = [{"lr": lr, "momentum": momentum, "weight_decay": weight_decay}]

    def zero_grad(self):
        pass


def create_optimizer(model, learning_rate=1e-3, momentum=0.9, weight_decay=5e-4):
    """Mirror of __main__.py:98-105 create_optimizer (defaults = the reference's CLI defaults)."""
    return SGD(learning_rate, momentum, weight_decay)


def conv_wgrad(x, dz, n, h, w, cin, cout, k, stride, pad):
    """Weight-pack gradient [k*k][cout][cin] of a general NHWC convolution (frcnn_conv_wgrad)."""
    lib = _lib()
    dwp = t.empty((k * k, cout, cin), dtype=t.float32, device=x.device)
    wsb = int(lib.frcnn_conv_wgrad_workspace_bytes(n, h, w, cin, cout, k, stride, pad))
    ws = _ws(wsb, x.device) if wsb else None
    nv.check(lib.frcnn_conv_wgrad_math(nv.ptr(x), nv.ptr(dz), nv.ptr(dwp), n, h, w, cin, cout, k, stride, pad, _GRAD_MATH,
                                       nv.ptr(ws), wsb, nv.stream_ptr()), "frcnn_conv_wgrad_math")
    return dwp


def conv_dgrad(dz, wf, residual, n, h, w, cin, cout, k, stride, pad):
    """Input gradient [n][h][w][cin] (+ residual) of a general NHWC convolution with (folded) weight pack wf."""
    lib = _lib()
    wd = t.empty((k * k, cin, cout), dtype=t.float32, device=dz.device)
    nv.check(lib.frcnn_pack_conv_dgrad(nv.ptr(wf), nv.ptr(wd), k * k, cout, cin, nv.stream_ptr()), "frcnn_pack_conv_dgrad")
    dx = t.empty((n, h, w, cin), dtype=t.float32, device=dz.device)
    wsb = int(lib.frcnn_conv_dgrad_workspace_bytes(n, h, w, cin, cout, k, stride, pad))
    ws = _ws(wsb, dz.device) if wsb else None
    nv.check(lib.frcnn_conv_dgrad_math(nv.ptr(dz), nv.ptr(wd), nv.ptr(residual), nv.ptr(dx), n, h, w, cin, cout, k, stride, pad,
                                       _GRAD_MATH, nv.ptr(ws), wsb, nv.stream_ptr()), "frcnn_conv_dgrad_math")
    return dx


class TrainState:
    """
    Packed master weights + momentum buffers of one FasterRCNNModel, and the backbone-specific halves of the step:
    `features_forward/backward` (stage 1) and `head_forward/backward` (RoI features -> feature vector).  The RPN and
    the detector heads are common: `rpn_conv`, `rpn_head`, `head` (+ biases).
    """
    C = 0          # feature-map channels
    V = 0          # feature-vector size

    def __init__(self, model):
        self.model = model
        if model.math_mode not in ("f32", "f32_winograd"):
            raise NotImplementedError("training runs in the f32 or f32_winograd math mode")
        # forward and data-gradient 3x3 convolutions of the wide VGG-16 / RPN layers as Winograd layers, weight gradients always direct
        self.winograd = model.math_mode == "f32_winograd"
        rp = model._stage2_region_proposal_network
        dn = model._stage3_detector_network
        # masters in the direct kernels' layout, whatever layout the inference mode packs
        self.rpn_conv, self.rpn_conv_b, self.rpn_head, self.rpn_head_b = rp.packed_direct()
        hw, hb = dn.packed()
        self.head, self.head_b = hw.clone(), hb.clone()
        self.device = self.head.device
        self.zero_bias = t.zeros((2048,), dtype=t.float32, device=self.device)
        self.momentum = {}
        self.steps = 0
        self.dirty = False
        self._param_key = rt.param_key(list(model.parameters()))

    def parameters_changed(self):
        """True when any nn.Parameter was replaced or written (version bump) since the masters were cloned / last synced."""
        return rt.param_key(list(self.model.parameters())) != self._param_key

    def trainable(self):
        """name -> packed master tensor, for everything SGD updates."""
        raise NotImplementedError

    def after_update(self):
        """Hook: derived packs (BN-folded weights) are rebuilt from the masters after SGD."""

    def folded_convs(self):
        """name -> _TrainConv for the masters whose folded pack is rebuilt by the update itself (frcnn_sgd_step_fold)."""
        return {}

    def apply_sgd(self, grads, lr, momentum, weight_decay):
        lib = _lib()
        folded = self.folded_convs()
        for name, w in self.trainable().items():
            g = grads[name]
            assert g.shape == w.shape and g.is_contiguous() and w.is_contiguous(), name
            buf = None
            first = 1
            if momentum != 0.0:
                buf = self.momentum.get(name)
                first = 0
                if buf is None:
                    buf = t.empty_like(w)
                    self.momentum[name] = buf
                    first = 1
            c = folded.get(name)
            if c is not None:
                nv.check(lib.frcnn_sgd_step_fold(nv.ptr(w), nv.ptr(g), nv.ptr(buf), w.numel(), lr, momentum, weight_decay, first,
                                                 nv.ptr(c.scale), nv.ptr(c.folded), c.cout, c.cin, nv.stream_ptr()), "frcnn_sgd_step_fold")
            else:
                nv.check(lib.frcnn_sgd_step(nv.ptr(w), nv.ptr(g), nv.ptr(buf), w.numel(), lr, momentum, weight_decay, first,
                                            nv.stream_ptr()), "frcnn_sgd_step")
        self.after_update()
        self.steps += 1
        self.dirty = True

    @t.no_grad()
    def sync_to_parameters(self):
        """Writes the packed masters back into the nn.Parameters (reference layouts and key names)."""
        if not self.dirty:
            return
        m = self.model
        if self.parameters_changed():
            raise RuntimeError("parameters were modified while weights trained by train_step were still pending in the packed "
                               "masters; the write-back would silently overwrite them")
        rp = m._stage2_region_proposal_network
        c = int(rp._rpn_conv1.weight.shape[0])
        rp._rpn_conv1.weight.copy_(self.rpn_conv.permute(1, 2, 0).reshape(c, c, 3, 3))
        rp._rpn_class.weight.copy_(self.rpn_head[0:9].reshape(9, c, 1, 1))
        rp._rpn_boxes.weight.copy_(self.rpn_head[9:45].reshape(36, c, 1, 1))
        dn = m._stage3_detector_network
        ncls = m._num_classes
        dn._classifier.weight.copy_(self.head[0:ncls])
        dn._regressor.weight.copy_(self.head[ncls:ncls + 4 * (ncls - 1)])
        self._sync_backbone()
        self.dirty = False
        self._param_key = rt.param_key(list(m.parameters()))

    def _sync_backbone(self):
        raise NotImplementedError


class VGG16TrainState(TrainState):
    """VGG-16: blocks 1-2 frozen (vgg16.py:49-58); conv3_1..conv5_3, fc1, fc2 train."""
    C, V = 512, 4096

    def __init__(self, model):
        super().__init__(model)
        fe = model._stage1_feature_extractor
        pv = model._stage3_detector_network._pool_to_feature_vector
        if pv._dropout1.p > 0 or pv._dropout2.p > 0:
            raise NotImplementedError("dropout > 0 is not implemented in the train step (reference default: 0.0)")
        # clones: the inference-side packed caches are rebuilt from the parameters, these are the training masters
        self.conv = fe.packed_direct()
        self._frozen_banks = {}                # one-launch Winograd banks of the frozen layers (their weights never change here)
        w1p, b1, w2, b2 = pv.packed_direct()      # float32 masters whatever arithmetic inference uses for fc1 / fc2
        self.fc1, self.fc1_b, self.fc2, self.fc2_b = w1p, b1.clone(), w2.clone(), b2.clone()

    def trainable(self):
        out = {"conv%d" % i: self.conv[i][0] for i in _TRAINABLE_CONVS}
        out.update(rpn_conv=self.rpn_conv, rpn_head=self.rpn_head, fc1=self.fc1, fc2=self.fc2, head=self.head)
        return out

    def _sync_backbone(self):
        m = self.model
        fe = m._stage1_feature_extractor
        for i in _TRAINABLE_CONVS:
            conv = fe.convs()[i]
            co, ci = int(conv.weight.shape[0]), int(conv.weight.shape[1])
            conv.weight.copy_(self.conv[i][0].permute(1, 2, 0).reshape(co, ci, 3, 3))
        pv = m._stage3_detector_network._pool_to_feature_vector
        pv._fc1.weight.copy_(self.fc1.reshape(4096, 49, 512).permute(0, 2, 1).reshape(4096, 512 * 49))
        pv._fc2.weight.copy_(self.fc2)

    # ---- stage 1 (vgg16.py:76-96) -------------------------------------------------------------------------
    def features_forward(self, image, inject=None):
        """inject (parity tests only): {"conv<i>": HWC tensor} replaces the post-ReLU output of trainable conv i by the given
        activations (an oracle's), "conv4_in" the input of conv3_1: the backward then runs on prescribed ReLU / max-pool masks."""
        lib = _lib()
        H, W = int(image.shape[2]), int(image.shape[3])
        x_in, y_out = {}, {}
        inject = inject or {}
        cur = t.empty((H, W, 64), dtype=t.float32, device=self.device)
        nv.check(lib.frcnn_conv3x3_c3(nv.ptr(image), nv.ptr(self.conv[0][0]), nv.ptr(self.conv[0][1]), nv.ptr(cur), H, W, 64,
                                      nv.RELU, nv.stream_ptr()), "frcnn_conv3x3_c3")
        for i in range(1, 13):
            _, cin, cout, pool = vgg16._LAYERS[i]
            wp, b = self.conv[i]
            if i in _TRAINABLE_CONVS:
                if ("conv%d_in" % i) in inject:
                    cur = inject["conv%d_in" % i]
                x_in[i] = cur
                y = conv3x3_forward(cur, wp, b, cin, cout, self.winograd)
                y = inject.get("conv%d" % i, y)
                y_out[i] = y
                cur = maxpool2x2(y) if pool else y
            else:
                # frozen (vgg16.py:49-58): pool fused; in the f32_winograd mode the same one-launch Winograd layer as inference,
                # its bank built once (conv1_2, conv2_1, conv2_2: 0.91 ms of direct convolutions -> 0.47 ms)
                if self.winograd and nv.uses_winograd_fused(cin, cout):
                    if i not in self._frozen_banks:
                        self._frozen_banks[i] = winograd_bank(wp, cout, cin)
                    wp = self._frozen_banks[i]
                cur = vgg16.conv3x3(cur, wp, b, cin, cout, relu=True, pool=pool)
        return cur, (x_in, y_out)

    def features_backward(self, g, saved, grads):
        """autograd of vgg16.py:84-96; `g` = gradient with respect to the (post-ReLU) feature map."""
        x_in, y_out = saved
        side = _SideGrads(grads, self.device)          # the weight gradients on the second stream, under the data-gradient chain
        for i in range(12, 3, -1):
            _, cin, cout, _ = vgg16._LAYERS[i]
            side.flush(keep=1)
            relu_backward(g, y_out[i])
            side.run("conv%d" % i, lambda x=x_in[i], dz=g, ci=cin, co=cout: conv3x3_wgrad(x, dz, ci, co), (x_in[i], g))
            if i > 4:
                gx = conv3x3_dgrad(g, self.conv[i][0], cin, cout, self.zero_bias, self.winograd)
                g = maxpool2x2_backward(y_out[i - 1], gx) if (i - 1) in _POOL_AFTER else gx
        side.flush()

    # ---- RoI features -> feature vector (vgg16.py:129-133) ------------------------------------------------
    def head_forward(self, roi_out, inject=None):
        inject = inject or {}
        h1 = inject.get("fc1", None)
        if h1 is None:
            h1 = vgg16.linear(roi_out, self.fc1, self.fc1_b, 4096, relu=True)
        h2 = inject.get("fc2", None)
        if h2 is None:
            h2 = vgg16.linear(h1, self.fc2, self.fc2_b, 4096, relu=True)
        return h2, (roi_out, h1, h2)

    def head_backward(self, dh2, saved, grads, detail=None):
        roi_out, h1, h2 = saved
        S = int(h2.shape[0])
        if detail is not None:
            detail["dh2"] = dh2.clone()
        side = _SideGrads(grads, self.device)          # fc2's and fc1's weight gradients on the second stream, under the chain dh2 -> dh1 -> d roi
        relu_backward(dh2, h2)
        side.run("fc2", lambda: gemm_tn(dh2, 4096, h1, 4096, 4096, 4096, S), (dh2, h1))
        dh2_t, sp = transpose(dh2, S, 4096, 4096)
        dh1 = gemm_tn(dh2_t, sp, self.fc2, 4096, S, 4096, 4096)
        if detail is not None:
            detail["dh1"] = dh1.clone()
            detail.update(h1=h1, h2=h2)
        relu_backward(dh1, h1)
        side.run("fc1", lambda: gemm_tn(dh1, 4096, roi_out, 49 * 512, 4096, 49 * 512, S), (dh1, roi_out))
        dh1_t, sp = transpose(dh1, S, 4096, 4096)
        droi = gemm_tn(dh1_t, sp, self.fc1, 49 * 512, S, 49 * 512, 4096)
        side.flush()       # (before "rpn_head": `grads` fills in the one-stream order, also on a rank without a proposal batch)
        return droi

    def zero_head_grads(self, grads):
        for name in ("fc2", "fc1"):
            grads[name] = t.zeros_like(self.trainable()[name])


_wgrad_streams = {}


def _wgrad_stream(device):
    """The process's second stream of a device for the weight-gradient GEMMs of the train step (one, shared by every model)."""
    key = str(t.device(device))
    st = _wgrad_streams.get(key)
    if st is None:
        st = t.cuda.Stream(device=t.device(device))
        _wgrad_streams[key] = st
    return st


class _SideGrads:
    """
    Weight gradients of the trainable bottlenecks on a SECOND stream (round 6).  The backward of a block is a chain -- ReLU mask, data
    gradient, ReLU mask, data gradient ... -- whose kernels fill 35-50 % of the chip (`tools/cu_time_model.py report DIR N` over a ResNet-101
    step: 10.9 ms of kernel time, 4.4 ms of CU time), and the weight gradient of each convolution hangs OFF that chain: nothing of the
    backward reads it.  `frcnn_bottleneck_backward` (ABI 16: one C call per block instead of ~25 -- the step was bound by the host's enqueue
    rate, 8.6 of its 11.4 ms) enqueues them on the second stream behind an event of the main stream each, so they run under the chain's
    next kernels; `add()` takes a block's gradients with the event that follows them on the second stream, `flush(keep)` hands them to
    `grads` in their production order -- the main stream waits for the block's event first, so whoever reads `grads[name]` next (the
    data-parallel exchange, SGD) is ordered behind them.  `keep` = how many of the most recent blocks stay pending: a block flushes all but
    the previous one, whose kernels have had a block's time to finish.  The values are the one-stream step's bit for bit (same kernels, same
    operands).  FRCNN_TRAIN_WGRAD_STREAM=0: one stream.
    """
    def __init__(self, grads, device):
        import os
        self.grads, self.device, self.pending = grads, t.device(device), []
        self.enabled = os.environ.get("FRCNN_TRAIN_WGRAD_STREAM", "1") != "0"

    def stream(self):
        return _wgrad_stream(self.device) if self.enabled else None

    def add(self, named_grads, scratch):
        """named_grads: [(name, tensor)] in production order, just enqueued (on the second stream when enabled); scratch: main-stream
        tensors the second stream still reads -- kept referenced until the main stream has waited for the block's event, so the allocator
        cannot hand their memory to a kernel that runs before the second stream is done with it (tensor.record_stream would say the same
        to the allocator at ~4x the host time: measured, 6.2 against 4.6 ms to enqueue a ResNet-101 backward)."""
        done = None
        if self.enabled:
            done = t.cuda.Event()
            done.record(_wgrad_stream(self.device))
        self.pending.append((named_grads, done, scratch if self.enabled else None))

    def run(self, name, fn, reads):
        """One gradient from Python: grads[name] = fn(), with fn's launches on the second stream behind an event of the main stream
        (VGG-16's nine convolution and two fc weight gradients: a dozen per step, so the ~35 us of host time this form costs each do not
        matter there).  `reads`: the main-stream tensors fn reads."""
        if not self.enabled:
            self.grads[name] = fn()
            return
        main, side = t.cuda.current_stream(self.device), _wgrad_stream(self.device)
        ready = t.cuda.Event()
        ready.record(main)
        side.wait_event(ready)
        with t.cuda.stream(side):
            g = fn()
        self.add([(name, g)], list(reads))

    def flush(self, keep=0):
        main = t.cuda.current_stream(self.device)
        while len(self.pending) > keep:
            named_grads, done, _scratch = self.pending.pop(0)
            if done is not None:
                main.wait_event(done)
            for name, g in named_grads:
                self.grads[name] = g


class _TrainConv:
    """One conv + frozen BatchNorm of a trainable Bottleneck: raw master pack, BN scale/shift, folded pack."""
    def __init__(self, conv, bn):
        lib = _lib()
        w = rt.as_f32_cuda(conv.weight.detach(), "conv weight")
        self.cout, self.cin, self.k = int(w.shape[0]), int(w.shape[1]), int(w.shape[2])
        self.stride, self.pad = int(conv.stride[0]), int(conv.padding[0])
        self.conv = conv
        # [tap][co][ci]; clone: for a 1x1 conv the permuted view IS the parameter's storage
        self.raw = w.permute(2, 3, 0, 1).reshape(self.k * self.k, self.cout, self.cin).clone(memory_format=t.contiguous_format)
        self.scale = t.empty((self.cout,), dtype=t.float32, device=w.device)
        self.shift = t.empty((self.cout,), dtype=t.float32, device=w.device)
        args = [rt.as_f32_cuda(x.detach(), "bn tensor") for x in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]
        nv.check(lib.frcnn_bn_scale_shift(nv.ptr(args[0]), nv.ptr(args[1]), nv.ptr(args[2]), nv.ptr(args[3]), float(bn.eps),
                                          self.cout, nv.ptr(self.scale), nv.ptr(self.shift), nv.stream_ptr()), "frcnn_bn_scale_shift")
        self.folded = t.empty_like(self.raw)
        self.refold()
        # frcnn_train_conv of frcnn_bottleneck_backward (folded / scale are written in place by every update: the pointers stay)
        self.cstruct = nv.TrainConv(self.folded.data_ptr(), self.scale.data_ptr(), None, None, self.cin, self.cout, self.k, self.stride, self.pad, 0)

    def refold(self):
        nv.check(_lib().frcnn_scale_rows(nv.ptr(self.raw), nv.ptr(self.scale), nv.ptr(self.folded), self.k * self.k, self.cout,
                                         self.cin, nv.stream_ptr()), "frcnn_scale_rows")

    def forward(self, x, n, h, w, relu, residual=None):
        from .models import resnet
        return resnet.conv_nhwc(x, self.folded, self.shift, n, h, w, self.cin, self.cout, self.k, self.stride, self.pad, relu,
                                residual=residual, math=_GRAD_MATH)

    def wgrad(self, x, dz, n, h, w):
        """Gradient with respect to the RAW weight: the folded weight's gradient times the BN scale of its output channel."""
        g = conv_wgrad(x, dz, n, h, w, self.cin, self.cout, self.k, self.stride, self.pad)
        nv.check(_lib().frcnn_scale_rows(nv.ptr(g), nv.ptr(self.scale), nv.ptr(g), self.k * self.k, self.cout, self.cin,
                                         nv.stream_ptr()), "frcnn_scale_rows")
        return g

    def dgrad(self, dz, residual, n, h, w):
        return conv_dgrad(dz, self.folded, residual, n, h, w, self.cin, self.cout, self.k, self.stride, self.pad)

    def sync(self):
        self.conv.weight.copy_(self.raw.permute(1, 2, 0).reshape(self.cout, self.cin, self.k, self.k))


class _TrainBlock:
    """A trainable Bottleneck (torchvision v1.5): out = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + identity)."""
    def __init__(self, block, name):
        self.name = name
        self.c1 = _TrainConv(block.conv1, block.bn1)
        self.c2 = _TrainConv(block.conv2, block.bn2)
        self.c3 = _TrainConv(block.conv3, block.bn3)
        self.cd = _TrainConv(block.downsample[0], block.downsample[1]) if block.downsample is not None else None

    def convs(self):
        out = {"conv1": self.c1, "conv2": self.c2, "conv3": self.c3}
        if self.cd is not None:
            out["downsample"] = self.cd
        return out

    def forward(self, x, n, h, w):
        t1, _, _ = self.c1.forward(x, n, h, w, True)
        t2, ho, wo = self.c2.forward(t1, n, h, w, True)
        identity = x
        if self.cd is not None:
            identity, _, _ = self.cd.forward(x, n, h, w, False)
        out, _, _ = self.c3.forward(t2, n, ho, wo, True, residual=identity)
        return out, ho, wo, (x, t1, t2, out, n, h, w, ho, wo)

    def backward(self, g, saved, side, need_dx):
        """`g` = gradient with respect to the block output (consumed); returns the gradient with respect to x or None.
        `side`: the step's _SideGrads (or a plain gradient dict: a block's backward on its own).  ONE call, frcnn_bottleneck_backward: the
        ReLU masks and data gradients on the current stream, the four weight gradients -- in the order conv3, conv2, conv1, downsample --
        on the second one."""
        x, t1, t2, out, n, h, w, ho, wo = saved
        own = not isinstance(side, _SideGrads)
        if own:
            side = _SideGrads(side, g.device)
        side.flush(keep=1)                                           # (the blocks before the previous one: their kernels are long done)
        dev, lib = g.device, _lib()
        convs = [("conv3", self.c3), ("conv2", self.c2), ("conv1", self.c1)] + ([("downsample", self.cd)] if self.cd is not None else [])
        named, keepalive = [], []
        for cname, c in convs:
            gw = t.empty((c.k * c.k, c.cout, c.cin), dtype=t.float32, device=dev)
            wd = t.empty((c.k * c.k, c.cin, c.cout), dtype=t.float32, device=dev)
            c.cstruct.grad, c.cstruct.wd = gw.data_ptr(), wd.data_ptr()
            named.append((self.name + "." + cname, gw))
            keepalive.append(wd)
        width, cin = self.c1.cout, self.c1.cin
        d_t2 = t.empty((n, ho, wo, width), dtype=t.float32, device=dev)
        d_t1 = t.empty((n, h, w, width), dtype=t.float32, device=dev)
        dx = t.empty((n, h, w, cin), dtype=t.float32, device=dev) if need_dx else None
        dx_id = t.empty((n, h, w, cin), dtype=t.float32, device=dev) if (need_dx and self.cd is not None) else None
        C_ = nv.C
        pcd = C_.byref(self.cd.cstruct) if self.cd is not None else None
        key = (n, h, w, ho, wo)
        sizes = getattr(self, "_bw_ws", {}).get(key)
        if sizes is None:
            mb, sb = C_.c_size_t(0), C_.c_size_t(0)
            nv.check(lib.frcnn_bottleneck_backward_workspace_bytes(C_.byref(self.c1.cstruct), C_.byref(self.c2.cstruct), C_.byref(self.c3.cstruct), pcd,
                                                                   n, h, w, ho, wo, C_.byref(mb), C_.byref(sb)), "frcnn_bottleneck_backward_workspace_bytes")
            sizes = (int(mb.value), int(sb.value))
            self._bw_ws = {key: sizes}
        ws_main = _ws(sizes[0], dev) if sizes[0] else None
        ws_side = _ws(sizes[1], dev) if sizes[1] else None
        side_stream = side.stream()
        nv.check(lib.frcnn_bottleneck_backward(C_.byref(self.c1.cstruct), C_.byref(self.c2.cstruct), C_.byref(self.c3.cstruct), pcd,
                                               nv.ptr(x), nv.ptr(t1), nv.ptr(t2), nv.ptr(out), nv.ptr(g), nv.ptr(d_t2), nv.ptr(d_t1), nv.ptr(dx_id),
                                               nv.ptr(dx), n, h, w, ho, wo, _GRAD_MATH, nv.ptr(ws_main), sizes[0], nv.ptr(ws_side), sizes[1],
                                               nv.stream_ptr(), side_stream.cuda_stream if side_stream is not None else None),
                 "frcnn_bottleneck_backward")
        # what the second stream still reads or writes when this returns: the block's gradient, the two intermediate gradients, its workspace
        # (x, t1, t2 stay referenced by the step until after the last flush)
        side.add(named, [g, d_t2, d_t1] + ([ws_side] if ws_side is not None else []))
        if own:
            side.flush()
        return dx


class ResNetTrainState(TrainState):
    """
    ResNet-50/101/152: conv1, bn1, layer1 and every BatchNorm frozen (resnet.py:48-55,86,123); the convolutions of layer2,
    layer3 (feature extractor) and layer4 (per-RoI head) train.  A frozen BatchNorm is an affine map per channel, so each
    conv+BN runs as ONE convolution with the folded weight W * scale[co]; the master (what SGD and weight decay act on) is
    the raw weight, its gradient = scale[co] * the folded weight's gradient, and the folded pack is rebuilt after every update.
    """
    C, V = 1024, 2048

    def __init__(self, model):
        super().__init__(model)
        fe = model._stage1_feature_extractor
        seq = fe._feature_extractor
        pk = fe.packed()
        n1 = len(seq[4])
        self.stem = (pk["stem"][0], pk["stem"][1])
        self.frozen_blocks = pk["blocks"][:n1]                      # layer1: folded inference packs
        with t.cuda.device(self.device):
            self.blocks = [_TrainBlock(b, "layer2.%d" % i) for i, b in enumerate(seq[5])] + \
                          [_TrainBlock(b, "layer3.%d" % i) for i, b in enumerate(seq[6])]
            l4 = model._stage3_detector_network._pool_to_feature_vector._layer4
            self.head_blocks = [_TrainBlock(b, "layer4.%d" % i) for i, b in enumerate(l4)]

    def trainable(self):
        out = {}
        for blk in self.blocks + self.head_blocks:
            for cname, c in blk.convs().items():
                out[blk.name + "." + cname] = c.raw
        out.update(rpn_conv=self.rpn_conv, rpn_head=self.rpn_head, head=self.head)
        return out

    def folded_convs(self):
        return {blk.name + "." + cname: c for blk in self.blocks + self.head_blocks for cname, c in blk.convs().items()}

    def after_update(self):
        pass                                     # the folded packs were rebuilt by frcnn_sgd_step_fold (same product as _TrainConv.refold)

    def _sync_backbone(self):
        for blk in self.blocks + self.head_blocks:
            for c in blk.convs().values():
                c.sync()

    # ---- stage 1 (resnet.py:38-46) ------------------------------------------------------------------------
    def features_forward(self, image):
        from .models import resnet
        lib = _lib()
        h, w = int(image.shape[2]), int(image.shape[3])
        h1, w1 = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = t.empty((h1, w1, 64), dtype=t.float32, device=self.device)
        nv.check(lib.frcnn_conv7x7_s2_c3(nv.ptr(image), nv.ptr(self.stem[0]), nv.ptr(self.stem[1]), nv.ptr(y), h, w, 64, nv.RELU,
                                         nv.stream_ptr()), "frcnn_conv7x7_s2_c3")
        h2, w2 = (h1 - 1) // 2 + 1, (w1 - 1) // 2 + 1
        cur = t.empty((1, h2, w2, 64), dtype=t.float32, device=self.device)
        nv.check(lib.frcnn_maxpool3x3_s2_nhwc(nv.ptr(y), nv.ptr(cur), h1, w1, 64, nv.stream_ptr()), "frcnn_maxpool3x3_s2_nhwc")
        h, w = h2, w2
        for pb in self.frozen_blocks:
            cur, h, w = resnet.run_block(cur, 1, h, w, pb)
        saved = []
        for blk in self.blocks:
            cur, h, w, sv = blk.forward(cur, 1, h, w)
            saved.append(sv)
        return cur[0], saved                                        # [fh][fw][1024]

    def features_backward(self, g, saved, grads):
        g = g.reshape(1, g.shape[0], g.shape[1], g.shape[2])
        side = _SideGrads(grads, self.device)
        for i in range(len(self.blocks) - 1, -1, -1):
            g = self.blocks[i].backward(g, saved[i], side, need_dx=(i > 0))       # layer1 below is frozen: no dx for block 0
        side.flush()                                                             # every gradient handed over, in production order, before SGD

    # ---- RoI features -> feature vector (resnet.py:109-118) -----------------------------------------------
    def head_forward(self, roi_out):
        S = int(roi_out.shape[0])
        cur = roi_out.reshape(S, 7, 7, 1024)
        h = w = 7
        saved = []
        for blk in self.head_blocks:
            cur, h, w, sv = blk.forward(cur, S, h, w)
            saved.append(sv)
        vec = t.empty((S, 2048), dtype=t.float32, device=self.device)
        nv.check(_lib().frcnn_spatial_mean_nhwc(nv.ptr(cur), nv.ptr(vec), S, h, w, 2048, nv.stream_ptr()), "frcnn_spatial_mean_nhwc")
        return vec, (saved, S, h, w)

    def head_backward(self, dvec, saved, grads, detail=None):
        blocks_saved, S, h, w = saved
        g = t.empty((S, h, w, 2048), dtype=t.float32, device=self.device)
        nv.check(_lib().frcnn_spatial_mean_backward(nv.ptr(dvec), nv.ptr(g), S, h, w, 2048, nv.stream_ptr()),
                 "frcnn_spatial_mean_backward")
        side = _SideGrads(grads, self.device)
        for i in range(len(self.head_blocks) - 1, -1, -1):
            g = self.head_blocks[i].backward(g, blocks_saved[i], side, need_dx=True)
        side.flush()       # (before "rpn_head": the order in which `grads` fills is the one-stream step's, also on a rank without a proposal batch)
        return g.reshape(S, 49 * 1024)

    def zero_head_grads(self, grads):
        # in the backward pass's production order: a rank without a proposal batch must exchange the same messages as the others
        for blk in reversed(self.head_blocks):
            convs = blk.convs()
            for cname in ("conv3", "conv2", "conv1", "downsample"):
                if cname in convs:
                    grads[blk.name + "." + cname] = t.zeros_like(convs[cname].raw)


class GradientAverager:
    """
    Data-parallel training (beyond the reference, which trains on one GPU): every rank runs `train_step` on its own sample
    and the weight gradients are averaged over the ranks before the SGD update, `torch.distributed.all_reduce` over RCCL/xGMI
    (backend "nccl") or gloo.

    OVERLAPPED with the backward pass: `train_step` collects its gradients in `track()`'s dict, which hands every tensor to
    `ready()` the moment the kernel producing it has been enqueued.  A tensor of `direct_bytes` or more (fc1's 411 MB gradient
    is produced FIRST, before the RPN and the whole backbone backward) is all-reduced in place right away with `async_op=True`:
    the process group's own stream waits for the producing kernel and the exchange runs under the rest of the backward.
    Smaller tensors are coalesced, in production order (the same on every rank), into flat buckets of `bucket_bytes` that are
    sent as soon as they fill (xGMI rings are per-link bound: few large messages).  `finish()` sends the last bucket, waits for
    every exchange (a stream-level wait on the GPU), divides by the world size and scatters the buckets back.
    Attach with `enable_data_parallel(model)`.  `__call__(grads)` is the one-shot form (all tensors at once, same arithmetic).
    """
    def __init__(self, group=None, bucket_bytes=64 << 20, direct_bytes=8 << 20):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("GradientAverager needs an initialised torch.distributed process group")
        self.dist, self.group, self.bucket_bytes, self.direct_bytes = dist, group, int(bucket_bytes), int(direct_bytes)
        self.world = dist.get_world_size(group)
        self._pending, self._pending_bytes, self._inflight = [], 0, []
        self.messages = 0                                # all-reduce calls of the last step (tests / reports)

    class _Tracked(dict):
        def __init__(self, owner):
            super().__init__()
            self._owner = owner

        def __setitem__(self, name, tensor):
            # a gradient is handed to the exchange exactly once, final: a second assignment would send it twice (or race with the
            # in-place reduce of the first)
            if name in self:
                raise RuntimeError("gradient %r was assigned twice in one step: the first tensor is already being all-reduced" % name)
            super().__setitem__(name, tensor)
            self._owner.ready(name, tensor)

    def track(self):
        """The gradient dict of one step: assigning grads[name] = tensor hands the (final) tensor to the exchange."""
        self._pending, self._pending_bytes, self._inflight, self.messages = [], 0, [], 0
        return GradientAverager._Tracked(self)

    def _send(self, flat, parts):
        work = self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._inflight.append((work, flat, parts))
        self.messages += 1

    def _flush(self):
        if self._pending:
            parts = self._pending
            self._send(t.cat([g.reshape(-1) for _, g in parts]), parts)
            self._pending, self._pending_bytes = [], 0

    def ready(self, name, g):
        nbytes = g.numel() * g.element_size()
        if nbytes >= self.direct_bytes and g.is_contiguous():
            self._send(g.view(-1), None)                 # in place, no staging copy
            return
        self._pending.append((name, g))
        self._pending_bytes += nbytes
        if self._pending_bytes >= self.bucket_bytes:
            self._flush()

    def finish(self):
        self._flush()
        inv = 1.0 / float(self.world)
        for work, flat, parts in self._inflight:
            work.wait()
            flat.mul_(inv)
            if parts is not None:
                off = 0
                for _, g in parts:
                    k = g.numel()
                    g.copy_(flat[off:off + k].view_as(g))
                    off += k
        self._inflight = []

    def abort(self):
        """After an exception between track() and finish(): waits for every exchange already started, so that no collective is left
        in flight on tensors that are about to be freed, and drops the step's state.  (The peers of a rank that failed mid-step still
        need their own error handling: the process group's timeout ends their wait.)"""
        for work, _, _ in self._inflight:
            try:
                work.wait()
            except Exception:
                pass
        self._pending, self._pending_bytes, self._inflight = [], 0, []

    def __call__(self, grads):
        tracked = self.track()
        for name in sorted(grads):
            tracked[name] = grads[name]
        self.finish()


def enable_data_parallel(model, group=None, bucket_bytes=64 << 20, direct_bytes=8 << 20):
    """Average the weight gradients of every `train_step` over the ranks of `group`, overlapped with the backward pass
    (see GradientAverager)."""
    model._gradient_sync = GradientAverager(group, bucket_bytes, direct_bytes)
    return model


def make_train_state(model):
    if model._num_classes > nv.MAX_NUM_CLASSES_TRAIN:
        raise NotImplementedError("train_step supports num_classes <= %d (its loss / gradient kernels keep the 128-row stacked head); "
                                  "inference supports up to %d" % (nv.MAX_NUM_CLASSES_TRAIN, nv.MAX_NUM_CLASSES))
    return ResNetTrainState(model) if model._is_resnet else VGG16TrainState(model)


def _flat_anchor_indices(index_map, fw):
    a = np.asarray(index_map).reshape(-1, 3).astype(np.int64)
    return (a[:, 0] * fw + a[:, 1]) * 9 + a[:, 2]


def train_step(model, optimizer, image_data, anchor_map, anchor_valid_map, gt_rpn_map, gt_rpn_object_indices,
               gt_rpn_background_indices, gt_boxes, detail=None):
    """
    One training step on one image: same arguments and return value as faster_rcnn.py:228-362.
    `gt_boxes` is [[Box]] (objects with .class_index and .corners, datasets/training_sample.py) .
    `detail`, if a dict, receives gradients and intermediates for the parity tests.
    """
    global _GRAD_MATH
    _GRAD_MATH = nv.GRAD_MATHS[model.grad_math]
    try:
        return _train_step(model, optimizer, image_data, anchor_map, anchor_valid_map, gt_rpn_map, gt_rpn_object_indices,
                           gt_rpn_background_indices, gt_boxes, detail)
    except BaseException:
        sync = getattr(model, "_gradient_sync", None)
        if sync is not None:
            sync.abort()            # never leave async all-reduces in flight on tensors this frame is about to free
        raise
    finally:
        _GRAD_MATH = 0              # the module-level helpers (gemm_tn, conv3x3_wgrad, ...) default to float32 outside a step


def _train_step(model, optimizer, image_data, anchor_map, anchor_valid_map, gt_rpn_map, gt_rpn_object_indices,
                gt_rpn_background_indices, gt_boxes, detail):
    model.train()
    assert image_data.shape[0] == 1, "Batch size must be 1"
    assert len(gt_rpn_map.shape) == 5 and gt_rpn_map.shape[0] == 1, "Batch size must be 1"
    assert len(gt_rpn_object_indices) == 1, "Batch size must be 1"
    assert len(gt_rpn_background_indices) == 1, "Batch size must be 1"
    assert len(gt_boxes) == 1, "Batch size must be 1"
    lr, momentum, weight_decay = sgd_hyper_parameters(optimizer)
    st = model._training_state()
    dev = st.device
    lib = _lib()
    image = rt.as_f32_cuda(image_data, "image_data")
    H, W = int(image.shape[2]), int(image.shape[3])
    ncls = model._num_classes
    nd = 4 * (ncls - 1)
    C, V = st.C, st.V
    hc0 = time.perf_counter() if HOST_CLOCKS is not None else 0.0
    with t.no_grad(), t.cuda.device(dev):
        s = nv.stream_ptr()
        # ---- stage 1 forward, keeping what the backward needs ------------------------------------------
        # parity-test hook: detail["inject"] prescribes forward activations (see VGG16TrainState.features_forward)
        inject = (detail or {}).get("inject") or {}
        if inject and not isinstance(st, VGG16TrainState):
            raise NotImplementedError("activation injection is implemented for the VGG-16 train state")
        fm, fsaved = st.features_forward(image, inject) if inject else st.features_forward(image)   # [fh][fw][C]
        fh, fw = int(fm.shape[0]), int(fm.shape[1])
        P = fh * fw
        # ---- stage 2 forward (rpn.py:88-156, 12000 / 2000 in training: faster_rcnn.py:301-302) --------
        trunk = conv3x3_forward(fm, st.rpn_conv, st.rpn_conv_b, C, C, st.winograd)
        trunk = inject.get("rpn_trunk", trunk)
        head = t.zeros((P, 128), dtype=t.float32, device=dev)
        wsb = int(lib.frcnn_linear_workspace_bytes(P, 45, C))
        ws = _ws(wsb, dev)
        nv.check(lib.frcnn_linear(nv.ptr(trunk), C, nv.ptr(st.rpn_head), nv.ptr(st.rpn_head_b), nv.ptr(head), 128, P, 45, C,
                                  0, nv.ptr(ws), wsb, s), "frcnn_linear")
        amap = rt.to_device_map(anchor_map, dev)
        vmap = rt.to_device_map(anchor_valid_map, dev)
        ctx = rpn_mod.scratch_context(dev, H, W)
        pre_nms, post_nms = 12000, 2000
        scores = t.empty((P * 9,), dtype=t.float32, device=dev)
        sorted_idx = t.empty((pre_nms,), dtype=t.int32, device=dev)
        props = t.empty((post_nms, 4), dtype=t.float32, device=dev)
        counts = t.zeros((4,), dtype=t.int32, device=dev)
        nv.check(lib.frcnn_rpn_proposals(ctx.handle, nv.ptr(head), 128, nv.ptr(amap),
                                         None if model._allow_edge_proposals else nv.ptr(vmap), fh, fw, H, W, pre_nms, post_nms,
                                         float(model.rpn_nms_threshold), float(model.rpn_min_side), nv.ptr(scores),
                                         nv.ptr(sorted_idx), nv.ptr(props), nv.ptr(counts), s), "frcnn_rpn_proposals")
        # ---- anchor mini-batch (faster_rcnn.py:364-419): python RNG, as the reference ------------------
        pos, neg = gt_rpn_object_indices[0], gt_rpn_background_indices[0]
        mb = model._rpn_minibatch_size
        assert len(pos) + len(neg) >= mb, "Image has insufficient anchors for RPN minibatch size of %d" % mb
        assert len(pos) > 0, "Image does not have any positive anchors"
        assert mb % 2 == 0, "RPN minibatch size must be evenly divisible"
        n_pos = min(mb // 2, len(pos))
        n_neg = mb - n_pos
        pi = random.sample(range(len(pos)), n_pos)
        ni = random.sample(range(len(neg)), n_neg)
        flat = np.concatenate([_flat_anchor_indices(np.asarray(pos)[pi], fw), _flat_anchor_indices(np.asarray(neg)[ni], fw)])
        rpn_sample = t.from_numpy(flat.astype(np.int32)).to(dev)
        rpn_map = rt.as_f32_cuda(gt_rpn_map.to(dev) if isinstance(gt_rpn_map, t.Tensor) else t.from_numpy(gt_rpn_map).to(dev),
                                 "gt_rpn_map").reshape(P * 9, 6)
        # ---- proposal labelling + sampling (faster_rcnn.py:421-561) -----------------------------------
        boxes = gt_boxes[0]
        gt_corners = t.from_numpy(np.array([b.corners for b in boxes], dtype=np.float32)).to(dev)
        gt_cls = t.from_numpy(np.array([b.class_index for b in boxes], dtype=np.int32)).to(dev)
        M = int(gt_corners.shape[0])
        cap = post_nms + M
        lab_props = t.empty((cap, 4), dtype=t.float32, device=dev)
        lab_cls = t.empty((cap,), dtype=t.int32, device=dev)
        lab_onehot = t.empty((cap, ncls), dtype=t.float32, device=dev)
        lab_deltas = t.empty((cap, 2, nd), dtype=t.float32, device=dev)
        lab_count = t.zeros((1,), dtype=t.int32, device=dev)
        means = (nv.C.c_float * 4)(*[float(v) for v in model._detector_box_delta_means])
        stds = (nv.C.c_float * 4)(*[float(v) for v in model._detector_box_delta_stds])
        nv.check(lib.frcnn_label_proposals(nv.ptr(props), counts.data_ptr() + 8, post_nms, nv.ptr(gt_corners), nv.ptr(gt_cls), M,
                                           ncls, 0.0, 0.5, means, stds, nv.ptr(lab_props), nv.ptr(lab_cls), nv.ptr(lab_onehot),
                                           nv.ptr(lab_deltas), nv.ptr(lab_count), s), "frcnn_label_proposals")
        hc1 = time.perf_counter() if HOST_CLOCKS is not None else 0.0
        K = int(lab_count.item())                                     # host sync 1 (the reference's len()/where)
        hc2 = time.perf_counter() if HOST_CLOCKS is not None else 0.0
        class_indices = lab_cls[:K].cpu().to(t.int64)
        sample_idx = _sample_proposal_indices(class_indices, model._proposal_batch_size, 0.25)
        S = int(sample_idx.shape[0])
        losses = t.zeros((4,), dtype=t.float32, device=dev)
        sync = getattr(model, "_gradient_sync", None)
        grads = sync.track() if sync is not None else {}              # data parallel: every gradient is exchanged as soon as it exists
        dfm = None
        if S > 0:
            idx_dev = sample_idx.to(t.int32).to(dev)
            s_props = t.empty((S, 4), dtype=t.float32, device=dev)
            s_onehot = t.empty((S, ncls), dtype=t.float32, device=dev)
            s_deltas = t.empty((S, 2, nd), dtype=t.float32, device=dev)
            for src, dst, rf in ((lab_props, s_props, 4), (lab_onehot, s_onehot, ncls), (lab_deltas, s_deltas, 2 * nd)):
                nv.check(lib.frcnn_gather_rows(nv.ptr(src), nv.ptr(idx_dev), S, rf, nv.ptr(dst), s), "frcnn_gather_rows")
            # ---- stage 3 forward (detector.py:65-80) ------------------------------------------------
            roi_out = t.empty((S, 49 * C), dtype=t.float32, device=dev)
            cnt = t.tensor([S], dtype=t.int32, device=dev)
            dn = model._stage3_detector_network
            if dn.pooling == "align":
                nv.check(lib.frcnn_roi_align(nv.ptr(fm), fh, fw, C, nv.ptr(s_props), nv.ptr(cnt), S, 7, 1.0 / 16.0, dn.sampling_ratio, 0,
                                             nv.ptr(roi_out), s), "frcnn_roi_align")
            else:
                nv.check(lib.frcnn_roi_pool(nv.ptr(fm), fh, fw, C, nv.ptr(s_props), nv.ptr(cnt), S, 7, 1.0 / 16.0,
                                            nv.ptr(roi_out), s), "frcnn_roi_pool")
            roi_out = inject.get("roi_out", roi_out)
            vec, hsaved = st.head_forward(roi_out, inject) if inject else st.head_forward(roi_out)
            logits = vgg16.linear(vec, st.head, st.head_b, ncls + nd, relu=False)
            classes = t.empty((S, ncls), dtype=t.float32, device=dev)
            nv.check(lib.frcnn_softmax_rows(nv.ptr(logits), ncls + nd, nv.ptr(classes), S, ncls, s), "frcnn_softmax_rows")
            deltas = logits[:, ncls:].contiguous()
            dlogits = t.empty((S, 128), dtype=t.float32, device=dev)
            nv.check(lib.frcnn_detector_loss(nv.ptr(classes), nv.ptr(deltas), nv.ptr(s_onehot), nv.ptr(s_deltas), S, ncls,
                                             losses.data_ptr() + 8, nv.ptr(dlogits), 128, s), "frcnn_detector_loss")
            # ---- stage 3 backward ---------------------------------------------------------------------
            grads["head"] = gemm_tn(dlogits, 128, vec, V, 128, V, S)
            dl_t, sp = transpose(dlogits, S, 128, 128)
            dvec = gemm_tn(dl_t, sp, st.head, V, S, V, 128)
            droi = st.head_backward(dvec, hsaved, grads, detail)
            dfm = t.empty((fh, fw, C), dtype=t.float32, device=dev)
            if dn.pooling == "align":
                nv.check(lib.frcnn_roi_align_backward(nv.ptr(s_props), S, fh, fw, C, 7, 1.0 / 16.0, dn.sampling_ratio, 0, nv.ptr(droi),
                                                      nv.ptr(dfm), 0, s), "frcnn_roi_align_backward")
            else:
                wsb = int(lib.frcnn_roi_pool_backward_workspace_bytes(S, 7, C))
                ws = _ws(wsb, dev)
                nv.check(lib.frcnn_roi_pool_backward(nv.ptr(fm), fh, fw, C, nv.ptr(s_props), S, 7, 1.0 / 16.0, nv.ptr(droi),
                                                     nv.ptr(dfm), 0, nv.ptr(ws), wsb, s), "frcnn_roi_pool_backward")
            if detail is not None:
                detail.update(sampled_props=s_props, sampled_onehot=s_onehot, sampled_deltas=s_deltas, classes=classes,
                              deltas=deltas, dlogits=dlogits, vec=vec, roi_out=roi_out, dfm_roi=dfm.clone(), droi=droi)
        else:
            grads["head"] = t.zeros_like(st.head)
            st.zero_head_grads(grads)
        # ---- RPN losses + backward (rpn.py:176-272) ---------------------------------------------------
        dhead = t.empty((P, 128), dtype=t.float32, device=dev)
        nv.check(lib.frcnn_rpn_loss(nv.ptr(head), 128, P, nv.ptr(rpn_sample), int(rpn_sample.shape[0]), nv.ptr(rpn_map),
                                    nv.ptr(losses), nv.ptr(dhead), s), "frcnn_rpn_loss")
        grads["rpn_head"] = gemm_tn(dhead, 128, trunk, C, 128, C, P)
        dhead_t, pp = transpose(dhead, P, 128, 128)
        dtrunk = gemm_tn(dhead_t, pp, st.rpn_head, C, P, C, 128).reshape(fh, fw, C)
        relu_backward(dtrunk, trunk)
        grads["rpn_conv"] = conv3x3_wgrad(fm, dtrunk, C, C)
        g = conv3x3_dgrad(dtrunk, st.rpn_conv, C, C, st.zero_bias, st.winograd)
        if dfm is not None:
            nv.check(lib.frcnn_add_inplace(nv.ptr(g), nv.ptr(dfm), g.numel(), s), "frcnn_add_inplace")
        if detail is not None:
            detail.update(dfm=g.clone(), dhead=dhead, head=head, trunk=trunk, fm=fm, rpn_sample=rpn_sample,
                          proposals=props, counts=counts, labelled=(lab_props[:K], lab_cls[:K], lab_onehot[:K], lab_deltas[:K]),
                          sample_idx=sample_idx)
        # ---- stage 1 backward ---------------------------------------------------------------------------
        st.features_backward(g, fsaved, grads)
        # ---- SGD (torch.optim.SGD.step, __main__.py:98-105) --------------------------------------------
        if sync is not None:
            sync.finish()                                             # data parallel: the exchanges started during the backward complete
        if detail is not None:
            detail["grads"] = {k: v.clone() for k, v in grads.items()}
        st.apply_sgd(grads, lr, momentum, weight_decay)
        hc3 = time.perf_counter() if HOST_CLOCKS is not None else 0.0
        lv = losses.cpu().numpy()                                     # host sync 2
        if HOST_CLOCKS is not None:
            HOST_CLOCKS.append((hc0, hc1, hc2, hc3, time.perf_counter()))
    # total in float32 left to right, as the reference adds the four float32 scalars (faster_rcnn.py:344)
    total = np.float32(np.float32(np.float32(lv[0] + lv[1]) + lv[2]) + lv[3])
    return model.Loss(rpn_class=float(lv[0]), rpn_regression=float(lv[1]), detector_class=float(lv[2]),
                      detector_regression=float(lv[3]), total=float(total))


def _sample_proposal_indices(class_indices, max_proposals, positive_fraction):
    """faster_rcnn.py:512-561 on the host: returns the selected row indices (int64 CPU tensor)."""
    n = int(class_indices.shape[0])
    if max_proposals <= 0:
        return t.arange(n)
    positive_indices = t.where(class_indices > 0)[0]
    negative_indices = t.where(class_indices <= 0)[0]
    num_samples = min(max_proposals, n)
    num_positive_samples = min(round(num_samples * positive_fraction), len(positive_indices))
    num_negative_samples = min(num_samples - num_positive_samples, len(negative_indices))
    if num_positive_samples <= 0 or num_negative_samples <= 0:
        return t.zeros((0,), dtype=t.int64)
    positive_sample_indices = positive_indices[t.randperm(len(positive_indices))[0:num_positive_samples]]
    negative_sample_indices = negative_indices[t.randperm(len(negative_indices))[0:num_negative_samples]]
    return t.cat([positive_sample_indices, negative_sample_indices])
