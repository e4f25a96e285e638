"""
The train step of FasterRCNNModel (reference: models/faster_rcnn.py:228-362 `train_step`, the samplers at
:364-561, the losses at models/rpn.py:176-272 and models/detector.py:83-155, torch.optim.SGD built at
__main__.py:98-105) for the VGG-16 backbone -- SURVEY.md section 8 rows f2 + f3.

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
    requires_grad (blocks 1-2 of VGG-16 frozen, biases never updated);
  * host-side randomness is drawn exactly as the reference draws it (python `random.sample` for the anchor
    mini-batch, `torch.randperm` on the CPU generator for the proposal batch), so a seeded run selects the
    same samples;
  * there are two host synchronisations per step, as in the reference: the count/labels of the labelled
    proposals (the reference's `len()` / `t.where` calls) and the final read of the loss values.
"""
import random

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


def gemm_tn(a, lda, b, ldb, m, n, r, out=None, ldc=None):
    """C[m][n] = sum_r A[r][m] B[r][n] (frcnn_gemm_tn)."""
    dev = a.device
    ldc = n if ldc is None else ldc
    c = t.empty((m, ldc), dtype=t.float32, device=dev) if out is None else out
    lib = _lib()
    wsb = int(lib.frcnn_gemm_tn_workspace_bytes(m, n, r))
    ws = _ws(wsb, dev) if wsb else None
    nv.check(lib.frcnn_gemm_tn(nv.ptr(a), lda, nv.ptr(b), ldb, nv.ptr(c), ldc, m, n, r, nv.ptr(ws), wsb, nv.stream_ptr()),
             "frcnn_gemm_tn")
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
    nv.check(lib.frcnn_conv3x3_wgrad(nv.ptr(x_hwc), nv.ptr(dz_hwc), nv.ptr(dwp), h, w, cin, cout, nv.ptr(ws), wsb,
                                     nv.stream_ptr()), "frcnn_conv3x3_wgrad")
    return dwp


def conv3x3_dgrad(dz_hwc, wp, cin, cout, zero_bias):
    """Gradient with respect to the input of y = conv3x3(x, wp): a 3x3 conv of dz (cout channels) to cin channels."""
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


class TrainState:
    """Packed master weights + momentum buffers of one FasterRCNNModel (VGG-16)."""
    def __init__(self, model):
        self.model = model
        fe = model._stage1_feature_extractor
        rp = model._stage2_region_proposal_network
        dn = model._stage3_detector_network
        pv = dn._pool_to_feature_vector
        if model.math_mode != "f32":
            raise NotImplementedError("training runs in the exact-f32 math mode")
        if pv._dropout1.p > 0 or pv._dropout2.p > 0:
            raise NotImplementedError("dropout > 0 is not implemented in the train step (reference default: 0.0)")
        # clones: the inference-side packed caches are rebuilt from the parameters, these are the training masters
        self.conv = [(wp.clone(), b.clone()) for wp, b in fe.packed()]
        wc, bc, wh, bh = rp.packed()
        self.rpn_conv, self.rpn_conv_b, self.rpn_head, self.rpn_head_b = wc.clone(), bc.clone(), wh.clone(), bh.clone()
        w1p, b1, w2, b2 = pv.packed()
        self.fc1, self.fc1_b, self.fc2, self.fc2_b = w1p.clone(), b1.clone(), w2.clone(), b2.clone()
        hw, hb = dn.packed()
        self.head, self.head_b = hw.clone(), hb.clone()
        self.device = self.fc1.device
        self.zero_bias = t.zeros((1024,), dtype=t.float32, device=self.device)
        self.momentum = {}
        self.steps = 0
        self.dirty = False

    def trainable(self):
        """name -> packed master tensor, for everything SGD updates."""
        out = {"conv%d" % i: self.conv[i][0] for i in _TRAINABLE_CONVS}
        out.update(rpn_conv=self.rpn_conv, rpn_head=self.rpn_head, fc1=self.fc1, fc2=self.fc2, head=self.head)
        return out

    def apply_sgd(self, grads, lr, momentum, weight_decay):
        lib = _lib()
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
            nv.check(lib.frcnn_sgd_step(nv.ptr(w), nv.ptr(g), nv.ptr(buf), w.numel(), lr, momentum, weight_decay, first,
                                        nv.stream_ptr()), "frcnn_sgd_step")
        self.steps += 1
        self.dirty = True

    @t.no_grad()
    def sync_to_parameters(self):
        """Writes the packed masters back into the nn.Parameters (OIHW convs, (C,7,7)-ordered fc1, separate heads)."""
        if not self.dirty:
            return
        m = self.model
        fe = m._stage1_feature_extractor
        for i in _TRAINABLE_CONVS:
            conv = fe.convs()[i]
            co, ci = int(conv.weight.shape[0]), int(conv.weight.shape[1])
            conv.weight.copy_(self.conv[i][0].permute(1, 2, 0).reshape(co, ci, 3, 3))
        rp = m._stage2_region_proposal_network
        c = int(rp._rpn_conv1.weight.shape[0])
        rp._rpn_conv1.weight.copy_(self.rpn_conv.permute(1, 2, 0).reshape(c, c, 3, 3))
        rp._rpn_class.weight.copy_(self.rpn_head[0:9].reshape(9, c, 1, 1))
        rp._rpn_boxes.weight.copy_(self.rpn_head[9:45].reshape(36, c, 1, 1))
        dn = m._stage3_detector_network
        pv = dn._pool_to_feature_vector
        pv._fc1.weight.copy_(self.fc1.reshape(4096, 49, 512).permute(0, 2, 1).reshape(4096, 512 * 49))
        pv._fc2.weight.copy_(self.fc2)
        ncls = m._num_classes
        dn._classifier.weight.copy_(self.head[0:ncls])
        dn._regressor.weight.copy_(self.head[ncls:ncls + 4 * (ncls - 1)])
        self.dirty = False


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
    model.train()
    assert image_data.shape[0] == 1, "Batch size must be 1"
    assert len(gt_rpn_map.shape) == 5 and gt_rpn_map.shape[0] == 1, "Batch size must be 1"
    assert len(gt_rpn_object_indices) == 1, "Batch size must be 1"
    assert len(gt_rpn_background_indices) == 1, "Batch size must be 1"
    assert len(gt_boxes) == 1, "Batch size must be 1"
    if model._is_resnet:
        raise NotImplementedError("train_step is implemented for the VGG-16 backbone")
    lr, momentum, weight_decay = sgd_hyper_parameters(optimizer)
    st = model._training_state()
    dev = st.device
    lib = _lib()
    image = rt.as_f32_cuda(image_data, "image_data")
    H, W = int(image.shape[2]), int(image.shape[3])
    ncls = model._num_classes
    nd = 4 * (ncls - 1)
    with t.no_grad(), t.cuda.device(dev):
        s = nv.stream_ptr()
        # ---- stage 1 forward, keeping what the backward needs (vgg16.py:76-96) -----------------------
        x_in, y_out = {}, {}
        cur = t.empty((H, W, 64), dtype=t.float32, device=dev)
        nv.check(lib.frcnn_conv3x3_c3(nv.ptr(image), nv.ptr(st.conv[0][0]), nv.ptr(st.conv[0][1]), nv.ptr(cur), H, W, 64,
                                      nv.RELU, s), "frcnn_conv3x3_c3")
        for i in range(1, 13):
            _, cin, cout, pool = vgg16._LAYERS[i]
            wp, b = st.conv[i]
            if i in _TRAINABLE_CONVS:
                x_in[i] = cur
                y = vgg16.conv3x3(cur, wp, b, cin, cout, relu=True, pool=False)
                y_out[i] = y
                cur = maxpool2x2(y) if pool else y
            else:
                cur = vgg16.conv3x3(cur, wp, b, cin, cout, relu=True, pool=pool)      # frozen: pool fused
        fm = cur                                                                       # [fh][fw][512]
        fh, fw = int(fm.shape[0]), int(fm.shape[1])
        P = fh * fw
        # ---- stage 2 forward (rpn.py:88-156, 12000 / 2000 in training: faster_rcnn.py:301-302) --------
        trunk = vgg16.conv3x3(fm, st.rpn_conv, st.rpn_conv_b, 512, 512, relu=True, pool=False)
        head = t.zeros((P, 128), dtype=t.float32, device=dev)
        wsb = int(lib.frcnn_linear_workspace_bytes(P, 45, 512))
        ws = _ws(wsb, dev)
        nv.check(lib.frcnn_linear(nv.ptr(trunk), 512, nv.ptr(st.rpn_head), nv.ptr(st.rpn_head_b), nv.ptr(head), 128, P, 45, 512,
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
        K = int(lab_count.item())                                     # host sync 1 (the reference's len()/where)
        class_indices = lab_cls[:K].cpu().to(t.int64)
        sample_idx = _sample_proposal_indices(class_indices, model._proposal_batch_size, 0.25)
        S = int(sample_idx.shape[0])
        losses = t.zeros((4,), dtype=t.float32, device=dev)
        grads = {}
        dfm = None
        if S > 0:
            idx_dev = sample_idx.to(t.int32).to(dev)
            s_props = t.empty((S, 4), dtype=t.float32, device=dev)
            s_onehot = t.empty((S, ncls), dtype=t.float32, device=dev)
            s_deltas = t.empty((S, 2, nd), dtype=t.float32, device=dev)
            for src, dst, rf in ((lab_props, s_props, 4), (lab_onehot, s_onehot, ncls), (lab_deltas, s_deltas, 2 * nd)):
                nv.check(lib.frcnn_gather_rows(nv.ptr(src), nv.ptr(idx_dev), S, rf, nv.ptr(dst), s), "frcnn_gather_rows")
            # ---- stage 3 forward (detector.py:65-80) ------------------------------------------------
            roi_out = t.empty((S, 49 * 512), dtype=t.float32, device=dev)
            cnt = t.tensor([S], dtype=t.int32, device=dev)
            nv.check(lib.frcnn_roi_pool(nv.ptr(fm), fh, fw, 512, nv.ptr(s_props), nv.ptr(cnt), S, 7, 1.0 / 16.0,
                                        nv.ptr(roi_out), s), "frcnn_roi_pool")
            h1 = vgg16.linear(roi_out, st.fc1, st.fc1_b, 4096, relu=True)
            h2 = vgg16.linear(h1, st.fc2, st.fc2_b, 4096, relu=True)
            logits = vgg16.linear(h2, st.head, st.head_b, ncls + nd, relu=False)
            classes = t.empty((S, ncls), dtype=t.float32, device=dev)
            nv.check(lib.frcnn_softmax_rows(nv.ptr(logits), ncls + nd, nv.ptr(classes), S, ncls, s), "frcnn_softmax_rows")
            deltas = logits[:, ncls:].contiguous()
            dlogits = t.empty((S, 128), dtype=t.float32, device=dev)
            nv.check(lib.frcnn_detector_loss(nv.ptr(classes), nv.ptr(deltas), nv.ptr(s_onehot), nv.ptr(s_deltas), S, ncls,
                                             losses.data_ptr() + 8, nv.ptr(dlogits), 128, s), "frcnn_detector_loss")
            # ---- stage 3 backward ---------------------------------------------------------------------
            grads["head"] = gemm_tn(dlogits, 128, h2, 4096, 128, 4096, S)
            dl_t, sp = transpose(dlogits, S, 128, 128)
            dh2 = gemm_tn(dl_t, sp, st.head, 4096, S, 4096, 128)
            if detail is not None:
                detail["dh2"] = dh2.clone()
            relu_backward(dh2, h2)
            grads["fc2"] = gemm_tn(dh2, 4096, h1, 4096, 4096, 4096, S)
            dh2_t, sp = transpose(dh2, S, 4096, 4096)
            dh1 = gemm_tn(dh2_t, sp, st.fc2, 4096, S, 4096, 4096)
            if detail is not None:
                detail["dh1"] = dh1.clone()
            relu_backward(dh1, h1)
            grads["fc1"] = gemm_tn(dh1, 4096, roi_out, 49 * 512, 4096, 49 * 512, S)
            dh1_t, sp = transpose(dh1, S, 4096, 4096)
            droi = gemm_tn(dh1_t, sp, st.fc1, 49 * 512, S, 49 * 512, 4096)
            dfm = t.empty((fh, fw, 512), dtype=t.float32, device=dev)
            wsb = int(lib.frcnn_roi_pool_backward_workspace_bytes(S, 7, 512))
            ws = _ws(wsb, dev)
            nv.check(lib.frcnn_roi_pool_backward(nv.ptr(fm), fh, fw, 512, nv.ptr(s_props), S, 7, 1.0 / 16.0, nv.ptr(droi),
                                                 nv.ptr(dfm), 0, nv.ptr(ws), wsb, s), "frcnn_roi_pool_backward")
            if detail is not None:
                detail.update(sampled_props=s_props, sampled_onehot=s_onehot, sampled_deltas=s_deltas, classes=classes,
                              deltas=deltas, dlogits=dlogits, h1=h1, h2=h2, roi_out=roi_out, dfm_roi=dfm.clone(), droi=droi)
        else:
            for name in ("head", "fc2", "fc1"):
                grads[name] = t.zeros_like(st.trainable()[name])
        # ---- RPN losses + backward (rpn.py:176-272) ---------------------------------------------------
        dhead = t.empty((P, 128), dtype=t.float32, device=dev)
        nv.check(lib.frcnn_rpn_loss(nv.ptr(head), 128, P, nv.ptr(rpn_sample), int(rpn_sample.shape[0]), nv.ptr(rpn_map),
                                    nv.ptr(losses), nv.ptr(dhead), s), "frcnn_rpn_loss")
        grads["rpn_head"] = gemm_tn(dhead, 128, trunk, 512, 128, 512, P)
        dhead_t, pp = transpose(dhead, P, 128, 128)
        dtrunk = gemm_tn(dhead_t, pp, st.rpn_head, 512, P, 512, 128).reshape(fh, fw, 512)
        relu_backward(dtrunk, trunk)
        grads["rpn_conv"] = conv3x3_wgrad(fm, dtrunk, 512, 512)
        g = conv3x3_dgrad(dtrunk, st.rpn_conv, 512, 512, st.zero_bias)
        if dfm is not None:
            nv.check(lib.frcnn_add_inplace(nv.ptr(g), nv.ptr(dfm), g.numel(), s), "frcnn_add_inplace")
        if detail is not None:
            detail.update(dfm=g.clone(), dhead=dhead, head=head, trunk=trunk, fm=fm, rpn_sample=rpn_sample,
                          proposals=props, counts=counts, labelled=(lab_props[:K], lab_cls[:K], lab_onehot[:K], lab_deltas[:K]),
                          sample_idx=sample_idx)
        # ---- stage 1 backward (autograd of vgg16.py:84-96; blocks 1-2 are frozen) ----------------------
        for i in range(12, 3, -1):
            _, cin, cout, _ = vgg16._LAYERS[i]
            relu_backward(g, y_out[i])
            grads["conv%d" % i] = conv3x3_wgrad(x_in[i], g, cin, cout)
            if i > 4:
                gx = conv3x3_dgrad(g, st.conv[i][0], cin, cout, st.zero_bias)
                g = maxpool2x2_backward(y_out[i - 1], gx) if (i - 1) in _POOL_AFTER else gx
        # ---- SGD (torch.optim.SGD.step, __main__.py:98-105) --------------------------------------------
        if detail is not None:
            detail["grads"] = {k: v.clone() for k, v in grads.items()}
        st.apply_sgd(grads, lr, momentum, weight_decay)
        lv = losses.cpu().numpy()                                     # host sync 2
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
