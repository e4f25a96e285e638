"""
Training path (SURVEY.md section 8 rows f2 + f3) on a real MI355X, through the C ABI:
  * every backward operator against torch-CPU autograd (float64 ground truth, float32 torch as the yardstick)
    or oracle/train_oracle.py on the same seeded inputs;
  * FasterRCNNModel.train_step against tests/golden/train_vgg16_*.npz, which oracle/make_golden.py --train
    captured from the imported reference's own train_step (losses, gradients, weight updates, the sampled
    anchors / proposals under the same RNG seeds).
Tolerances: index / selection work exact; float32 GEMM-class results within a few float32-torch error
magnitudes of the float64 truth (stated per test); losses 1e-5 relative.
"""
import ctypes as C
import os
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fasterrcnn_amd import _native as nv
from fasterrcnn_amd import synthetic
from fasterrcnn_amd import training as T
from fasterrcnn_amd.datasets.training_sample import Box
from oracle import frcnn_oracle as O
from oracle import train_oracle as TO

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def S():
    return nv.stream_ptr()


def gpu(x):
    return torch.as_tensor(x).to(DEV).contiguous()


_KEEP = []


def dptr(x):
    """Device pointer of a fresh device copy of x; the copy is kept alive (a temporary would be freed and its
    memory reused before the kernel runs)."""
    d = gpu(x)
    _KEEP.append(d)
    if len(_KEEP) > 64:
        torch.cuda.synchronize()
        del _KEEP[:32]
    return nv.ptr(d)


def err_vs_f64(got, truth64, yard32):
    """max |got - truth| relative to the truth's scale, and the same for the float32 torch yardstick."""
    scale = max(float(np.abs(truth64).max()), 1e-30)
    return float(np.abs(got.astype(np.float64) - truth64).max()) / scale, float(np.abs(yard32.astype(np.float64) - truth64).max()) / scale


# ---- frcnn_gemm_tn ------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,R,lda,ldb", [(128, 4096, 37, 128, 4096), (101, 512, 128, 104, 512), (300, 130, 77, 300, 132),
                                           (2294, 512, 128, 2296, 512), (128, 512, 5000, 128, 512), (7, 6, 3, 8, 8),
                                           (256, 256, 4096, 256, 256)])
def test_gemm_tn(M, N, R, lda, ldb):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + R)
    a = torch.randn((R, lda), generator=g)
    b = torch.randn((R, ldb), generator=g)
    truth = (a[:, :M].double().T @ b[:, :N].double()).numpy()
    yard = (a[:, :M].T @ b[:, :N]).numpy()
    ey = float(np.abs(yard.astype(np.float64) - truth).max()) / float(np.abs(truth).max())
    da, db = gpu(a), gpu(b)
    c = torch.full((M, N), float("nan"), device=DEV)
    lib = nv.lib()
    wsb = int(lib.frcnn_gemm_tn_workspace_bytes(M, N, R))
    ws = torch.empty((wsb // 4 + 1,), device=DEV)
    nv.check(lib.frcnn_gemm_tn(nv.ptr(da), lda, nv.ptr(db), ldb, nv.ptr(c), N, M, N, R, nv.ptr(ws), wsb, S()), "gemm_tn")
    # float32 accumulation over R terms: a chain of R/2 MFMA steps (shorter with split-R); torch's blocked sum is the yardstick
    tol = max(4 * ey, 1.2e-7 * R ** 0.5)
    e, ey = err_vs_f64(c.cpu().numpy(), truth, yard)
    assert e <= tol, (e, ey)
    # without workspace (no split) the result is the same up to summation order
    c2 = torch.empty((M, N), device=DEV)
    nv.check(lib.frcnn_gemm_tn(nv.ptr(da), lda, nv.ptr(db), ldb, nv.ptr(c2), N, M, N, R, None, 0, S()), "gemm_tn")
    e2, _ = err_vs_f64(c2.cpu().numpy(), truth, yard)
    assert e2 <= tol, (e2, ey)
    # deterministic
    c3 = torch.empty((M, N), device=DEV)
    nv.check(lib.frcnn_gemm_tn(nv.ptr(da), lda, nv.ptr(db), ldb, nv.ptr(c3), N, M, N, R, nv.ptr(ws), wsb, S()), "gemm_tn")
    assert torch.equal(c, c3)


def test_gemm_tn_rejects_bad_arguments():
    lib = nv.lib()
    x = torch.zeros((64, 64), device=DEV)
    assert lib.frcnn_gemm_tn(nv.ptr(x), 63, nv.ptr(x), 64, nv.ptr(x), 64, 64, 64, 64, None, 0, S()) == -1    # lda % 4
    assert lib.frcnn_gemm_tn(nv.ptr(x), 64, nv.ptr(x), 64, nv.ptr(x), 63, 64, 63, 64, None, 0, S()) == -1    # ldc odd
    assert lib.frcnn_gemm_tn(None, 64, nv.ptr(x), 64, nv.ptr(x), 64, 64, 64, 64, None, 0, S()) == -1


# ---- conv3x3 backward -----------------------------------------------------------------------------------
def conv_case(H, W, cin, cout, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((1, cin, H, W), generator=g)
    w = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5
    dz = torch.randn((1, cout, H, W), generator=g)
    return x, w, dz


def torch_conv_grads(x, w, dz, dtype):
    x = x.to(dtype).requires_grad_(True)
    w = w.to(dtype).requires_grad_(True)
    y = F.conv2d(x, w, None, padding=1)
    y.backward(dz.to(dtype))
    return x.grad.detach(), w.grad.detach()


@pytest.mark.parametrize("H,W,cin,cout", [(20, 33, 64, 64), (37, 62, 128, 64), (9, 7, 16, 128), (75, 125, 64, 128)])
def test_conv3x3_wgrad_and_dgrad(H, W, cin, cout):
    x, w, dz = conv_case(H, W, cin, cout, H * W + cin)
    dx64, dw64 = torch_conv_grads(x, w, dz, torch.float64)
    dx32, dw32 = torch_conv_grads(x, w, dz, torch.float32)
    x_hwc = gpu(x[0].permute(1, 2, 0))
    dz_hwc = gpu(dz[0].permute(1, 2, 0))
    dwp = T.conv3x3_wgrad(x_hwc, dz_hwc, cin, cout)                               # [9][cout][cin]
    got_dw = dwp.permute(1, 2, 0).reshape(cout, cin, 3, 3).cpu().numpy()
    e, ey = err_vs_f64(got_dw, dw64.numpy(), dw32.numpy())
    assert e <= max(4 * ey, 2e-6), ("wgrad", e, ey)
    if cin % 64:
        return          # the data gradient is a forward 3x3 conv producing `cin` channels: that kernel needs multiples of 64
    # data gradient: forward kernel on the flipped / transposed pack
    wp = gpu(w.permute(2, 3, 0, 1).reshape(9, cout, cin))                         # the frcnn_pack_conv3x3 layout
    zero = torch.zeros((1024,), device=DEV)
    got_dx = T.conv3x3_dgrad(dz_hwc, wp, cin, cout, zero).permute(2, 0, 1).cpu().numpy()
    e, ey = err_vs_f64(got_dx, dx64[0].numpy(), dx32[0].numpy())
    assert e <= max(4 * ey, 2e-6), ("dgrad", e, ey)


@pytest.mark.parametrize("H,W,cin,cout", [(37, 62, 256, 256), (19, 21, 128, 256), (9, 14, 512, 256)])
def test_winograd_forward_and_data_gradient_from_the_master_pack(H, W, cin, cout):
    """f32_winograd train step: forward and data-gradient convolutions of the wide layers run as Winograd layers whose filter
    banks are rebuilt from the tap-major master pack (frcnn_pack_conv3x3_winograd_taps)."""
    x, w, dz = conv_case(H, W, cin, cout, H * W + cout)
    dx64, _ = torch_conv_grads(x, w, dz, torch.float64)
    dx32, _ = torch_conv_grads(x, w, dz, torch.float32)
    x, w, dz = x.detach(), w.detach(), dz.detach()
    wp = gpu(w.permute(2, 3, 0, 1).reshape(9, cout, cin))                         # the frcnn_pack_conv3x3 layout
    # filter banks == numpy G g G^T of the filter / of the rotated, channel-transposed filter
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
    w64 = w.double().numpy()
    u = T.winograd_bank(wp, cout, cin, fused=False).cpu().numpy().astype(np.float64)
    ref_u = np.einsum("ia,kcab,jb->ijkc", G, w64, G).reshape(16, cout, cin)
    assert np.abs(u - ref_u).max() <= 1e-7 * np.abs(ref_u).max()
    ud = T.winograd_bank(wp, cout, cin, data_gradient=True, fused=False).cpu().numpy().astype(np.float64)
    wrot = np.flip(w64, axis=(2, 3)).transpose(1, 0, 2, 3)                       # [cin][cout][3][3]
    ref_ud = np.einsum("ia,kcab,jb->ijkc", G, wrot, G).reshape(16, cin, cout)
    assert np.abs(ud - ref_ud).max() <= 1e-7 * np.abs(ref_ud).max()
    # forward (bias + ReLU) and data gradient against float64 / the direct kernels
    # (the one-launch kernel's banks, which the step uses, are permutations of these two: tests/test_winofused_gpu.py)
    assert nv.uses_winograd_fused(cin, cout) and nv.uses_winograd_fused(cout, cin)
    b = torch.randn((cout,)) * 0.1
    x_hwc, dz_hwc = gpu(x[0].permute(1, 2, 0)), gpu(dz[0].permute(1, 2, 0))
    y64 = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1))[0].permute(1, 2, 0).numpy()
    yw = T.conv3x3_forward(x_hwc, wp, gpu(b), cin, cout, True).cpu().numpy().astype(np.float64)
    yd = T.conv3x3_forward(x_hwc, wp, gpu(b), cin, cout, False).cpu().numpy().astype(np.float64)
    scale = np.abs(y64).max()
    ew, ed = np.abs(yw - y64).max() / scale, np.abs(yd - y64).max() / scale
    assert ew <= 5 * ed + 3e-6, ("forward", ew, ed)
    zero = torch.zeros((1024,), device=DEV)
    gw = T.conv3x3_dgrad(dz_hwc, wp, cin, cout, zero, winograd=True).permute(2, 0, 1).cpu().numpy()
    gd = T.conv3x3_dgrad(dz_hwc, wp, cin, cout, zero, winograd=False).permute(2, 0, 1).cpu().numpy()
    e_w, ey = err_vs_f64(gw, dx64[0].numpy(), dx32[0].numpy())
    e_d, _ = err_vs_f64(gd, dx64[0].numpy(), dx32[0].numpy())
    print("winograd dgrad %dx%d %d<-%d: err %.3g (direct %.3g, torch f32 %.3g)" % (H, W, cin, cout, e_w, e_d, ey))
    assert e_w <= max(5 * ey, 5 * e_d, 3e-6), ("dgrad", e_w, e_d, ey)


def test_pack_conv3x3_matches_torch_layout():
    """The test above builds the forward pack with torch; make sure that is what frcnn_pack_conv3x3 produces."""
    w = torch.randn((64, 16, 3, 3))
    out = torch.empty((9, 64, 16), device=DEV)
    nv.check(nv.lib().frcnn_pack_conv3x3(dptr((w)), nv.ptr(out), 64, 16, S()), "pack")
    assert torch.equal(out.cpu(), w.permute(2, 3, 0, 1).reshape(9, 64, 16))


# ---- elementwise backward pieces --------------------------------------------------------------------------
def test_relu_backward_and_add():
    g = torch.Generator().manual_seed(3)
    for n in (1, 5, 4096, 100003):
        y = torch.randn((n,), generator=g).clamp_min(0)
        dy = torch.randn((n,), generator=g)
        d = gpu(dy.clone())
        T.relu_backward(d, gpu(y))
        assert torch.equal(d.cpu(), torch.where(y > 0, dy, torch.zeros(())))
        a, b = torch.randn((n,), generator=g), torch.randn((n,), generator=g)
        da = gpu(a)
        nv.check(nv.lib().frcnn_add_inplace(nv.ptr(da), dptr((b)), n, S()), "add")
        assert torch.equal(da.cpu(), a + b)


@pytest.mark.parametrize("H,W,c", [(8, 8, 64), (75, 125, 16), (9, 7, 4)])
def test_maxpool2x2_backward(H, W, c):
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn((1, c, H, W), generator=g)
    x[0, :, 0:2, 0:2] = 1.5                    # a tie in the first window: gradient goes to the first element
    x[0, :, 2:4, 2:4] = 0.0
    xr = x.clone().requires_grad_(True)
    y = F.max_pool2d(xr, 2, 2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    got = T.maxpool2x2_backward(gpu(x[0].permute(1, 2, 0)), gpu(dy[0].permute(1, 2, 0))).permute(2, 0, 1).cpu()
    assert torch.equal(got, xr.grad[0])


def test_transpose_and_gather_rows():
    g = torch.Generator().manual_seed(1)
    for rows, cols, ldi in ((128, 128, 128), (37, 101, 104), (2294, 45, 128), (1, 7, 8)):
        x = torch.randn((rows, ldi), generator=g)
        y, ldo = T.transpose(gpu(x), rows, cols, ldi)
        assert ldo % 4 == 0 and ldo >= rows
        want = torch.zeros((cols, ldo))
        want[:, :rows] = x[:, :cols].T
        assert torch.equal(y.cpu(), want)
    src = torch.randn((50, 84), generator=g)
    idx = torch.tensor([3, 3, 49, 0, 17], dtype=torch.int32)
    dst = torch.empty((5, 84), device=DEV)
    nv.check(nv.lib().frcnn_gather_rows(dptr((src)), dptr((idx)), 5, 84, nv.ptr(dst), S()), "gather")
    assert torch.equal(dst.cpu(), src[idx.long()])


def test_sgd_step_matches_torch_optim():
    g = torch.Generator().manual_seed(9)
    w0 = torch.randn((1000,), generator=g)
    p = torch.nn.Parameter(w0.clone())
    opt = torch.optim.SGD([{"params": [p], "weight_decay": 5e-4}], lr=1e-3, momentum=0.9)
    w = gpu(w0.clone())
    buf = torch.empty_like(w)
    for step in range(3):
        grad = torch.randn((1000,), generator=g)
        p.grad = grad.clone()
        opt.step()
        nv.check(nv.lib().frcnn_sgd_step(nv.ptr(w), dptr((grad)), nv.ptr(buf), 1000, 1e-3, 0.9, 5e-4, 1 if step == 0 else 0, S()),
                 "sgd")
        assert float((w.cpu() - p.detach()).abs().max()) <= 2e-7 * float(p.detach().abs().max())


# ---- RoI pool backward ------------------------------------------------------------------------------------
def test_roi_pool_backward_matches_oracle():
    g = torch.Generator().manual_seed(11)
    fh, fw, c = 20, 32, 64
    fm = torch.randn((1, c, fh, fw), generator=g)
    fm[0, :, 3:6, 4:8] = 0.25                                                   # ties inside some bins
    props = torch.tensor([[0, 0, 319, 511], [40, 60, 200, 300], [100, 100, 101, 101], [10, 400, 330, 516],
                          [-20, -20, 50, 50], [48, 48, 112, 176]], dtype=torch.float32)       # (y1, x1, y2, x2)
    fmr = fm.clone().requires_grad_(True)
    pooled = TO.roi_pool_autograd(fmr, props)
    dout = torch.randn(pooled.shape, generator=g)
    pooled.backward(dout)
    want = fmr.grad[0].permute(1, 2, 0).contiguous()
    n = props.shape[0]
    lib = nv.lib()
    dfm = torch.full((fh, fw, c), float("nan"), device=DEV)
    wsb = int(lib.frcnn_roi_pool_backward_workspace_bytes(n, 7, c))
    ws = torch.empty((wsb // 4,), device=DEV)
    fm_hwc = gpu(fm[0].permute(1, 2, 0))
    d_dout = gpu(dout.permute(0, 2, 3, 1))
    nv.check(lib.frcnn_roi_pool_backward(nv.ptr(fm_hwc), fh, fw, c, dptr((props)), n, 7, 1.0 / 16.0, nv.ptr(d_dout),
                                         nv.ptr(dfm), 0, nv.ptr(ws), wsb, S()), "roi_pool_backward")
    got = dfm.cpu()
    # same addends, possibly different order of the float32 sums
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    assert torch.equal(got == 0, want == 0)
    base = torch.randn((fh, fw, c), generator=g)
    acc = gpu(base.clone())
    nv.check(lib.frcnn_roi_pool_backward(nv.ptr(fm_hwc), fh, fw, c, dptr((props)), n, 7, 1.0 / 16.0, nv.ptr(d_dout),
                                         nv.ptr(acc), 1, nv.ptr(ws), wsb, S()), "roi_pool_backward")
    assert float((acc.cpu() - (base + got)).abs().max()) <= 1e-6 * float(want.abs().max())


# ---- labelling ----------------------------------------------------------------------------------------------
def label_on_gpu(props, n_valid, gt, gt_cls, ncls, bg_thr, obj_thr):
    cap = props.shape[0] + gt.shape[0]
    nd = 4 * (ncls - 1)
    out_p = torch.empty((cap, 4), device=DEV)
    out_c = torch.empty((cap,), dtype=torch.int32, device=DEV)
    out_o = torch.empty((cap, ncls), device=DEV)
    out_d = torch.empty((cap, 2, nd), device=DEV)
    cnt = torch.zeros((1,), dtype=torch.int32, device=DEV)
    means = (C.c_float * 4)(0, 0, 0, 0)
    stds = (C.c_float * 4)(0.1, 0.1, 0.2, 0.2)
    nv.check(nv.lib().frcnn_label_proposals(dptr((props)), dptr((torch.tensor([n_valid], dtype=torch.int32))),
                                            props.shape[0], dptr((gt)), dptr((gt_cls.to(torch.int32))), gt.shape[0],
                                            ncls, bg_thr, obj_thr, means, stds, nv.ptr(out_p), nv.ptr(out_c), nv.ptr(out_o),
                                            nv.ptr(out_d), nv.ptr(cnt), S()), "label")
    k = int(cnt.item())
    return out_p[:k].cpu(), out_c[:k].cpu(), out_o[:k].cpu(), out_d[:k].cpu()


@pytest.mark.parametrize("n,bg_thr", [(2000, 0.0), (1500, 0.1), (37, 0.0), (0, 0.0)])
def test_label_proposals_matches_oracle(n, bg_thr):
    rng = np.random.RandomState(n + 5)
    gt = torch.tensor([[100, 200, 400, 700], [50, 50, 300, 180], [300, 600, 580, 990]], dtype=torch.float32)
    gt_cls = torch.tensor([7, 15, 3])
    y1 = rng.uniform(0, 500, n); x1 = rng.uniform(0, 900, n)
    props = np.stack([y1, x1, y1 + rng.uniform(16, 300, n), x1 + rng.uniform(16, 400, n)], axis=1).astype(np.float32)
    if n >= 37:
        props[5] = gt[0].numpy() + np.float32(3)          # a few sure positives
        props[6] = gt[1].numpy()
        props[7] = [0, 0, 16, 16]
    props_t = torch.from_numpy(props).reshape(-1, 4)
    buf = torch.cat([props_t, torch.full((13, 4), 7.0)])  # rows beyond n_valid must be ignored
    p, c, o, d = label_on_gpu(buf, n, gt, gt_cls, 21, bg_thr, 0.5)
    rp, ro, rd = TO.label_proposals(props_t, gt, gt_cls, 21, bg_thr, 0.5)
    assert p.shape == rp.shape
    assert torch.equal(p, rp)                             # selection and order exact
    assert torch.equal(o, ro)
    assert torch.equal(c.long(), ro.argmax(dim=1))
    assert torch.equal(d[:, 0, :], rd[:, 0, :])           # masks exact
    # targets: (ty, tx) exact float32 arithmetic; (th, tw) within 1 ulp of torch's log
    assert torch.equal(d[:, 1, 0::4], rd[:, 1, 0::4]) and torch.equal(d[:, 1, 1::4], rd[:, 1, 1::4])
    assert float((d[:, 1, :] - rd[:, 1, :]).abs().max()) <= 2e-6 * max(1.0, float(rd[:, 1, :].abs().max())) if p.shape[0] else True


# ---- losses -------------------------------------------------------------------------------------------------
def test_rpn_loss_and_gradient():
    g = torch.Generator().manual_seed(4)
    fh, fw = 22, 30
    P = fh * fw
    head = torch.zeros((P, 128))
    head[:, 0:9] = torch.randn((P, 9), generator=g) * 3
    head[:, 9:45] = torch.randn((P, 36), generator=g)
    head[0, 0] = 120.0; head[1, 1] = -120.0            # saturated sigmoid: exercises the BCE clamps
    rpn_map = torch.zeros((1, fh, fw, 9, 6))
    rpn_map[..., 1] = (torch.rand((1, fh, fw, 9), generator=g) < 0.3).float()
    rpn_map[..., 2:6] = torch.randn((1, fh, fw, 9, 4), generator=g) * 0.5
    flat = torch.randperm(P * 9, generator=g)[:254]
    flat = torch.cat([flat[(flat != 0) & (flat != 10)], torch.tensor([0, 10])])     # include the saturated anchors
    rpn_map[..., 0].view(-1)[flat] = 1.0
    hr = head.clone().requires_grad_(True)
    scores = torch.sigmoid(hr[:, 0:9]).reshape(1, fh, fw, 9)
    deltas = hr[:, 9:45].reshape(1, fh, fw, 36)
    lc = TO.rpn_class_loss(scores, rpn_map)
    lr_ = TO.rpn_regression_loss(deltas, rpn_map)
    (lc + lr_).backward()
    losses = torch.zeros((2,), device=DEV)
    dhead = torch.full((P, 128), float("nan"), device=DEV)
    nv.check(nv.lib().frcnn_rpn_loss(dptr((head)), 128, P, dptr((flat.to(torch.int32))), int(flat.shape[0]),
                                     dptr((rpn_map.reshape(-1, 6))), nv.ptr(losses), nv.ptr(dhead), S()), "rpn_loss")
    got = losses.cpu()
    assert abs(float(got[0]) - float(lc)) <= 1e-5 * abs(float(lc))
    assert abs(float(got[1]) - float(lr_)) <= 1e-5 * abs(float(lr_))
    gd = dhead.cpu()
    assert float((gd - hr.grad).abs().max()) <= 1e-5 * float(hr.grad.abs().max())
    assert torch.equal(gd == 0, hr.grad == 0) or float((gd - hr.grad).abs().max()) <= 1e-9


def test_detector_loss_and_gradient():
    g = torch.Generator().manual_seed(6)
    n, ncls = 128, 21
    nd = 4 * (ncls - 1)
    logits = torch.randn((n, ncls), generator=g) * 3
    logits[0, 3] = 60.0                                   # a saturated row
    deltas = torch.randn((n, nd), generator=g)
    cls = torch.randint(0, ncls, (n,), generator=g)
    cls[0] = 5
    onehot = F.one_hot(cls, ncls).float()
    gtd = torch.zeros((n, 2, nd))
    gtd[:, 0, :] = torch.repeat_interleave(onehot, 4, dim=1)[:, 4:]
    gtd[:, 1, :] = torch.randn((n, nd), generator=g) * 1.5
    lg = logits.clone().requires_grad_(True)
    dl = deltas.clone().requires_grad_(True)
    classes = F.softmax(lg, dim=1)
    l1 = TO.detector_class_loss(classes, onehot)
    l2 = TO.detector_regression_loss(dl, gtd)
    (l1 + l2).backward()
    losses = torch.zeros((2,), device=DEV)
    dlog = torch.full((n, 128), float("nan"), device=DEV)
    nv.check(nv.lib().frcnn_detector_loss(dptr((classes.detach())), dptr((deltas)), dptr((onehot)), dptr((gtd)),
                                          n, ncls, nv.ptr(losses), nv.ptr(dlog), 128, S()), "detector_loss")
    got = losses.cpu()
    assert abs(float(got[0]) - float(l1)) <= 1e-5 * abs(float(l1))
    assert abs(float(got[1]) - float(l2)) <= 1e-5 * abs(float(l2))
    gd = dlog.cpu()
    assert float((gd[:, :ncls] - lg.grad).abs().max()) <= 2e-5 * float(lg.grad.abs().max())
    assert float((gd[:, ncls:ncls + nd] - dl.grad).abs().max()) <= 1e-6 * float(dl.grad.abs().max())
    assert float(gd[:, ncls + nd:].abs().max()) == 0.0


# ---- the whole step against the reference-derived fixtures ------------------------------------------------------
def sample_positions(n, count=2048):
    return np.unique(np.linspace(0, n - 1, min(count, n)).astype(np.int64))


_KEY_TO_PACKED = {
    "_stage2_region_proposal_network._rpn_conv1.weight": "rpn_conv",
    "_stage3_detector_network._pool_to_feature_vector._fc1.weight": "fc1",
    "_stage3_detector_network._pool_to_feature_vector._fc2.weight": "fc2",
}


def canonical_grads(packed, ncls=21):
    """packed-layout gradients of training.train_step -> the reference's parameter layouts / key names."""
    out = {}
    names = [n for n, _, _, _ in __import__("fasterrcnn_amd.models.vgg16", fromlist=["_LAYERS"])._LAYERS]
    for i in range(4, 13):
        g = packed["conv%d" % i]
        out["_stage1_feature_extractor.%s.weight" % names[i]] = g.permute(1, 2, 0).reshape(g.shape[1], g.shape[2], 3, 3)
    g = packed["rpn_conv"]
    out["_stage2_region_proposal_network._rpn_conv1.weight"] = g.permute(1, 2, 0).reshape(512, 512, 3, 3)
    out["_stage2_region_proposal_network._rpn_class.weight"] = packed["rpn_head"][0:9].reshape(9, 512, 1, 1)
    out["_stage2_region_proposal_network._rpn_boxes.weight"] = packed["rpn_head"][9:45].reshape(36, 512, 1, 1)
    out["_stage3_detector_network._pool_to_feature_vector._fc1.weight"] = packed["fc1"].reshape(4096, 49, 512).permute(0, 2, 1).reshape(4096, 25088)
    out["_stage3_detector_network._pool_to_feature_vector._fc2.weight"] = packed["fc2"]
    out["_stage3_detector_network._classifier.weight"] = packed["head"][0:ncls]
    out["_stage3_detector_network._regressor.weight"] = packed["head"][ncls:ncls + 4 * (ncls - 1)]
    return out


def canonical_grads_resnet(packed, ncls=21):
    """packed-layout gradients of the ResNet train step -> the reference's parameter layouts / key names."""
    out = {}
    fe = "_stage1_feature_extractor._feature_extractor."
    l4 = "_stage3_detector_network._pool_to_feature_vector._layer4."
    for name, g in packed.items():
        if not name.startswith("layer"):
            continue
        layer, blk, conv = name.split(".")
        prefix = {"layer2": fe + "5.", "layer3": fe + "6.", "layer4": l4}[layer] + blk + "."
        key = prefix + ("downsample.0.weight" if conv == "downsample" else conv + ".weight")
        k = int(round(g.shape[0] ** 0.5))
        out[key] = g.permute(1, 2, 0).reshape(g.shape[1], g.shape[2], k, k)
    out["_stage2_region_proposal_network._rpn_conv1.weight"] = packed["rpn_conv"].permute(1, 2, 0).reshape(1024, 1024, 3, 3)
    out["_stage2_region_proposal_network._rpn_class.weight"] = packed["rpn_head"][0:9].reshape(9, 1024, 1, 1)
    out["_stage2_region_proposal_network._rpn_boxes.weight"] = packed["rpn_head"][9:45].reshape(36, 1024, 1, 1)
    out["_stage3_detector_network._classifier.weight"] = packed["head"][0:ncls]
    out["_stage3_detector_network._regressor.weight"] = packed["head"][ncls:ncls + 4 * (ncls - 1)]
    return out


@pytest.mark.parametrize("backbone,tag,math_mode", [("vgg16", "352x480_s4", "f32_winograd"), ("vgg16", "416x544_s6", "f32_winograd"),
                                                    ("vgg16", "352x480_s4", "f32"), ("resnet50", "352x480_s4", "f32_winograd"),
                                                    ("resnet101", "320x416_s6", "f32_winograd"), ("resnet50", "352x480_s4", "f32")])
def test_train_step_matches_reference_fixture(backbone, tag, math_mode, golden_dir, sd_cpu):
    from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
    from fasterrcnn_amd.models.vgg16 import VGG16Backbone
    from fasterrcnn_amd.models import resnet
    gold = np.load(os.path.join(golden_dir, "train_%s_%s.npz" % (backbone, tag)))
    seed, h, w = int(gold["seed"]), int(gold["height"]), int(gold["width"])
    if backbone == "vgg16":
        model = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
        model.load_state_dict(sd_cpu, strict=True)
        img = synthetic.image(seed, h, w).unsqueeze(0).cuda()
        fshape = (512, h // 16, w // 16)
        canon = canonical_grads
    else:
        arch = {"resnet50": "ResNet50", "resnet101": "ResNet101"}[backbone]
        model = FasterRCNNModel(num_classes=21, backbone=resnet.ResNetBackbone(getattr(resnet.Architecture, arch)))
        model.load_state_dict(synthetic.resnet_state_dict(1234, arch), strict=True)
        img = synthetic.image_rgb(seed, h, w).unsqueeze(0).cuda()
        fshape = (1024, -(-h // 16), -(-w // 16))
        canon = canonical_grads_resnet
    model = model.cuda()
    assert model.math_mode == "f32_winograd"          # the default; in it the wide forward / data-gradient convolutions are Winograd layers
    model.math_mode = math_mode
    gts = synthetic.ground_truth(seed, h, w)
    boxes = [Box(class_index=c, class_name="x", corners=k) for c, k in gts]
    am, vm = O.generate_anchor_maps((3, h, w), fshape, 16)
    rmap, obj, bg = O.generate_rpn_map(am, vm, np.stack([k for _, k in gts]))
    opt = T.create_optimizer(model, learning_rate=float(gold["lr"]), momentum=float(gold["momentum"]),
                             weight_decay=float(gold["weight_decay"]))
    keys = [str(k) for k in gold["train_keys"]]
    n_samp = int(gold["sample_count"]) if "sample_count" in gold.files else 2048
    before = {k: v.clone() for k, v in model.state_dict().items()}
    random.seed(int(gold["rng_seed"])); torch.manual_seed(int(gold["rng_seed"]))
    lr, mom, wd = float(gold["lr"]), float(gold["momentum"]), float(gold["weight_decay"])
    st = model._training_state()
    bufs = {}
    for step in range(int(gold["steps"])):
        pre = "s%d_" % step
        detail = {}
        w_before = {k: v.double().clone() for k, v in st.trainable().items()}
        loss = T.train_step(model, opt, img, am, vm, torch.from_numpy(rmap).unsqueeze(0), [obj], [bg], [boxes], detail=detail)
        # --- SGD bookkeeping is exact with respect to THIS run's gradients (independent of any selection flips):
        #     g' = g + wd w; buf = g' (first step) | momentum buf + g'; w -= lr buf        (torch.optim.SGD)
        for k, w0 in w_before.items():
            gp = detail["grads"][k].double() + wd * w0
            bufs[k] = gp if step == 0 else mom * bufs[k] + gp
            want_w = w0 - lr * bufs[k]
            got_w = st.trainable()[k].double()
            assert float((got_w - want_w).abs().max()) <= 1.2e-7 * float(want_w.abs().max()), (step, k)
            # the float32 momentum buffer follows the same recurrence
            assert float((st.momentum[k].double() - bufs[k]).abs().max()) <= 3e-7 * float(bufs[k].abs().max()), (step, k)
            bufs[k] = st.momentum[k].double().clone()
        # --- discrete selections: identical to the reference's under the same seeds.  The anchor mini-batch depends
        #     only on the RNG; the proposal batch also on the RPN output, which from step 1 on is computed with weights
        #     that already differ from the reference's by the flip-level noise described below.
        assert np.array_equal(detail["rpn_sample"].cpu().numpy(), gold[pre + "rpn_sample_flat"])
        n_props = int(detail["counts"][2].item())
        same_selection = (n_props == int(gold[pre + "n_rpn_proposals"])
                          and detail["labelled"][0].shape[0] == int(gold[pre + "n_labelled"])
                          and np.array_equal(detail["sample_idx"].numpy().astype(np.int32), gold[pre + "proposal_sample_indices"])
                          and np.array_equal(detail["sampled_onehot"].cpu().numpy().argmax(axis=1).astype(np.int32),
                                             gold[pre + "sampled_class_idx"]))
        want = gold[pre + "losses"]
        got = np.array([loss.rpn_class, loss.rpn_regression, loss.detector_class, loss.detector_regression, loss.total])
        if step == 0:
            assert same_selection, "step 0 runs on identical weights: proposals and samples must match the reference run"
        if not same_selection:
            # an RPN rank / NMS decision flipped: another (equally valid) proposal batch was drawn.  Only coarse agreement
            # with the fixture is meaningful for this step; the RPN losses do not depend on the proposal batch.
            print("step %d: proposal batch differs from the reference run (flip-level weight differences)" % step)
            assert np.all(np.abs(got[:2] - want[:2]) <= 1e-3 * np.abs(want[:2])), (got, want)
            assert np.all(np.abs(got - want) <= 0.25 * np.abs(want) + 0.05), (got, want)
            before = {k: v.clone() for k, v in model.state_dict().items()}
            continue
        sp = detail["sampled_props"].cpu().numpy()
        # north_star: boxes within 1e-3 px of the reference (on identical weights, i.e. step 0); ResNet-101's 91 convolutions
        # ahead of the RPN accumulate a little more float32 difference (1.2e-3 px measured)
        box_tol = (1e-3 if backbone != "resnet101" else 5e-3) if step == 0 else 2e-2
        assert np.abs(sp - gold[pre + "sampled_props"]).max() <= box_tol
        # --- losses
        assert np.all(np.abs(got - want) <= (2e-5 if step == 0 else 2e-4) * np.abs(want) + 1e-7), (got, want)
        # --- gradients (sampled entries + norms), in the reference's layouts.
        # Two float32 implementations of the forward differ by ~1e-6 relative, which flips a handful of the
        # ~4e7 ReLU / max-pool / RoI-argmax decisions that sit within that distance of a tie (measured with
        # tests/train_parity_report.py: e.g. 1 of 524288 fc1 activations); each flip moves a few gradient entries
        # by up to ~1e-3 of the tensor's largest entry while the bulk agrees to ~1e-6.  Hence: a tight bound on
        # the MEDIAN error, looser bounds on the L2 error and the norm.  The backward operators themselves are
        # held to float32 accuracy on identical inputs by the per-operator tests above.
        grads = canon(detail["grads"])
        gscale = max(float(gold[pre + "gnorm/" + k]) for k in keys)
        # ResNet-101: 100 ReLU layers -> proportionally more near-tie flips than the 13-layer / 50-layer nets
        # (median bound: the largest value seen over the fixtures and both math modes is 5.7e-5, conv3_1 of the 416x544 case in
        #  the f32_winograd mode -- the deepest tensor of the backward pass collects every flip above it)
        med_tol, l2_tol, norm_tol = (1e-4, 1e-2, 5e-3) if backbone != "resnet101" else (3e-4, 5e-2, 2e-2)
        worst = {"median": (0.0, ""), "L2": (0.0, ""), "norm": (0.0, "")}
        for k in keys:
            gk = grads[k].reshape(-1)
            pos = torch.from_numpy(sample_positions(gk.shape[0], n_samp)).to(DEV)
            got_s = gk[pos].cpu().numpy().astype(np.float64)
            want_s = gold[pre + "gsample/" + k].astype(np.float64)
            wn = float(gold[pre + "gnorm/" + k])
            ref_max = max(float(np.abs(want_s).max()), 1e-7 * gscale)
            for name, val in (("median", float(np.median(np.abs(got_s - want_s))) / ref_max),
                              ("L2", float(np.linalg.norm(got_s - want_s)) / max(float(np.linalg.norm(want_s)), 1e-7 * gscale)),
                              ("norm", abs(float(gk.double().norm()) - wn) / max(wn, 1e-7 * gscale))):
                if val > worst[name][0]:
                    worst[name] = (val, k.split(".")[-2])
        print("train %s %s %s step %d: worst gradient errors %s" % (backbone, tag, math_mode, step,
              ", ".join("%s %.2e (%s)" % (n, v[0], v[1]) for n, v in worst.items())))
        for k in keys:
            gk = grads[k].reshape(-1)
            pos = torch.from_numpy(sample_positions(gk.shape[0], n_samp)).to(DEV)
            got_s = gk[pos].cpu().numpy().astype(np.float64)
            want_s = gold[pre + "gsample/" + k].astype(np.float64)
            wn = float(gold[pre + "gnorm/" + k])
            gn = float(gk.double().norm())
            assert abs(gn - wn) <= norm_tol * wn + 1e-7 * gscale, (k, gn, wn)
            ref_max = max(float(np.abs(want_s).max()), 1e-7 * gscale)
            d = np.abs(got_s - want_s)
            assert np.median(d) <= med_tol * ref_max, (k, "median", float(np.median(d)), ref_max)
            assert np.linalg.norm(got_s - want_s) <= l2_tol * max(np.linalg.norm(want_s), 1e-7 * gscale), (k, "L2")
        # --- weight update (same criteria; the update is lr x (g + wd w) with momentum from step 1 on)
        after = model.state_dict()
        for k in keys:
            dw = (after[k].double() - before[k].double()).reshape(-1)
            pos = torch.from_numpy(sample_positions(dw.shape[0], n_samp)).to(DEV)
            got_s = dw[pos].cpu().numpy()
            want_s = gold[pre + "dwsample/" + k].astype(np.float64)
            wn = float(gold[pre + "dwnorm/" + k])
            # |update| ~ 1e-6 x a weight of ~1e-2: float32 rounding of the stored weight is the noise floor
            floor = 6e-8 * float(after[k].abs().max())
            d = np.abs(got_s - want_s)
            assert np.median(d) <= med_tol * float(np.abs(want_s).max()) + 2 * floor, (k, "dw median")
            assert np.linalg.norm(got_s - want_s) <= l2_tol * np.linalg.norm(want_s) + 2 * floor * len(d) ** 0.5, (k, "dw L2")
            assert abs(float(dw.norm()) - wn) <= norm_tol * wn + floor * dw.shape[0] ** 0.5, (k, "dw norm")
        for k in before:
            if k not in keys:
                assert torch.equal(after[k], before[k]), "frozen parameter / bias changed: %s" % k
        before = {k: v.clone() for k, v in after.items()}
    # the trained model still predicts (weights are re-packed from the synced parameters)
    det = model.predict(img, score_threshold=0.05)
    assert sorted(det.keys()) == list(range(1, 21))


def test_train_step_full_size_is_deterministic_and_learns(sd_cpu):
    """BASELINE's 600x1000 size: three steps on one sample, run twice from the same seeds -> bit-identical losses and weights
    (split reductions and the RoI-pool backward are fixed-order), finite, and the loss falls."""
    from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
    from fasterrcnn_amd.models.vgg16 import VGG16Backbone
    h, w, seed = 600, 1000, 2
    img = synthetic.image(seed, h, w).unsqueeze(0).cuda()
    gts = synthetic.ground_truth(seed, h, w)
    boxes = [Box(class_index=c, class_name="x", corners=k) for c, k in gts]
    am, vm = O.generate_anchor_maps((3, h, w), (512, h // 16, w // 16), 16)
    rmap, obj, bg = O.generate_rpn_map(am, vm, np.stack([k for _, k in gts]))
    rmap_t = torch.from_numpy(rmap).unsqueeze(0).cuda()
    runs = []
    for _ in range(2):
        model = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
        model.load_state_dict(sd_cpu, strict=True)
        model = model.cuda()
        opt = T.create_optimizer(model, learning_rate=1e-6)
        random.seed(5); torch.manual_seed(5)
        losses = [model.train_step(opt, img, am, vm, rmap_t, [obj], [bg], [boxes]) for _ in range(3)]
        sd = model.state_dict()
        runs.append((losses, {k: v.clone() for k, v in sd.items()}))
    (l0, s0), (l1, s1) = runs
    assert [x.total for x in l0] == [x.total for x in l1]
    assert all(np.isfinite([x.rpn_class, x.rpn_regression, x.detector_class, x.detector_regression, x.total]).all() for x in l0)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
    assert l0[-1].total < l0[0].total
    changed = [k for k in s0 if not torch.equal(s0[k].cpu(), sd_cpu[k])]
    assert len(changed) == 16 and all("weight" in k for k in changed)        # 9 convs + 3 RPN + fc1 fc2 + 2 heads; no bias


@pytest.mark.parametrize("roi_pooling", ["pool", "align"])
def test_resnet101_train_step_full_size_is_deterministic_and_learns(roi_pooling):
    """BASELINE configs[4]'s model at its size: ResNet-101, 600x1000, batch 1 (fp32; RoIPool as the reference trains, and RoIAlign):
    four steps run twice from the same seeds -> bit-identical losses and weights, finite; the loss falls; exactly the conv weights
    of layer2-4 + RPN + heads move (every BatchNorm and conv1 / layer1 frozen: resnet.py:48-55,86,123)."""
    from fasterrcnn_amd.models import resnet
    from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
    h, w, seed = 600, 1000, 2
    sd0 = synthetic.resnet_state_dict(1234, "ResNet101")
    img = synthetic.image_rgb(seed, h, w).unsqueeze(0).cuda()
    gts = synthetic.ground_truth(seed, h, w)
    boxes = [Box(class_index=c, class_name="x", corners=k) for c, k in gts]
    am, vm = O.generate_anchor_maps((3, h, w), (1024, -(-h // 16), -(-w // 16)), 16)
    rmap, obj, bg = O.generate_rpn_map(am, vm, np.stack([k for _, k in gts]))
    rmap_t = torch.from_numpy(rmap).unsqueeze(0).cuda()
    runs = []
    for _ in range(2):
        model = FasterRCNNModel(num_classes=21, backbone=resnet.ResNetBackbone(resnet.Architecture.ResNet101), roi_pooling=roi_pooling)
        model.load_state_dict(sd0, strict=True)
        model = model.cuda()
        opt = T.create_optimizer(model, learning_rate=3e-6)
        random.seed(5); torch.manual_seed(5)
        losses = [model.train_step(opt, img, am, vm, rmap_t, [obj], [bg], [boxes]) for _ in range(4)]
        runs.append((losses, {k: v.clone() for k, v in model.state_dict().items()}))
    (l0, s0), (l1, s1) = runs
    assert [x.total for x in l0] == [x.total for x in l1]
    assert all(np.isfinite([x.rpn_class, x.rpn_regression, x.detector_class, x.detector_regression, x.total]).all() for x in l0)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
    print("ResNet-101 600x1000 %s: total loss %s" % (roi_pooling, ["%.4f" % x.total for x in l0]))
    assert l0[-1].total < l0[0].total
    changed = sorted(k for k in s0 if not torch.equal(s0[k].cpu(), sd0[k]))
    from oracle import train_oracle as TO
    assert changed == sorted(TO.trainable_weight_keys(sd0)) and len(changed) == 98


# ---- general conv backward (ResNet bottlenecks) -------------------------------------------------------------------
@pytest.mark.parametrize("N,H,W,cin,cout,k,stride,pad", [(1, 20, 33, 64, 128, 1, 1, 0), (1, 21, 33, 64, 256, 1, 2, 0),
                                                         (1, 20, 33, 64, 64, 3, 2, 1), (5, 7, 7, 128, 64, 3, 2, 1),
                                                         (3, 4, 4, 64, 64, 3, 1, 1), (1, 38, 63, 256, 64, 3, 1, 1),
                                                         (128, 4, 4, 64, 256, 1, 1, 0)])
def test_conv_wgrad_and_dgrad_general(N, H, W, cin, cout, k, stride, pad):
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + k + stride)
    x = torch.randn((N, cin, H, W), generator=g)
    w = torch.randn((cout, cin, k, k), generator=g) * (2.0 / (k * k * cin)) ** 0.5
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dz = torch.randn((N, cout, Ho, Wo), generator=g)
    res = torch.randn((N, cin, H, W), generator=g)

    def grads(dtype):
        xx = x.to(dtype).requires_grad_(True)
        ww = w.to(dtype).requires_grad_(True)
        F.conv2d(xx, ww, None, stride=stride, padding=pad).backward(dz.to(dtype))
        return xx.grad.detach(), ww.grad.detach()
    dx64, dw64 = grads(torch.float64)
    dx32, dw32 = grads(torch.float32)
    lib = nv.lib()
    x_n = gpu(x.permute(0, 2, 3, 1))
    dz_n = gpu(dz.permute(0, 2, 3, 1))
    dwp = torch.full((k * k, cout, cin), float("nan"), device=DEV)
    wsb = int(lib.frcnn_conv_wgrad_workspace_bytes(N, H, W, cin, cout, k, stride, pad))
    ws = torch.empty((wsb // 4 + 1,), device=DEV)
    nv.check(lib.frcnn_conv_wgrad(nv.ptr(x_n), nv.ptr(dz_n), nv.ptr(dwp), N, H, W, cin, cout, k, stride, pad, nv.ptr(ws), wsb, S()),
             "conv_wgrad")
    got_dw = dwp.permute(1, 2, 0).reshape(cout, cin, k, k).cpu().numpy()
    e, ey = err_vs_f64(got_dw, dw64.numpy(), dw32.numpy())
    assert e <= max(4 * ey, 2e-6), ("wgrad", e, ey)
    # data gradient (+ fused residual)
    wp = gpu(w.permute(2, 3, 0, 1).reshape(k * k, cout, cin))
    wd = torch.empty((k * k, cin, cout), device=DEV)
    nv.check(lib.frcnn_pack_conv_dgrad(nv.ptr(wp), nv.ptr(wd), k * k, cout, cin, S()), "pack_conv_dgrad")
    assert torch.equal(wd.cpu(), wp.cpu().permute(0, 2, 1))
    res_n = gpu(res.permute(0, 2, 3, 1))
    dx = torch.full((N, H, W, cin), float("nan"), device=DEV)
    wsb = int(lib.frcnn_conv_dgrad_workspace_bytes(N, H, W, cin, cout, k, stride, pad))
    ws = torch.empty((wsb // 4 + 1,), device=DEV)
    for residual in (None, res_n):
        nv.check(lib.frcnn_conv_dgrad(nv.ptr(dz_n), nv.ptr(wd), nv.ptr(residual), nv.ptr(dx), N, H, W, cin, cout, k, stride, pad,
                                      nv.ptr(ws), wsb, S()), "conv_dgrad")
        want64 = dx64 + (res.double() if residual is not None else 0)
        want32 = dx32 + (res if residual is not None else 0)
        e, ey = err_vs_f64(dx.permute(0, 3, 1, 2).cpu().numpy(), want64.numpy(), want32.numpy())
        assert e <= max(4 * ey, 2e-6), ("dgrad", residual is not None, e, ey)


def test_scale_rows_bn_affine_and_mean_backward():
    g = torch.Generator().manual_seed(8)
    lib = nv.lib()
    taps, cout, cin = 9, 64, 32
    src = torch.randn((taps, cout, cin), generator=g)
    sc = torch.rand((cout,), generator=g) + 0.5
    d_src, d_sc = gpu(src), gpu(sc)
    dst = torch.empty_like(d_src)
    nv.check(lib.frcnn_scale_rows(nv.ptr(d_src), nv.ptr(d_sc), nv.ptr(dst), taps, cout, cin, S()), "scale_rows")
    assert torch.equal(dst.cpu(), src * sc.reshape(1, cout, 1))
    gamma, beta = torch.rand((300,), generator=g) + 0.5, torch.randn((300,), generator=g)
    mean, var = torch.randn((300,), generator=g), torch.rand((300,), generator=g) + 0.5
    dg, db, dm, dv = gpu(gamma), gpu(beta), gpu(mean), gpu(var)
    scale, shift = torch.empty((300,), device=DEV), torch.empty((300,), device=DEV)
    nv.check(lib.frcnn_bn_scale_shift(nv.ptr(dg), nv.ptr(db), nv.ptr(dm), nv.ptr(dv), 1e-5, 300, nv.ptr(scale), nv.ptr(shift), S()),
             "bn_scale_shift")
    want_scale = gamma / torch.sqrt(var + 1e-5)
    assert float((scale.cpu() - want_scale).abs().max()) <= 1.2e-7 * float(want_scale.abs().max())
    assert float((shift.cpu() - (beta - mean * want_scale)).abs().max()) <= 1e-6
    x = torch.randn((6, 32, 4, 4), generator=g, requires_grad=True)
    y = x.mean(-1).mean(-1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    d_dy = gpu(dy)
    dx = torch.empty((6, 4, 4, 32), device=DEV)
    nv.check(lib.frcnn_spatial_mean_backward(nv.ptr(d_dy), nv.ptr(dx), 6, 4, 4, 32, S()), "spatial_mean_backward")
    assert torch.equal(dx.permute(0, 3, 1, 2).cpu(), x.grad)


@pytest.mark.parametrize("arch", ["vgg16", "resnet50"])
def test_data_parallel_gradient_exchange_single_rank(sd_cpu, arch):
    """The RCCL exchange of training.GradientAverager (asynchronous all-reduces started during the backward pass, completed before
    the SGD update) on a one-rank group must leave the step bit-identical (average of one); the world-2 arithmetic is covered on
    CPU by tests/test_distributed_gloo.py.  resnet50: the bottlenecks' weight gradients come off the SECOND stream (training._SideGrads,
    frcnn_bottleneck_backward) and reach the exchange behind their events, in the one-stream order."""
    import socket
    import torch.distributed as dist
    from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
    from fasterrcnn_amd.models.vgg16 import VGG16Backbone
    h, w, seed = 352, 480, 4
    vgg = arch == "vgg16"
    img = (synthetic.image if vgg else synthetic.image_rgb)(seed, h, w).unsqueeze(0).cuda()
    gts = synthetic.ground_truth(seed, h, w)
    boxes = [Box(class_index=c, class_name="x", corners=k) for c, k in gts]
    am, vm = O.generate_anchor_maps((3, h, w), (512, h // 16, w // 16) if vgg else (1024, -(-h // 16), -(-w // 16)), 16)
    rmap, obj, bg = O.generate_rpn_map(am, vm, np.stack([k for _, k in gts]))
    rmap_t = torch.from_numpy(rmap).unsqueeze(0)
    sd = sd_cpu if vgg else synthetic.resnet_state_dict(1234, "ResNet50")

    def run(parallel):
        if vgg:
            model = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
        else:
            from fasterrcnn_amd.models import resnet
            model = FasterRCNNModel(num_classes=21, backbone=resnet.ResNetBackbone(resnet.Architecture.ResNet50))
        model.load_state_dict(sd, strict=True)
        model = model.cuda()
        if parallel:
            T.enable_data_parallel(model, bucket_bytes=64 << 20 if vgg else 4 << 20)
        opt = T.create_optimizer(model, learning_rate=1e-6)
        random.seed(9); torch.manual_seed(9)
        loss = model.train_step(opt, img, am, vm, rmap_t, [obj], [bg], [boxes])
        if parallel:
            # fc1, fc2, rpn_conv and the 512-channel conv gradients (>= 8 MB) travel in place as they are produced, the rest in buckets
            assert model._gradient_sync.messages >= 8 and not model._gradient_sync._inflight
        return loss, {k: v.clone() for k, v in model.state_dict().items()}

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    l0, s0 = run(False)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        l1, s1 = run(True)
    finally:
        dist.destroy_process_group()
    assert l0 == l1
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


def test_direct_parameter_writes_between_steps_are_honoured(sd_cpu):
    """the packed training masters are cloned once.  A parameter written directly between two steps (sub-module
    load_state_dict, manual re-init, a torch optimizer) must reach the next step instead of being overwritten by stale masters;
    `.data` writes (no version bump) need invalidate_packed(); a conflicting write while trained weights are pending raises."""
    from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
    from fasterrcnn_amd.models.vgg16 import VGG16Backbone
    h, w, seed = 352, 480, 4
    img = synthetic.image(seed, h, w).unsqueeze(0).cuda()
    gts = synthetic.ground_truth(seed, h, w)
    boxes = [Box(class_index=c, class_name="x", corners=k) for c, k in gts]
    am, vm = O.generate_anchor_maps((3, h, w), (512, h // 16, w // 16), 16)
    rmap, obj, bg = O.generate_rpn_map(am, vm, np.stack([k for _, k in gts]))
    rmap_t = torch.from_numpy(rmap).unsqueeze(0)

    def fresh():
        m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
        m.load_state_dict(sd_cpu, strict=True)
        return m.cuda()

    def step(m, opt, s):
        random.seed(s); torch.manual_seed(s)
        return m.train_step(opt, img, am, vm, rmap_t, [obj], [bg], [boxes])

    key = "_stage3_detector_network._regressor.weight"
    # run A: step, then the regressor is re-initialised through the parameter API, then a second step
    a = fresh()
    oa = T.create_optimizer(a, learning_rate=1e-6)
    step(a, oa, 1)
    a.sync_parameters()
    with torch.no_grad():
        a._stage3_detector_network._regressor.weight.mul_(0.5)          # in-place, bumps _version
    la = step(a, oa, 2)
    sa = a.state_dict()
    # run B: the same, but the write goes through a state_dict round trip of the whole model (the path that always worked)
    b = fresh()
    ob = T.create_optimizer(b, learning_rate=1e-6)
    step(b, ob, 1)
    sd_mid = {k: v.clone() for k, v in b.state_dict().items()}
    sd_mid[key] = sd_mid[key] * 0.5
    momentum = b._train_state.momentum
    b.load_state_dict(sd_mid, strict=True)
    b._training_state().momentum = momentum                               # load_state_dict drops the optimizer state; keep it comparable
    lb = step(b, ob, 2)
    sb = b.state_dict()
    assert la == lb
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    # the halved regressor really was used: the detector regression loss differs from an un-modified second step
    c = fresh()
    oc = T.create_optimizer(c, learning_rate=1e-6)
    step(c, oc, 1)
    lc = step(c, oc, 2)
    assert lc.detector_regression != la.detector_regression
    # a write while trained weights are still pending in the masters is refused, not silently lost
    with torch.no_grad():
        c._stage3_detector_network._regressor.weight.mul_(2.0)
    with pytest.raises(RuntimeError):
        c.state_dict()
    # .data writes bypass the version counter: invalidate_packed() is the documented hook
    d = fresh().eval()
    im = synthetic.image(2, 224, 320).unsqueeze(0).cuda()
    before = d(image_data=im)
    d._stage3_detector_network._regressor.weight.data.mul_(2.0)
    d.invalidate_packed()
    after = d(image_data=im)
    assert torch.allclose(after[2], before[2] * 2.0, rtol=1e-5, atol=1e-6)


def test_constructor_refuses_unsupported_capacities():
    from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
    from fasterrcnn_amd.models.vgg16 import VGG16Backbone
    with pytest.raises(ValueError, match="num_classes"):
        FasterRCNNModel(num_classes=104, backbone=VGG16Backbone(dropout_probability=0.0))
    from fasterrcnn_amd import training
    m81 = FasterRCNNModel(num_classes=81, backbone=VGG16Backbone(dropout_probability=0.0)).cuda()      # inference: fine (round 3)
    with pytest.raises(NotImplementedError, match="num_classes"):
        training.make_train_state(m81)                                                                    # the train step keeps <= 26
    del m81
    m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0)).cuda().eval()
    m.max_proposals_post_nms = 1000
    with pytest.raises(ValueError, match="max_proposals_post_nms"):
        m.predict(synthetic.image(1, 64, 64).unsqueeze(0).cuda(), 0.05)


def test_backward_chain_elementwise_on_injected_oracle_activations(sd_cpu):
    """the statistical gradient bounds above come from ReLU / max-pool / RoI-argmax decisions that two float32
    forwards take differently near ties.  Here the decisions are PRESCRIBED: the oracle's forward activations (post-ReLU conv
    outputs, RPN trunk, RoI-pooled features, fc1, fc2) are injected into the train step, so the whole backward chain -- loss
    gradients, every GEMM / conv weight and data gradient, ReLU / max-pool / RoI-pool backward -- runs on identical inputs and
    masks, and EVERY gradient tensor must agree elementwise to float32 summation-order accuracy (models/faster_rcnn.py:355)."""
    from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
    from fasterrcnn_amd.models.vgg16 import VGG16Backbone
    from oracle import train_oracle as TO
    h, w, seed = 352, 480, 4
    img = synthetic.image(seed, h, w).unsqueeze(0)
    gts = synthetic.ground_truth(seed, h, w)
    boxes = [Box(class_index=c, class_name="x", corners=k) for c, k in gts]
    am, vm = O.generate_anchor_maps((3, h, w), (512, h // 16, w // 16), 16)
    rmap, obj, bg = O.generate_rpn_map(am, vm, np.stack([k for _, k in gts]))
    random.seed(5); torch.manual_seed(5)
    od = {}
    o_losses, o_grads, _, _ = TO.train_step(sd_cpu, img, am, vm, torch.from_numpy(rmap).unsqueeze(0), obj, bg,
                                            np.stack([k for _, k in gts]), np.array([c for c, _ in gts]), 21, 1e-6, 0.9, 5e-4, detail=od)
    hwc = lambda x: x.detach()[0].permute(1, 2, 0).contiguous().cuda()
    names = [n for n, _ in TO.VGG_LAYERS]
    inject = {"conv%d" % i: hwc(od[names[i]]) for i in range(4, 13)}
    inject["conv4_in"] = hwc(torch.nn.functional.max_pool2d(od[names[3]], 2, 2))
    # the oracle keeps the watched activations only as gradients; recompute the three that are not in `od` from its tensors
    p = {k: v for k, v in sd_cpu.items()}
    fm_o = od[names[12]].detach()
    trunk_o = torch.relu(torch.nn.functional.conv2d(fm_o, p["_stage2_region_proposal_network._rpn_conv1.weight"],
                                                    p["_stage2_region_proposal_network._rpn_conv1.bias"], padding=1))
    inject["rpn_trunk"] = hwc(trunk_o)
    sprops = od["sampled"][0]
    pooled_o = TO.roi_pool_autograd(fm_o, sprops).detach()                                  # (S, 512, 7, 7)
    S = pooled_o.shape[0]
    inject["roi_out"] = pooled_o.permute(0, 2, 3, 1).contiguous().reshape(S, 49 * 512).cuda()
    pv = "_stage3_detector_network._pool_to_feature_vector."
    h1_o = torch.relu(torch.nn.functional.linear(pooled_o.reshape(S, -1), p[pv + "_fc1.weight"], p[pv + "_fc1.bias"]))
    h2_o = torch.relu(torch.nn.functional.linear(h1_o, p[pv + "_fc2.weight"], p[pv + "_fc2.bias"]))
    inject["fc1"], inject["fc2"] = h1_o.cuda().contiguous(), h2_o.cuda().contiguous()
    model = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
    model.load_state_dict(sd_cpu, strict=True)
    model = model.cuda()
    for mode in ("f32_winograd", "f32"):
        model.load_state_dict(sd_cpu, strict=True)
        model.math_mode = mode
        opt = T.create_optimizer(model, learning_rate=1e-6)
        random.seed(5); torch.manual_seed(5)
        detail = {"inject": inject}
        loss = T.train_step(model, opt, img.cuda(), am, vm, torch.from_numpy(rmap).unsqueeze(0), [obj], [bg], [boxes], detail=detail)
        assert np.array_equal(detail["sample_idx"].numpy(), od["proposal_sample_indices"])
        got = np.array([loss.rpn_class, loss.rpn_regression, loss.detector_class, loss.detector_regression])
        want = np.array([o_losses[k] for k in ("rpn_class", "rpn_regression", "detector_class", "detector_regression")])
        assert np.all(np.abs(got - want) <= 1e-5 * np.abs(want) + 1e-7), (got, want)
        grads = canonical_grads(detail["grads"])
        worst = (0.0, "")
        for k, g_ref in o_grads.items():
            e = float((grads[k].cpu() - g_ref).abs().max()) / float(g_ref.abs().max())
            if e > worst[0]:
                worst = (e, k)
            assert e <= 1e-5, (mode, k, e)
        print("injected-activation backward (%s): worst elementwise gradient error / max|g| = %.2e (%s)" % (mode, worst[0], worst[1]))
