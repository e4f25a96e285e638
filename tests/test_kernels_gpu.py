"""
Per-kernel parity on a real MI355X: every entry point of include/frcnn_hip.h is called through
the C ABI (ctypes) on seeded inputs and compared with oracle/frcnn_oracle.py (or the torch-CPU
fp32/fp64 op the reference calls).  Integer / index / max-pool work must be bit-exact; floating
point GEMM-class kernels are held to a tolerance derived from an fp64 ground truth (stated in each
test).
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fasterrcnn_amd import _native as nv
from fasterrcnn_amd import runtime as rt
from oracle import frcnn_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def S():
    return nv.stream_ptr()


def gpu(x):
    return torch.as_tensor(x).to(DEV).contiguous()


@pytest.fixture(scope="module")
def ctx():
    return rt.Context(DEV, 608, 1008, 300)


def test_device_is_gfx950():
    nv.require_gpu()
    assert nv.lib().frcnn_device_count() >= 1


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ishape,fshape", [((3, 600, 1000), (512, 37, 62)), ((3, 600, 1000), (1024, 38, 63)),
                                           ((3, 224, 320), (512, 14, 20)), ((3, 333, 517), (512, 20, 32))])
def test_anchors_bit_exact(ishape, fshape):
    am = torch.empty((fshape[1], fshape[2], 36), device=DEV)
    vm = torch.empty((fshape[1], fshape[2], 9), device=DEV)
    nv.check(nv.lib().frcnn_anchors(ishape[1], ishape[2], fshape[1], fshape[2], 16, nv.ptr(am), nv.ptr(vm), S()), "anchors")
    ram, rvm = O.generate_anchor_maps(ishape, fshape, 16)
    assert np.array_equal(am.cpu().numpy(), ram)
    assert np.array_equal(vm.cpu().numpy(), rvm)


# ---------------------------------------------------------------------------------------------
def conv_ref(x_chw, w, b, relu, pool, dtype):
    y = F.conv2d(x_chw.unsqueeze(0).to(dtype), w.to(dtype), b.to(dtype), padding=1)
    if relu:
        y = F.relu(y)
    if pool:
        y = F.max_pool2d(y, 2, 2)
    return y[0]


def check_conv(y_hwc, x_chw, w, b, relu, pool, what):
    """Tolerance: |ours - fp64 truth| <= 4e-6 * sqrt(K) * max|truth| (fp32 accumulation over K
    terms; the torch-CPU fp32 result is printed beside it and lands in the same band)."""
    ours = y_hwc.cpu().permute(2, 0, 1).double()
    truth = conv_ref(x_chw, w, b, relu, pool, torch.float64)
    cpu32 = conv_ref(x_chw, w, b, relu, pool, torch.float32).double()
    assert ours.shape == truth.shape, (ours.shape, truth.shape)
    scale = float(truth.abs().max()) + 1e-30
    k = w.shape[1] * 9
    e_ours = float((ours - truth).abs().max()) / scale
    e_cpu = float((cpu32 - truth).abs().max()) / scale
    print("%s: rel err ours %.3g, torch-cpu-fp32 %.3g (K=%d)" % (what, e_ours, e_cpu, k))
    assert e_ours <= 4e-6 * np.sqrt(k), (what, e_ours)


@pytest.mark.parametrize("H,W", [(600, 1000), (33, 70), (1, 1), (5, 257)])
def test_conv3x3_c3(H, W):
    g = torch.Generator().manual_seed(H * 1000 + W)
    x = torch.randn((3, H, W), generator=g) * 60
    w = torch.randn((64, 3, 3, 3), generator=g) * 0.2
    b = torch.randn((64,), generator=g)
    wp = torch.empty((27, 64), device=DEV)
    dw, dx, db = gpu(w), gpu(x), gpu(b)          # keep device inputs alive across the async launches
    nv.check(nv.lib().frcnn_pack_conv3x3_c3(nv.ptr(dw), nv.ptr(wp), 64, S()), "pack")
    y = torch.empty((H, W, 64), device=DEV)
    nv.check(nv.lib().frcnn_conv3x3_c3(nv.ptr(dx), nv.ptr(wp), nv.ptr(db), nv.ptr(y), H, W, 64, nv.RELU, S()), "conv_c3")
    check_conv(y, x, w, b, True, False, "conv_c3 %dx%d" % (H, W))


@pytest.mark.parametrize("H,W,relu", [(600, 1000, True), (33, 70, True), (1, 1, True), (2, 64, False), (5, 257, False), (601, 999, True)])
def test_conv3x3_c3_matches_the_generic_kernel(H, W, relu):
    """The 64-channel kernel of conv1_1 (round 5: persistent, four adjacent pixels per thread, packed FMAs) against the pixel-per-thread
    kernel every other width runs on: the same fmaf chain per output, so the same bits.  (The generic kernel is reached with the filter
    bank twice in a row: 128 output channels, the first 64 and the last 64 both = the layer.)"""
    g = torch.Generator().manual_seed(H * 31 + W)
    x = gpu(torch.randn((3, H, W), generator=g) * 60)
    w = torch.randn((64, 3, 3, 3), generator=g) * 0.2
    b = torch.randn((64,), generator=g)
    lib = nv.lib()
    w64, b64 = gpu(w), gpu(b)
    w128, b128 = gpu(torch.cat([w, w])), gpu(torch.cat([b, b]))
    wp64, wp128 = torch.empty((27, 64), device=DEV), torch.empty((27, 128), device=DEV)
    nv.check(lib.frcnn_pack_conv3x3_c3(nv.ptr(w64), nv.ptr(wp64), 64, S()), "pack")
    nv.check(lib.frcnn_pack_conv3x3_c3(nv.ptr(w128), nv.ptr(wp128), 128, S()), "pack")
    fl = nv.RELU if relu else 0
    y64 = torch.full((H, W, 64), float("nan"), device=DEV)
    y128 = torch.full((H, W, 128), float("nan"), device=DEV)
    nv.check(lib.frcnn_conv3x3_c3(nv.ptr(x), nv.ptr(wp64), nv.ptr(b64), nv.ptr(y64), H, W, 64, fl, S()), "conv_c3")
    nv.check(lib.frcnn_conv3x3_c3(nv.ptr(x), nv.ptr(wp128), nv.ptr(b128), nv.ptr(y128), H, W, 128, fl, S()), "conv_c3")
    assert not torch.isnan(y64).any()
    assert torch.equal(y64, y128[:, :, :64]) and torch.equal(y64, y128[:, :, 64:])


@pytest.mark.parametrize("H,W,relu", [(600, 1000, True), (33, 70, True), (1, 1, True), (5, 257, False)])
def test_conv3x3_c3_leaves_the_channel_maxima(H, W, relu):
    """frcnn_conv3x3_c3_cmax (round 4): conv1_1 writes the per-pixel maximum |y| over its 64 output channels next to y -- the scale source of
    conv1_2 as a one-launch f32x3 layer.  y is the bits of frcnn_conv3x3_c3, the maxima those of frcnn_pixel_absmax over y (what the
    consumer would otherwise compute with one more pass over the tensor); cout != 64 is refused."""
    g = torch.Generator().manual_seed(H * 1000 + W + 1)
    x = gpu(torch.randn((3, H, W), generator=g) * 60)
    w = gpu(torch.randn((64, 3, 3, 3), generator=g) * 0.2)
    b = gpu(torch.randn((64,), generator=g))
    lib = nv.lib()
    wp = torch.empty((27, 64), device=DEV)
    nv.check(lib.frcnn_pack_conv3x3_c3(nv.ptr(w), nv.ptr(wp), 64, S()), "pack")
    fl = nv.RELU if relu else 0
    y0 = torch.empty((H, W, 64), device=DEV)
    nv.check(lib.frcnn_conv3x3_c3(nv.ptr(x), nv.ptr(wp), nv.ptr(b), nv.ptr(y0), H, W, 64, fl, S()), "conv_c3")
    y1 = torch.full((H, W, 64), float("nan"), device=DEV)
    cm = torch.full((H * W,), float("nan"), device=DEV)
    nv.check(lib.frcnn_conv3x3_c3_cmax(nv.ptr(x), nv.ptr(wp), nv.ptr(b), nv.ptr(y1), H, W, 64, fl, nv.ptr(cm), S()), "conv_c3_cmax")
    want = torch.empty((H * W,), device=DEV)
    nv.check(lib.frcnn_pixel_absmax(nv.ptr(y0), nv.ptr(want), H * W, 64, S()), "pixel_absmax")
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    assert torch.equal(cm, want) and torch.equal(cm, y0.abs().amax(dim=2).reshape(-1))
    assert lib.frcnn_conv3x3_c3_cmax(nv.ptr(x), nv.ptr(wp), nv.ptr(b), nv.ptr(y1), H, W, 32, fl, nv.ptr(cm), S()) == -1      # FRCNN_EINVAL
    assert lib.frcnn_conv3x3_c3_cmax(nv.ptr(x), nv.ptr(wp), nv.ptr(b), nv.ptr(y1), H, W, 64, fl, None, S()) == -1


@pytest.mark.parametrize("H,W,cin,cout,pool,relu", [
    (37, 62, 512, 512, False, True),     # block5 / RPN trunk shape
    (75, 125, 256, 512, True, True),     # block4 with fused pool (odd H, W: floor)
    (24, 40, 64, 64, True, True),        # cout = 64 tile config
    (9, 33, 16, 128, False, False),      # ragged width, no relu
    (2, 2, 16, 64, True, True),          # smallest poolable
    (1, 1, 16, 64, False, True),         # single pixel
    (8, 32, 128, 256, False, True),      # exact tile
    (7, 31, 32, 192, True, True),        # cout multiple of 64 but not 128
    (150, 250, 128, 256, False, True),   # block3 shape
])
def test_conv3x3_nhwc(H, W, cin, cout, pool, relu):
    g = torch.Generator().manual_seed(H * 7 + W * 13 + cin + cout)
    x = torch.randn((cin, H, W), generator=g)
    w = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.1
    wp = torch.empty((9, cout, cin), device=DEV)
    dw, db = gpu(w), gpu(b)
    nv.check(nv.lib().frcnn_pack_conv3x3(nv.ptr(dw), nv.ptr(wp), cout, cin, S()), "pack")
    # pack layout: [tap][cout][cin]
    assert torch.equal(wp.cpu(), w.permute(2, 3, 0, 1).reshape(9, cout, cin))
    xh = gpu(x.permute(1, 2, 0))
    oh, ow = (H // 2, W // 2) if pool else (H, W)
    y = torch.full((oh, ow, cout), float("nan"), device=DEV)
    flags = (nv.RELU if relu else 0) | (nv.POOL2 if pool else 0)
    lib = nv.lib()
    ws_bytes = int(lib.frcnn_conv3x3_workspace_bytes(H, W, cin, cout))
    ws = torch.empty((max(ws_bytes, 4) // 4,), device=DEV)
    nv.check(lib.frcnn_conv3x3_nhwc(nv.ptr(xh), nv.ptr(wp), nv.ptr(db), nv.ptr(y), H, W, cin, cout, flags,
                                    nv.ptr(ws), ws_bytes, S()), "conv")
    assert not torch.isnan(y).any()
    check_conv(y, x, w, b, relu, pool, "conv %dx%d %d->%d pool=%d (split-K ws %d B)" % (H, W, cin, cout, pool, ws_bytes))
    # without scratch the layer runs un-split: same result up to fp32 summation order
    y1 = torch.full((oh, ow, cout), float("nan"), device=DEV)
    nv.check(lib.frcnn_conv3x3_nhwc(nv.ptr(xh), nv.ptr(wp), nv.ptr(db), nv.ptr(y1), H, W, cin, cout, flags, None, 0, S()), "conv")
    check_conv(y1, x, w, b, relu, pool, "   un-split")
    if ws_bytes == 0:
        assert torch.equal(y, y1)


def test_conv_rejects_unsupported_shapes():
    d = torch.zeros(16, device=DEV)
    lib = nv.lib()
    assert lib.frcnn_conv3x3_nhwc(nv.ptr(d), nv.ptr(d), nv.ptr(d), nv.ptr(d), 4, 4, 3, 64, 0, None, 0, S()) == -1    # cin % 16
    assert lib.frcnn_conv3x3_nhwc(nv.ptr(d), nv.ptr(d), nv.ptr(d), nv.ptr(d), 4, 4, 16, 32, 0, None, 0, S()) == -1   # cout % 64
    assert lib.frcnn_conv3x3_c3(nv.ptr(d), nv.ptr(d), nv.ptr(d), nv.ptr(d), 4, 4, 64, nv.POOL2, S()) == -4


def test_maxpool_exact():
    g = torch.Generator().manual_seed(3)
    x = torch.randn((64, 75, 125), generator=g)
    y = torch.empty((37, 62, 64), device=DEV)
    dx = gpu(x.permute(1, 2, 0))
    nv.check(nv.lib().frcnn_maxpool2x2_nhwc(nv.ptr(dx), nv.ptr(y), 75, 125, 64, S()), "maxpool")
    assert torch.equal(y.cpu().permute(2, 0, 1), F.max_pool2d(x.unsqueeze(0), 2, 2)[0])


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,relu", [
    (300, 4096, 25088, True),    # fc1 of one image
    (300, 4096, 4096, True),     # fc2
    (300, 101, 4096, False),     # stacked detector heads
    (2294, 45, 512, False),      # stacked RPN 1x1 heads on the 37x62 map
    (1, 128, 16, False), (5, 200, 48, True), (321, 128, 32, False), (640, 256, 64, True), (0 + 17, 1, 16, False),
])
def test_linear(M, N, K, relu):
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn((M, K), generator=g)
    n_rows = (N + 127) // 128 * 128
    w = torch.zeros((n_rows, K))
    w[:N] = torch.randn((N, K), generator=g) * (1.0 / K) ** 0.5
    b = torch.randn((N,), generator=g)
    lib = nv.lib()
    ws_bytes = int(lib.frcnn_linear_workspace_bytes(M, N, K))
    ws = torch.empty((max(ws_bytes, 4) // 4,), device=DEV)
    ldy = N + 3
    y = torch.full((M, ldy), 7.0, device=DEV)
    da, dw, db = gpu(a), gpu(w), gpu(b)
    nv.check(lib.frcnn_linear(nv.ptr(da), K, nv.ptr(dw), nv.ptr(db), nv.ptr(y), ldy, M, N, K,
                              nv.RELU if relu else 0, nv.ptr(ws), ws_bytes, S()), "linear")
    out = y.cpu()
    assert (out[:, N:] == 7.0).all()                                  # nothing written past N
    truth = a.double() @ w[:N].double().t() + b.double()
    cpu32 = (a @ w[:N].t() + b).double()
    if relu:
        truth, cpu32 = truth.clamp(min=0), cpu32.clamp(min=0)
    scale = float(truth.abs().max()) + 1e-30
    e_ours = float((out[:, :N].double() - truth).abs().max()) / scale
    e_cpu = float((cpu32 - truth).abs().max()) / scale
    print("linear %dx%dx%d: rel err ours %.3g, torch-cpu-fp32 %.3g" % (M, N, K, e_ours, e_cpu))
    # tolerance: 4e-6 * sqrt(K) relative to the largest output (fp32 accumulation over K terms)
    assert e_ours <= 4e-6 * np.sqrt(K)
    # deterministic split-K: a second run is bit-identical
    y2 = torch.full((M, ldy), 7.0, device=DEV)
    nv.check(lib.frcnn_linear(nv.ptr(da), K, nv.ptr(dw), nv.ptr(db), nv.ptr(y2), ldy, M, N, K,
                              nv.RELU if relu else 0, nv.ptr(ws), ws_bytes, S()), "linear")
    assert torch.equal(y2.cpu(), out)


def test_fc_weight_permutation_and_stack_rows():
    g = torch.Generator().manual_seed(9)
    w = torch.randn((8, 512 * 49), generator=g)
    wp = torch.empty((8, 512 * 49), device=DEV)
    dw = gpu(w)
    nv.check(nv.lib().frcnn_pack_fc_chw_to_hwc(nv.ptr(dw), nv.ptr(wp), 8, 512, 49, S()), "pack_fc")
    assert torch.equal(wp.cpu(), w.reshape(8, 512, 49).permute(0, 2, 1).reshape(8, -1))
    w1, b1 = torch.randn((9, 64), generator=g), torch.randn((9,), generator=g)
    w2, b2 = torch.randn((36, 64), generator=g), torch.randn((36,), generator=g)
    wo, bo = torch.empty((128, 64), device=DEV), torch.empty((128,), device=DEV)
    d1, e1, d2, e2 = gpu(w1), gpu(b1), gpu(w2), gpu(b2)
    nv.check(nv.lib().frcnn_pack_stack_rows(nv.ptr(d1), nv.ptr(e1), 9, nv.ptr(d2), nv.ptr(e2), 36,
                                            64, 128, nv.ptr(wo), nv.ptr(bo), S()), "stack")
    assert torch.equal(wo.cpu()[:45], torch.cat([w1, w2])) and not wo.cpu()[45:].any()
    assert torch.equal(bo.cpu()[:45], torch.cat([b1, b2])) and not bo.cpu()[45:].any()


def test_softmax_rows():
    g = torch.Generator().manual_seed(4)
    x = torch.randn((300, 128), generator=g) * 3
    y = torch.empty((300, 21), device=DEV)
    dx = gpu(x)
    nv.check(nv.lib().frcnn_softmax_rows(nv.ptr(dx), 128, nv.ptr(y), 300, 21, S()), "softmax")
    ref = F.softmax(x[:, :21], dim=1)
    assert float((y.cpu() - ref).abs().max()) <= 2e-7         # a few ulp of values <= 1


# ---------------------------------------------------------------------------------------------
def run_proposals(ctx, head, am, vm, fh, fw, ih, iw, pre, post, allow_edge):
    a = fh * fw * 9
    scores = torch.empty((a,), device=DEV)
    sidx = torch.full((pre,), -1, dtype=torch.int32, device=DEV)
    props = torch.full((post, 4), -1.0, device=DEV)
    counts = torch.zeros((4,), dtype=torch.int32, device=DEV)
    nv.check(nv.lib().frcnn_rpn_proposals(ctx.handle, nv.ptr(head), 128, nv.ptr(am), None if allow_edge else nv.ptr(vm),
                                          fh, fw, ih, iw, pre, post, 0.7, 16.0, nv.ptr(scores), nv.ptr(sidx),
                                          nv.ptr(props), nv.ptr(counts), S()), "rpn_proposals")
    torch.cuda.synchronize()
    return scores.cpu(), sidx.cpu().numpy(), props.cpu(), counts.cpu().numpy()


@pytest.mark.parametrize("ih,iw,pre,post,allow_edge,seed", [
    (600, 1000, 6000, 300, True, 1), (600, 1000, 6000, 300, False, 2), (224, 320, 6000, 300, True, 3),
    (600, 1000, 1000, 50, True, 4), (333, 517, 6000, 300, False, 5), (600, 1000, 8192, 300, True, 6),
])
def test_rpn_proposals_vs_oracle(ctx, ih, iw, pre, post, allow_edge, seed):
    fh, fw = ih // 16, iw // 16
    g = torch.Generator().manual_seed(seed)
    head = torch.zeros((fh * fw, 128))
    head[:, 0:9] = torch.randn((fh * fw, 9), generator=g) * 1.5
    head[:, 9:45] = torch.randn((fh * fw, 36), generator=g) * 0.3
    am, vm = O.generate_anchor_maps((3, ih, iw), (512, fh, fw), 16)
    dh, dam, dvm = gpu(head), gpu(am), gpu(vm)
    scores, sidx, props, counts = run_proposals(ctx, dh, dam, dvm, fh, fw, ih, iw, pre, post, allow_edge)

    score_map = torch.sigmoid(head[:, 0:9]).reshape(1, fh, fw, 9)
    delta_map = head[:, 9:45].reshape(1, fh, fw, 36)
    detail = {}
    ref = O.proposals_from_maps(score_map, delta_map, (3, ih, iw), am, vm, pre, post, allow_edge, detail)
    # sigmoid: a few ulp
    assert float((scores - score_map.reshape(-1)).abs().max()) <= 2e-7
    n_sel = len(detail["sorted_idx"])
    assert counts[0] == n_sel
    # top-N order: exact wherever the oracle's neighbouring scores differ by more than sigmoid noise
    ref_idx = detail["sorted_idx"]
    ref_sorted_scores = score_map.reshape(-1)[torch.from_numpy(ref_idx)].double().numpy()
    same = sidx[:n_sel] == ref_idx
    gap = np.minimum(np.abs(np.diff(ref_sorted_scores, prepend=np.inf)), np.abs(np.diff(ref_sorted_scores, append=-np.inf)))
    assert same[gap > 1e-6].all(), "order differs where scores are well separated"
    assert sorted(sidx[:n_sel].tolist()) == sorted(ref_idx.tolist()) or (~same).sum() <= 4
    print("top-%d: %d/%d positions identical" % (n_sel, int(same.sum()), n_sel))
    assert counts[1] == detail["n_after_filter"]
    n = int(counts[2])
    assert n == ref.shape[0]
    # boxes: same expf only up to ulp -> <= 1e-3 px (north_star tolerance); typically ~1e-4
    err = float((props[:n] - ref).abs().max()) if n else 0.0
    print("proposals: %d, max |d| %.3g px" % (n, err))
    assert err <= 1e-3
    assert not props[n:].any()                                          # rows past the count are zeroed


def test_rpn_proposals_score_ties_break_to_higher_index(ctx):
    fh, fw, ih, iw = 14, 20, 224, 320
    head = torch.zeros((fh * fw, 128))
    head[:, 0:9] = 20.0                     # sigmoid saturates to exactly 1.0 everywhere: all tied
    am, vm = O.generate_anchor_maps((3, ih, iw), (512, fh, fw), 16)
    dh, dam, dvm = gpu(head), gpu(am), gpu(vm)
    scores, sidx, props, counts = run_proposals(ctx, dh, dam, dvm, fh, fw, ih, iw, 500, 300, True)
    a = fh * fw * 9
    assert (scores == 1.0).all()
    assert sidx[:500].tolist() == list(range(a - 1, a - 501, -1))
    detail = {}
    O.proposals_from_maps(torch.sigmoid(head[:, 0:9]).reshape(1, fh, fw, 9), head[:, 9:45].reshape(1, fh, fw, 36),
                          (3, ih, iw), am, vm, 500, 300, True, detail)
    assert detail["sorted_idx"].tolist() == sidx[:500].tolist()


def make_boxes(n, seed, clusters=40):
    rng = np.random.RandomState(seed)
    centers = rng.rand(clusters, 2) * np.array([560, 960]) + 20
    which = rng.randint(0, clusters, size=n)
    c = centers[which] + rng.randn(n, 2) * 6
    hw = np.abs(rng.randn(n, 2)) * 30 + 30 + rng.rand(n, 2) * 3
    b = np.concatenate([c - hw / 2, c + hw / 2], axis=1).astype(np.float32)
    return b


@pytest.mark.parametrize("n,thr,quant,clusters,max_keep", [
    (0, 0.7, False, 40, 2048), (1, 0.7, False, 40, 2048), (63, 0.7, False, 40, 2048), (64, 0.5, False, 40, 2048),
    (65, 0.3, False, 40, 2048), (1000, 0.7, True, 40, 2048), (6000, 0.7, False, 40, 2048), (8000, 0.3, True, 40, 2048),
    # round 5 (nms_reduce_kernel's look-ahead ring): few clusters = long runs of chunks that keep nothing; thousands of clusters = chunks
    # that keep more rows than the ring defers; > 8192 candidates = the second half of removed[]; small max_keep = the early exit
    (6000, 0.7, False, 3, 2048), (6000, 0.5, False, 3000, 2048), (6000, 0.7, False, 40, 10), (6000, 0.7, False, 3000, 300),
    (12000, 0.5, False, 40, 2048), (16384, 0.7, True, 400, 2048), (16384, 0.6, False, 8000, 2048), (8193, 0.7, False, 40, 2048),
    (257, 0.7, False, 40, 2048), (320, 0.7, False, 1, 2048)])
def test_nms_vs_oracle(ctx, n, thr, quant, clusters, max_keep):
    b = make_boxes(n, n + 1, clusters)
    rng = np.random.RandomState(n)
    s = rng.rand(n).astype(np.float32)
    if quant:
        s = np.round(s * 50).astype(np.float32) / 50      # many exact ties: stable order must hold
    keep = torch.full((2048,), -1, dtype=torch.int32, device=DEV)
    nk = torch.full((1,), -1, dtype=torch.int32, device=DEV)
    bb = gpu(b) if n else torch.zeros((1, 4), device=DEV)
    ss = gpu(s) if n else torch.zeros((1,), device=DEV)
    nv.check(nv.lib().frcnn_nms(ctx.handle, nv.ptr(bb), nv.ptr(ss), n, thr, max_keep, nv.ptr(keep), nv.ptr(nk), S()), "nms")
    ref = O.nms(b, s, thr)[:max_keep]
    got = keep.cpu().numpy()[: int(nk.item())]
    assert int(nk.item()) == len(ref)
    assert np.array_equal(got, ref.astype(np.int32))


@pytest.mark.parametrize("n,j_list", [(9000, (5, 100, 777)), (12000, (1, 64, 3807)), (16384, (8191,))])
def test_nms_wide_box0_does_not_reach_into_the_upper_half(ctx, n, j_list):
    """ADVICE r5 (nms_reduce_kernel, > 8192 candidates): the prologue filled the ring's odd row places with the LOWER half of row 0, which
    chunks 0 and 1 OR into words 128 .. 255 of removed[] -- every box j < 8192 that box 0 suppresses also suppressed candidate 8192 + j.
    Directed case (the training path's shape, pre_nms = 12000: models/faster_rcnn.py:302 of the reference): scores descend with the index,
    the first 8192 boxes sit on 40 sites (few kept: max_keep is not reached before index 8192), box 0 covers boxes j, and boxes
    8192 + j are ISOLATED: they must be kept."""
    rng = np.random.RandomState(n)
    sites = np.stack(np.meshgrid(np.arange(8) * 120.0, np.arange(5) * 110.0), axis=-1).reshape(-1, 2)          # 40 sites, 100 x 90 boxes
    which = rng.randint(1, 40, size=n)
    which[0] = 0
    for j in j_list:
        which[j] = 0                                       # box j sits on box 0's site: suppressed by box 0
    tl = sites[which] + rng.rand(n, 2) * 2.0
    b = np.concatenate([tl, tl + np.array([100.0, 90.0])], axis=1).astype(np.float32)
    for k, j in enumerate(j_list):                         # isolated boxes far from every site
        b[8192 + j] = [2000.0 + 150.0 * k, 2000.0, 2100.0 + 150.0 * k, 2090.0]
    s = np.linspace(1.0, 0.01, n).astype(np.float32)
    keep = torch.full((2048,), -1, dtype=torch.int32, device=DEV)
    nk = torch.full((1,), -1, dtype=torch.int32, device=DEV)
    bb, ss = gpu(b), gpu(s)                                # (named: the launch is asynchronous, a temporary's memory could be handed out again)
    nv.check(nv.lib().frcnn_nms(ctx.handle, nv.ptr(bb), nv.ptr(ss), n, 0.5, 2048, nv.ptr(keep), nv.ptr(nk), S()), "nms")
    got = keep.cpu().numpy()[: int(nk.item())]
    ref = O.nms(b, s, 0.5)[:2048]
    for j in j_list:
        assert j not in set(got.tolist())
        assert 8192 + j in set(ref.tolist())               # (the case is what it claims to be)
        assert 8192 + j in set(got.tolist()), "box 0 removed candidate %d through the upper half of removed[]" % (8192 + j)
    assert np.array_equal(got, ref.astype(np.int32))


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c,fh,fw,n", [(512, 37, 62, 300), (64, 20, 32, 57), (1024, 38, 63, 16)])
def test_roi_pool_bit_exact(c, fh, fw, n):
    g = torch.Generator().manual_seed(c + n)
    fm = torch.randn((1, c, fh, fw), generator=g)
    rng = np.random.RandomState(n)
    y1 = rng.uniform(-40, fh * 16, n); x1 = rng.uniform(-40, fw * 16, n)
    rois = np.stack([y1, x1, y1 + rng.uniform(0, 400, n), x1 + rng.uniform(0, 600, n)], axis=1).astype(np.float32)
    rois[0] = [0, 0, fh * 16, fw * 16]                     # whole map
    rois[1] = [8, 8, 8, 8]                                 # degenerate: one cell
    rois[2] = [fh * 16 + 100, fw * 16 + 100, fh * 16 + 200, fw * 16 + 300]   # outside: empty bins
    rois[3] = [24, 40, 24 + 7.99, 40 + 8.0]                # .5 rounding boundaries after /16
    maxr = n + 5
    r = torch.zeros((maxr, 4)); r[:n] = torch.from_numpy(rois)
    out = torch.full((maxr, 7, 7, c), float("nan"), device=DEV)
    cnt = torch.tensor([n], dtype=torch.int32, device=DEV)
    dfm, dr = gpu(fm[0].permute(1, 2, 0)), gpu(r)
    nv.check(nv.lib().frcnn_roi_pool(nv.ptr(dfm), fh, fw, c, nv.ptr(dr), nv.ptr(cnt), maxr, 7,
                                     1.0 / 16.0, nv.ptr(out), S()), "roi_pool")
    rois5 = np.zeros((n, 5), np.float32); rois5[:, 1:] = rois[:, [1, 0, 3, 2]]
    ref = O.roi_pool(fm.numpy(), rois5, 7, 1.0 / 16.0)               # (n, c, 7, 7)
    got = out.cpu()
    assert torch.equal(got[:n].permute(0, 3, 1, 2), torch.from_numpy(ref))
    assert not got[n:].any()                                         # rows past the count are zero


# ---------------------------------------------------------------------------------------------
def flat(d):
    rows = [np.hstack([np.full((v.shape[0], 1), float(c)), v]) for c, v in sorted(d.items()) if v.shape[0]]
    return np.vstack(rows) if rows else np.zeros((0, 6))


@pytest.mark.parametrize("n,thr,seed", [(300, 0.05, 1), (300, 0.7, 2), (37, 0.0, 3), (0, 0.05, 4), (300, 0.999, 5)])
def test_detections_vs_oracle(n, thr, seed):
    rng = np.random.RandomState(seed)
    maxr, ncls = 300, 21
    props = np.zeros((maxr, 4), np.float32)
    b = make_boxes(max(n, 1), seed, clusters=12)[:n]
    props[:n] = b
    logits = rng.randn(maxr, ncls).astype(np.float32) * 3
    classes = torch.softmax(torch.from_numpy(logits), dim=1).numpy()
    deltas = rng.randn(maxr, 80).astype(np.float32)
    out = torch.zeros((20, maxr, 5), dtype=torch.float64, device=DEV)
    cnt = torch.full((20,), -1, dtype=torch.int32, device=DEV)
    nr = torch.tensor([n], dtype=torch.int32, device=DEV)
    dp, dc, dd = gpu(props), gpu(classes), gpu(deltas)
    nv.check(nv.lib().frcnn_detections(nv.ptr(dp), nv.ptr(dc), nv.ptr(dd), nv.ptr(nr), maxr, ncls,
                                       600, 1000, thr, 0.3, nv.ptr(out), nv.ptr(cnt), S()), "detections")
    ref = O.detections(props[:n], classes[:n], deltas[:n], 600, 1000, thr)
    got_cnt = cnt.cpu().numpy()
    got = out.cpu().numpy()
    total = 0
    for c in range(1, 21):
        assert got_cnt[c - 1] == ref[c].shape[0], (c, got_cnt[c - 1], ref[c].shape)
        k = ref[c].shape[0]
        total += k
        if k:
            assert np.array_equal(got[c - 1, :k, 4], ref[c][:, 4])                   # scores: exact (float32 widened)
            # boxes: float64 arithmetic identical except exp() ulp -> 1e-9 px
            assert np.abs(got[c - 1, :k, :4] - ref[c][:, :4]).max() <= 1e-9
    print("detections: %d rows over 20 classes (thr %.3g)" % (total, thr))


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["known", "one", "five", "many"])
def test_rpn_targets_vs_oracle_and_reference(golden_dir, name):
    """anchor <-> GT IoU matching (anchors.py:137-262): labels and anchor lists bit-exact, float32 targets to 1e-6."""
    from fasterrcnn_amd.datasets.training_sample import Box
    from fasterrcnn_amd.models import anchors as A
    g = np.load(os.path.join(golden_dir, "small_ops.npz"))
    gt = g["rpn_gt_%s" % name]
    am, vm = O.generate_anchor_maps((3, 600, 1000), (512, 37, 62), 16)
    rmap, obj, bg = A.generate_rpn_map(am, vm, [Box(1, "x", c) for c in gt])
    omap, oobj, obg = O.generate_rpn_map(am, vm, gt)
    assert rmap.shape == (37, 62, 9, 6) and rmap.dtype == np.float32
    assert np.array_equal(rmap[..., 0:2], omap[..., 0:2])                       # trainable / object flags: exact
    assert np.array_equal(obj, oobj) and np.array_equal(bg, obg)                # (y,x,k) lists in the same order
    assert np.array_equal(obj.astype(np.int32), g["rpn_obj_%s" % name]) and len(bg) == int(g["rpn_nbg_%s" % name])
    assert np.array_equal(rmap[..., 2:4], omap[..., 2:4])                       # (gt_c - a_c)/a: float32 IEEE ops
    assert np.abs(rmap[..., 4:6] - omap[..., 4:6]).max() <= 1e-6                # logf: a few ulp


def test_rpn_targets_ragged_map_and_errors():
    from fasterrcnn_amd.datasets.training_sample import Box
    from fasterrcnn_amd.models import anchors as A
    am, vm = O.generate_anchor_maps((3, 333, 517), (512, 20, 32), 16)
    gt = np.array([[10, 20, 300, 400], [100, 100, 180, 260], [0, 0, 332, 516]], dtype=np.float32)
    rmap, obj, bg = A.generate_rpn_map(am, vm, [Box(1, "x", c) for c in gt])
    omap, oobj, obg = O.generate_rpn_map(am, vm, gt)
    assert np.array_equal(rmap[..., 0:4], omap[..., 0:4]) and np.array_equal(obj, oobj) and np.array_equal(bg, obg)
    with pytest.raises(ValueError):
        A.generate_rpn_map(am, vm, [])


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("h,w,min_side,flip,bgr", [(375, 500, 600, False, True), (500, 375, 600, True, True),
                                                   (1200, 1600, 600, False, False), (333, 517, None, False, True),
                                                   (97, 211, 600, True, False)])
def test_preprocess_vs_pil_and_oracle(h, w, min_side, flip, bgr):
    """datasets/image.py:89-100 + :43-57: resized 8-bit pixels == PIL bit for bit, float tensor == oracle."""
    from PIL import Image
    from fasterrcnn_amd.datasets import image as I
    rng = np.random.RandomState(h * 3 + w)
    img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    params = (I.PreprocessingParams(I.ChannelOrder.BGR, 1.0, [103.939, 116.779, 123.680], [1, 1, 1]) if bgr else
              I.PreprocessingParams(I.ChannelOrder.RGB, 1.0 / 255.0, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]))
    out, sf, shape, resized = I.preprocess_image(img, params, min_side, flip, return_resized=True)
    pil = Image.fromarray(img, mode="RGB")
    if flip:
        pil = pil.transpose(method=Image.FLIP_LEFT_RIGHT)
    if min_side is not None:
        f = I._compute_scale_factor(pil.width, pil.height, min_side)
        pil = pil.resize((int(pil.width * f), int(pil.height * f)), resample=Image.BILINEAR)
        assert sf == f
    assert shape == (3, h, w)
    assert np.array_equal(resized.cpu().numpy(), np.array(pil))                         # PIL itself: bit-exact
    ref = O.preprocess_image(img, bgr, params.scaling, params.means, params.stds, min_side, flip)
    assert tuple(out.shape) == ref.shape
    assert np.array_equal(out.cpu().numpy(), ref)                                       # float32 sequence: exact
