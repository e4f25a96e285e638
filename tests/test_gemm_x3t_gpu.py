"""
GPU parity tests of the "f32x3" arithmetic: the tile-record GEMM with two fp16 terms per row-scaled operand and three fp16 MFMAs per
product (csrc/gemm_x3t.hip), the Winograd F(2x2,3x3) layer on it (csrc/wino_x3.hip) and RoI pooling into its records (csrc/roipool.hip)
-- models/vgg16.py:89-96,129-133, models/rpn.py:88, models/detector.py:65-72 of the reference.

Tolerances.  The split keeps 22-23 bits of every operand relative to its ROW's largest element (|x 2^e - hi - lo| <= 2^-22 |x 2^e|,
checked through the record layout, with rows 2^+-30 apart in magnitude); a GEMM against float64 truth is no worse than 1.5x the exact-f32
MFMA kernel's own error + 2e-7 of max|y| (the bar of the f32x6 kernels: tests/test_wino_x6_gpu.py); a layer against a float64 convolution
no worse than 1.5x the one-launch float32 Winograd layer's error + 2e-7.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fasterrcnn_amd import _native as nv
from fasterrcnn_amd import synthetic
from fasterrcnn_amd.models import vgg16 as V

pytestmark = pytest.mark.gpu


def pad_to(n, t):
    return (n + t - 1) // t * t


def pack_x3t(a, rows_padded):
    """(batches, R, K) or (R, K) float32 CUDA -> (blob int8, record bytes per batch): records of every batch, then the scale arrays."""
    lib = nv.lib()
    if a.dim() == 2:
        a = a.unsqueeze(0)
    a = a.contiguous()
    nb, r, k = (int(v) for v in a.shape)
    per = int(lib.frcnn_x3t_record_bytes(rows_padded, k))
    blob = torch.full((int(lib.frcnn_x3t_blob_bytes(rows_padded, k, nb)),), 0x5B, dtype=torch.int8, device=a.device)
    assert blob.numel() == nb * (per + 4 * rows_padded)
    nv.check(lib.frcnn_pack_rows_x3t(nv.ptr(a), k, r * k, nv.ptr(blob), r, rows_padded, k, nb, nv.stream_ptr()), "pack_rows_x3t")
    return blob, per


def blob_planes(blob, per, rows_padded, k, nb):
    """-> hi, lo float64 (nb, rows_padded, k) and inv float32 (nb, rows_padded)."""
    raw = blob.cpu().numpy().view(np.uint8)
    rec = raw[:nb * per].view(np.float16).reshape(nb, k // 16, rows_padded // 32, 2, 2, 32, 8)
    f = rec.astype(np.float64).transpose(0, 3, 2, 5, 1, 4, 6).reshape(nb, 2, rows_padded, k)      # [b][term][rb, row][chunk, khalf, 8]
    inv = raw[nb * per:].view(np.float32).reshape(nb, rows_padded)
    return f[:, 0], f[:, 1], inv


def gemm_x3t(a_blob, a_per, a_rows, b_blob, b_per, b_rows, bias, m, n, k, batches, relu, shared_b=False, residual=None):
    lib = nv.lib()
    c = torch.full((batches, m, n), float("nan"), device="cuda")
    wsb = int(lib.frcnn_gemm_x3t_workspace_bytes(m, n, k, batches))
    ws = torch.empty((max(wsb, 4) // 4,), device="cuda")
    nb_b = 1 if shared_b else batches
    nv.check(lib.frcnn_gemm_x3t(nv.ptr(a_blob), a_blob.data_ptr() + batches * a_per, a_rows, a_per, a_rows,
                                nv.ptr(b_blob), b_blob.data_ptr() + nb_b * b_per, b_rows, 0 if shared_b else b_per, 0 if shared_b else b_rows,
                                nv.ptr(bias), nv.ptr(residual), nv.ptr(c), n, m * n, m, n, k, batches, nv.RELU if relu else 0, nv.ptr(ws), wsb,
                                nv.stream_ptr()), "gemm_x3t")
    torch.cuda.synchronize()
    return c


def test_x3t_split_scales_and_layout():
    gen = torch.Generator().manual_seed(1)
    a = torch.randn((2, 70, 48), generator=gen) * torch.exp2(torch.randint(-30, 31, (2, 70, 1), generator=gen).float())   # rows 2^+-30 apart
    a[0, 0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e-30, 65504.0, 1e-3, 0.1])
    a[1, 5] = 0.0                                                                                                        # a zero row
    blob, per = pack_x3t(a.cuda(), 96)
    assert per == 3 * 3 * 2048
    hi, lo, inv = blob_planes(blob, per, 96, 48, 2)
    x = a.numpy().astype(np.float64)
    for b in range(2):
        e = np.log2(inv[b].astype(np.float64))
        assert np.array_equal(e, np.round(e))                                    # exact powers of two
        assert (inv[b, 70:] == 1.0).all() and (hi[b, 70:] == 0).all() and (lo[b, 70:] == 0).all()
        scaled = x[b] / inv[b, :70, None]                                        # x 2^e
        rowmax = np.abs(scaled).max(axis=1)
        live = rowmax > 0
        assert (rowmax[live] >= 2.0 ** 14).all() and (rowmax[live] < 2.0 ** 15).all()
        err = np.abs(hi[b, :70] + lo[b, :70] - scaled)
        # 2^-22 relative where the low term is a normal fp16 number, 2^-25 absolute (half an ulp of the smallest subnormal) below
        assert (err <= np.maximum(2.0 ** -22 * np.abs(scaled), 2.0 ** -25)).all()
        assert np.abs(lo[b, :70]).max() <= 2.0 ** -10 * rowmax.max()
    assert inv[1, 5] == 1.0 and (hi[1, 5] == 0).all()


@pytest.mark.parametrize("n,h,w,c,stride", [
    (3, 7, 7, 1024, 1),       # block0.conv1 of the per-RoI layer4 (147 rows: a ragged last row block)
    (5, 7, 7, 1024, 2),       # its downsample: rows = every other pixel
    (2, 4, 4, 2048, 1),       # a 2048-channel input (eight 1 KB reads per row)
    (1, 9, 13, 16, 1), (1, 5, 6, 272, 2),      # one chunk; a channel count that is no multiple of 256
])
def test_split_pixels_x3t_one_launch_form_is_the_two_launch_form_bit_for_bit(n, h, w, c, stride):
    """frcnn_split_pixels_x3t with d_cmax = NULL (split_pixels_x3t_max_kernel: the row maxima reduced in the same launch -- the per-RoI
    head's 1x1 convolutions, models/resnet.py:94-118) writes the records and scales of frcnn_pixel_absmax + frcnn_split_pixels_x3t."""
    lib = nv.lib()
    g = torch.Generator().manual_seed(n * 1000 + c + stride)
    x = torch.randn((n, h, w, c), generator=g) * torch.exp2(torch.randint(-20, 20, (n, h, w, 1), generator=g).float())
    x[0, 0, 0] = 0.0                                                           # a zero row: scale 1
    x = x.cuda().contiguous()
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    rows = n * ho * wo
    rp = pad_to(rows, 32)
    per = int(lib.frcnn_x3t_record_bytes(rp, c))
    out = []
    for fused in (False, True):
        rec = torch.full((per,), 0x5B, dtype=torch.int8, device="cuda")
        inv = torch.full((rp,), float("nan"), device="cuda")
        cmax = torch.empty((n * h * w,), device="cuda")
        if not fused:
            nv.check(lib.frcnn_pixel_absmax(nv.ptr(x), nv.ptr(cmax), n * h * w, c, nv.stream_ptr()), "pixel_absmax")
        nv.check(lib.frcnn_split_pixels_x3t(nv.ptr(x), None if fused else nv.ptr(cmax), nv.ptr(rec), nv.ptr(inv), n, h, w, c, stride, rp,
                                            nv.stream_ptr()), "split_pixels_x3t")
        torch.cuda.synchronize()
        out.append((rec.cpu().numpy(), inv.cpu().numpy()))
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])
    assert np.isfinite(out[1][1]).all()


@pytest.mark.parametrize("M,N,K,batches,relu", [
    (2394, 512, 512, 16, False),     # the position GEMMs of conv4_2 / conv4_3
    (589, 512, 512, 16, False),      # conv5_x / RPN trunk
    (300, 4096, 4096, 1, True),      # fc2's shape (split-K)
    (300, 512, 25088, 1, True),      # fc1's reduction depth
    (64, 4096, 25088, 1, True),      # fc1's split (16 x 98 chunks): the 32-k-stage instantiation of the kernel
    (137, 260, 80, 3, False),        # ragged M and N, five 16-k stages, batches
    (1, 4, 16, 1, True),             # one row, one stage
    (321, 256, 32, 2, False),        # one row more than a tile, two stages
])
def test_gemm_x3t_against_float64_and_the_exact_f32_kernel(M, N, K, batches, relu):
    gen = torch.Generator().manual_seed(M + N + K)
    a = torch.randn((batches, M, K), generator=gen) * torch.exp(torch.randn((batches, 1, K), generator=gen))   # per-channel scales
    a = a * torch.exp2(torch.randint(-12, 13, (batches, M, 1), generator=gen).float())                        # rows of very different size
    w = torch.randn((batches, N, K), generator=gen) * (2.0 / K) ** 0.5
    b = torch.randn((N,), generator=gen) * 0.1
    ref = torch.einsum("bmk,bnk->bmn", a.double(), w.double()) + b.double()
    if relu:
        ref = ref.clamp(min=0)
    ad, wd, bd = a.cuda(), w.cuda(), b.cuda()
    a_rows, b_rows = pad_to(M, nv.X6T_ROW_TILE), pad_to(N, nv.X6T_COL_TILE)
    a_blob, a_per = pack_x3t(ad, a_rows)
    w_blob, w_per = pack_x3t(wd, b_rows)
    c = gemm_x3t(a_blob, a_per, a_rows, w_blob, w_per, b_rows, bd, M, N, K, batches, relu)
    assert not torch.isnan(c).any()
    # per-ROW error: the rows differ by 2^24 in magnitude, each must be accurate relative to its own scale
    row_scale = ref.abs().amax(dim=2, keepdim=True).clamp(min=1e-30)
    e3 = float(((c.cpu().double() - ref).abs() / row_scale).max())
    npad = pad_to(N, 128)
    wpad = torch.zeros((npad, K), device="cuda")
    wpad[:N] = wd[0]
    y32 = V.linear(ad[0].contiguous(), wpad, bd, N, relu)
    e32 = float(((y32.cpu().double() - ref[0]).abs() / row_scale[0]).max())
    print("gemm_x3t M=%d N=%d K=%d x%d: max err / row max|y| = %.3g (exact-f32 MFMA kernel, batch 0: %.3g)" % (M, N, K, batches, e3, e32))
    assert e3 <= 1.5 * e32 + 2e-7 and e3 <= 4e-6 * np.sqrt(K)
    c2 = gemm_x3t(a_blob, a_per, a_rows, w_blob, w_per, b_rows, bd, M, N, K, batches, relu)
    assert torch.equal(c, c2)                                   # deterministic (fixed-order split-K)
    if batches > 1:
        w0_blob, w0_per = pack_x3t(wd[0], b_rows)
        c3 = gemm_x3t(a_blob, a_per, a_rows, w0_blob, w0_per, b_rows, bd, M, N, K, batches, relu, shared_b=True)
        ref3 = torch.einsum("bmk,nk->bmn", a.double(), w[0].double()) + b.double()
        if relu:
            ref3 = ref3.clamp(min=0)
        rs3 = ref3.abs().amax(dim=2, keepdim=True).clamp(min=1e-30)
        assert float(((c3.cpu().double() - ref3).abs() / rs3).max()) <= 1.5 * e32 + 4e-7
    if M == 137:
        res = torch.randn((batches, M, N), generator=gen).cuda()
        c4 = gemm_x3t(a_blob, a_per, a_rows, w_blob, w_per, b_rows, bd, M, N, K, batches, True, residual=res)
        want = (torch.einsum("bmk,bnk->bmn", a.double(), w.double()) + b.double() + res.cpu().double()).clamp(min=0)
        assert float(((c4.cpu().double() - want).abs() / want.abs().amax(dim=2, keepdim=True).clamp(min=1e-30)).max()) <= 1.5 * e32 + 4e-7


def test_gemm_x3t_rejects_bad_arguments():
    lib = nv.lib()
    s = nv.stream_ptr()
    x = torch.zeros((1 << 20,), device="cuda")
    p = nv.ptr(x)
    ok = lambda *a: lib.frcnn_gemm_x3t(*a)
    assert ok(None, p, 320, 0, 0, p, p, 256, 0, 0, None, None, p, 256, 0, 8, 256, 64, 1, 0, p, 1 << 22, s) == -1
    assert ok(p, None, 320, 0, 0, p, p, 256, 0, 0, None, None, p, 256, 0, 8, 256, 64, 1, 0, p, 1 << 22, s) == -1      # no scales
    assert ok(p, p, 300, 0, 0, p, p, 256, 0, 0, None, None, p, 256, 0, 8, 256, 64, 1, 0, p, 1 << 22, s) == -1         # a_rows % 320
    assert ok(p, p, 320, 0, 0, p, p, 128, 0, 0, None, None, p, 256, 0, 8, 256, 64, 1, 0, p, 1 << 22, s) == -1         # b_rows % 256
    assert ok(p, p, 320, 0, 0, p, p, 256, 0, 0, None, None, p, 256, 0, 8, 256, 40, 1, 0, p, 1 << 22, s) == -4         # K % 16
    assert lib.frcnn_x3t_record_bytes(320, 512) == 32 * 10 * 2048 and lib.frcnn_x3t_record_bytes(100, 512) == 0
    assert lib.frcnn_x3t_blob_bytes(320, 512, 2) == 2 * (32 * 10 * 2048 + 1280)


def pack_x3(w_oihw):
    lib = nv.lib()
    cout, cin = int(w_oihw.shape[0]), int(w_oihw.shape[1])
    bank = torch.empty((16, cout, cin), device=w_oihw.device)
    nv.check(lib.frcnn_pack_conv3x3_winograd(nv.ptr(w_oihw), None, nv.ptr(bank), cout, cin, nv.stream_ptr()), "pack_winograd")
    u = torch.full((int(lib.frcnn_conv3x3_winograd_x3_pack_bytes(cout, cin)),), 0x5B, dtype=torch.int8, device=w_oihw.device)
    nv.check(lib.frcnn_pack_conv3x3_winograd_x3(nv.ptr(bank), nv.ptr(u), cout, cin, nv.stream_ptr()), "pack_winograd_x3")
    return u


def run_x3(x, u, b, cout, relu, pool, n_maps=1):
    lib = nv.lib()
    h, wd, cin = (int(v) for v in x.shape[-3:])
    oh, ow = (h // 2, wd // 2) if pool else (h, wd)
    y = torch.full((n_maps, oh, ow, cout) if n_maps > 1 else (oh, ow, cout), float("nan"), device=x.device)
    wsb = int(lib.frcnn_conv3x3_winograd_x3_workspace_bytes(n_maps, h, wd, cin, cout))
    ws = torch.empty((wsb,), dtype=torch.uint8, device=x.device)
    flags = (nv.RELU if relu else 0) | (nv.POOL2 if pool else 0)
    nv.check(lib.frcnn_conv3x3_nhwc_winograd_x3(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), n_maps, h, wd, cin, cout, flags, nv.ptr(ws), wsb,
                                                nv.stream_ptr()), "conv_winograd_x3")
    torch.cuda.synchronize()
    return y


def run_fused_f32(x, w_oihw, b, relu, pool):
    lib = nv.lib()
    h, wd, cin = (int(v) for v in x.shape)
    cout = int(w_oihw.shape[0])
    oh, ow = (h // 2, wd // 2) if pool else (h, wd)
    y = torch.full((oh, ow, cout), float("nan"), device=x.device)
    u = torch.empty((16 * cout * cin,), device=x.device)
    nv.check(lib.frcnn_pack_conv3x3_winograd_fused(nv.ptr(w_oihw), None, nv.ptr(u), cout, cin, nv.stream_ptr()), "pack_winograd_fused")
    flags = (nv.RELU if relu else 0) | (nv.POOL2 if pool else 0)
    nv.check(lib.frcnn_conv3x3_nhwc_winograd_fused(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), h, wd, cin, cout, flags,
                                                   nv.stream_ptr()), "conv_winograd_fused")
    torch.cuda.synchronize()
    return y


@pytest.mark.parametrize("h,w,cin,cout,relu,pool", [
    (75, 125, 512, 512, True, False),     # conv4_2
    (75, 125, 256, 512, True, False),     # conv4_1
    (75, 125, 512, 512, True, True),      # conv4_3 with the fused pool
    (37, 62, 512, 512, True, False),      # block 5 / RPN trunk
    (9, 11, 256, 256, False, False),      # tiny, odd, no ReLU
    (2, 2, 16, 256, True, True),          # a single tile
    (1, 5, 32, 260, True, False),         # one row; cout not a multiple of the column tile
])
def test_x3_layer_against_float64_and_the_float32_winograd_layer(h, w, cin, cout, relu, pool):
    gen = torch.Generator().manual_seed(h * 1000 + w + cin)
    x = torch.randn((h, w, cin), generator=gen) * torch.exp(torch.randn((1, 1, cin), generator=gen))     # per-channel scales
    # a dark corner: activations 2^-20 of the rest of the map -- the per-tile scale must keep its outputs accurate
    x[: max(h // 3, 1), : max(w // 3, 1)] *= 2.0 ** -20
    wt = torch.randn((cout, cin, 3, 3), generator=gen) * (2.0 / (9 * cin)) ** 0.5
    b = torch.zeros((cout,))
    ref = F.conv2d(x.permute(2, 0, 1).unsqueeze(0).double(), wt.double(), b.double(), padding=1)
    if relu:
        ref = ref.clamp(min=0)
    if pool:
        ref = F.max_pool2d(ref, 2)
    ref = ref[0].permute(1, 2, 0)
    xd, wd, bd = x.cuda(), wt.cuda(), b.cuda()
    u = pack_x3(wd)
    y = run_x3(xd, u, bd, cout, relu, pool)
    assert not torch.isnan(y).any()
    scale = float(ref.abs().max())
    e3 = float((y.cpu().double() - ref).abs().max()) / scale
    e32 = float((run_fused_f32(xd, wd, bd, relu, pool).cpu().double() - ref).abs().max()) / scale if cout % 64 == 0 else 1e-6
    print("winograd x3 %dx%d %d->%d: max err / max|y| = %.3g (float32 one-launch Winograd layer %.3g)" % (h, w, cin, cout, e3, e32))
    assert e3 <= 1.5 * e32 + 2e-7
    # the dark corner on its own scale: the outputs of the Winograd TILES whose 4 x 4 input patch is dark throughout (a tile that straddles
    # the edge mixes 2^20-times larger pixels into every V value -- in float32 Winograd as well -- and is not held to the corner's scale)
    dr, dc = max(h // 3, 1), max(w // 3, 1)
    ch = (2 * ((dr - 3) // 2) + 2 if dr >= 3 else 0) // (2 if pool else 1)
    cw = (2 * ((dc - 3) // 2) + 2 if dc >= 3 else 0) // (2 if pool else 1)
    if ch >= 1 and cw >= 1:
        sub, rsub = y[:ch, :cw].cpu().double(), ref[:ch, :cw]
        if float(rsub.abs().max()) > 0:
            ec = float((sub - rsub).abs().max()) / float(rsub.abs().max())
            print("    dark corner (activations x 2^-20): max err / corner max|y| = %.3g" % ec)
            assert ec <= 3e-6
    assert torch.equal(y, run_x3(xd, u, bd, cout, relu, pool))


def test_x3_layer_impulses_and_maps():
    h, w, cin, cout = 22, 70, 32, 256                              # 385 tiles: two GEMM row tiles, 13 record blocks
    gen = torch.Generator().manual_seed(9)
    wt = torch.randn((cout, cin, 3, 3), generator=gen) * 0.1
    b = torch.zeros((cout,))
    u = pack_x3(wt.cuda())
    for (py, px) in ((0, 0), (0, w - 1), (h - 1, 0), (h - 1, w - 1), (1, 63), (2, 64), (19, 33), (10, 35)):
        x = torch.zeros((h, w, cin))
        x[py, px, (py * 7 + px) % cin] = 1.5
        ref = F.conv2d(x.permute(2, 0, 1).unsqueeze(0).double(), wt.double(), None, padding=1)[0].permute(1, 2, 0)
        y = run_x3(x.cuda(), u, b.cuda(), cout, False, False)
        assert float((y.cpu().double() - ref).abs().max()) <= 1e-6, (py, px)
    # several maps in one call == the maps one by one
    xs = torch.randn((3, 9, 13, cin), generator=gen).cuda()
    ys = run_x3(xs, u, b.cuda(), cout, True, False, n_maps=3)
    for i in range(3):
        yi = run_x3(xs[i].contiguous(), u, b.cuda(), cout, True, False)
        assert float((ys[i] - yi).abs().max()) <= 1e-6 * float(yi.abs().max())


def test_roi_pool_x3t_records_hold_the_pooled_values(gpu_model):
    """The f32x3 fc path end to end on the model: with fc_math_mode "f32x3" the pooled features only exist as records; their effect is
    checked through the detector outputs against the exact-f32 fc path on the same proposals (class probabilities within 1e-5) in
    test_linear_x6t_gpu.py.  Here: degenerate RoIs (empty bins, RoIs outside the map) give finite outputs."""
    assert gpu_model.fc_math_mode == "f32x3"
    img = torch.zeros((1, 3, 224, 320), device="cuda")              # constant image: many tied / degenerate proposals
    p, c, d = gpu_model(image_data=img)
    assert torch.isfinite(c).all() and torch.isfinite(d).all()
    assert float((c.sum(dim=1) - 1.0).abs().max()) <= 1e-5


def test_roi_scale_x3t_covers_the_bins_float32_edges():
    """ceil(7 * (roi / 7.0f)) == roi + 1 in float32 for roi = 57, 114, 121, so the last bin of such a RoI reads one row /
    column beyond [rs, rs + roi).  The per-RoI fp16 scale must bound that cell too: with an outlier there (1000x the rest) the f32x3
    records used to overflow to hi = inf, lo = -inf (-> NaN in fc1).  The records must hold the float32 pooling's values to 2^-22."""
    lib = nv.lib()
    fh, fw, c, pooled = 60, 124, 32, 7
    gen = torch.Generator().manual_seed(57)
    fm = torch.rand((fh, fw, c), generator=gen)
    fm[57, :, :] = 1000.0 + 1000.0 * torch.rand((fw, c), generator=gen)      # the row just past a 57-cell RoI that starts at row 0
    fm[:, 114, :] = 1500.0                                                   # the column just past a 114-cell RoI from column 0
    fm[:, 122, :] = 3000.0                                                   # ... past a 121-cell RoI from column 1
    rois = torch.tensor([[0.0, 0.0, 56 * 16.0, 300.0],        # 57 rows: the last bin ends at row 58 (exclusive) in float32
                         [16.0, 0.0, 400.0, 113 * 16.0],      # 114 columns
                         [32.0, 16.0, 300.0, 121 * 16.0],     # 121 columns starting at column 1
                         [100.0, 100.0, 300.0, 400.0]])       # an ordinary RoI
    n, rec_rows, k = 4, 32, pooled * pooled * c
    d_fm, d_rois = fm.cuda(), rois.cuda()
    d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
    out = torch.empty((n, pooled, pooled, c), device="cuda")
    nv.check(lib.frcnn_roi_pool(nv.ptr(d_fm), fh, fw, c, nv.ptr(d_rois), nv.ptr(d_n), n, pooled, 1.0 / 16.0, nv.ptr(out), nv.stream_ptr()), "roi_pool")
    per = int(lib.frcnn_x3t_record_bytes(rec_rows, k))
    blob = torch.zeros((per + 4 * rec_rows,), dtype=torch.int8, device="cuda")
    cmax = torch.empty((fh * fw,), device="cuda")
    nv.check(lib.frcnn_roi_pool_x3t(nv.ptr(d_fm), fh, fw, c, nv.ptr(d_rois), nv.ptr(d_n), n, pooled, 1.0 / 16.0, nv.ptr(cmax),
                                    blob.data_ptr() + per, nv.ptr(blob), rec_rows, nv.stream_ptr()), "roi_pool_x3t")
    torch.cuda.synchronize()
    hi, lo, inv = blob_planes(blob, per, rec_rows, k, 1)
    ref = out.cpu().numpy().astype(np.float64).reshape(n, k)
    assert ref[0].max() >= 1000.0 and ref[1].max() == 1500.0 and ref[2].max() == 3000.0      # the outliers ARE inside the last bins
    assert np.isfinite(hi).all() and np.isfinite(lo).all()
    scaled = ref / inv[0, :n, None]
    assert (np.abs(scaled).max(axis=1) < 2.0 ** 15).all()                                    # every row bounded by its scale
    err = np.abs(hi[0, :n] + lo[0, :n] - scaled)
    assert (err <= np.maximum(2.0 ** -22 * np.abs(scaled), 2.0 ** -25)).all()
    assert (hi[0, n:] == 0).all() and (lo[0, n:] == 0).all()


def test_model_x3_layer_tables_against_the_golden_vectors(gpu_model, golden_dir):
    """Three tables on the 600x1000 golden fixture: the default (the whole x6 table + fc1 / fc2 in f32x3: chosen on the HELD-OUT set against
    the float64 truth, tests/test_holdout_gpu.py), no f32x3 layer, and round 3's table (conv5_1 left in f32x6).  All three are the same
    network to float32 rounding: feature maps within 5e-6 of each other, the SAME proposals in the SAME order as the reference, every row
    within the float32-noise bound, >= 99 % of the rows inside north_star's 1e-3 px (which table puts which row at 0.9e-3 or 1.1e-3 px is
    how the roundings fall -- rounds 2-3 picked the default by exactly that)."""
    assert gpu_model.winograd_x3_layers == nv.DEFAULT_X3_LAYERS_VGG16 == nv.DEFAULT_X6_LAYERS_VGG16
    g = np.load(os.path.join(golden_dir, "vgg16_600x1000_s0.npz"))
    img = synthetic.image(int(g["seed"]), 600, 1000).unsqueeze(0).cuda()
    ref = g["detections"]
    round3 = tuple(n for n in nv.DEFAULT_X6_LAYERS_VGG16 if n != "conv5_1")
    res = {}
    try:
        for name, layers in (("default", nv.DEFAULT_X3_LAYERS_VGG16), ("none", ()), ("round3", round3)):
            gpu_model.winograd_x3_layers = layers
            p, c, d = gpu_model(image_data=img)
            fm = gpu_model.context(0).tensor(0).clone()
            det = gpu_model.predict(image_data=img, score_threshold=0.05)
            res[name] = (p.cpu().numpy(), c.cpu().numpy(), fm, det)
    finally:
        gpu_model.winograd_x3_layers = nv.DEFAULT_X3_LAYERS_VGG16

    for name in ("default", "none", "round3"):
        pr, cl, fm, det = res[name]
        rel = float((fm - res["none"][2]).abs().max()) / float(res["none"][2].abs().max())
        assert pr.shape == g["proposals"].shape
        err = np.abs(pr.astype(np.float64) - g["proposals"]).max(axis=1)            # row by row: same proposals, same order
        n_det, worst_det, n_ours = 0, 0.0, sum(len(v) for v in det.values())
        for c in range(1, 21):
            r = ref[ref[:, 0] == c][:, 1:]
            m = min(len(r), len(det[c]))
            if m:
                e = np.abs(det[c][:m, :4] - r[:m, :4]).max(axis=1)
                worst_det = max(worst_det, float(e.max()))
                n_det += int(((e <= 1e-3) & (np.abs(det[c][:m, 4] - r[:m, 4]) <= 2e-4)).sum())
        print("f32x3 table %-7s: feature map %.3g of max vs the no-f32x3 table; proposals %d/300 within 1e-3 px at their row (worst %.3g, median "
              "%.3g); detections %d/%d (worst row %.3g px, ours %d rows)" % (name, rel, int((err <= 1e-3).sum()), err.max(), np.median(err),
                                                                          n_det, len(ref), worst_det, n_ours))
        assert rel <= 5e-6
        assert err.max() <= 2e-3 and (err <= 1e-3).mean() >= 0.99
        assert n_ours == len(ref) and worst_det <= 2e-3 and n_det >= 0.99 * len(ref)
    with pytest.raises(ValueError):
        gpu_model.winograd_x3_layers = ("conv1_2",)


@pytest.mark.parametrize("n,h,w,cin,cout,relu,pool", [
    (1, 9, 11, 32, 64, True, False),          # one block column, partial tiles
    (1, 38, 66, 64, 64, True, False),
    (1, 150, 250, 64, 128, True, False),      # conv2_1's shape at a quarter of the size
    (1, 75, 125, 256, 256, True, True),       # conv3_3 at half size, fused pool
    (2, 21, 35, 128, 192, False, False),      # two maps, three cout blocks, no ReLU
    (1, 37, 62, 512, 512, True, False),       # conv5_x / the RPN trunk (one-launch in the in-flight slots): 32 chunks, 8 cout blocks
    (1, 38, 62, 256, 512, True, True),        # conv4_3's shape class, fused pool
    (1, 21, 35, 256, 128, True, False),       # filter-resident block order, 4 XCDs per cout block: 6 tile blocks on a grid of 16 (surplus blocks leave)
    (1, 44, 70, 256, 64, True, True),         # ... one cout block shared by all 8 XCDs
])
def test_x3_one_launch_layer_against_the_three_launch_layer_and_float64(n, h, w, cin, cout, relu, pool):
    """csrc/wino_x3f.hip (round 4: the one-launch f32x3 Winograd layer of conv2_2 .. conv3_3): the same per-tile scale, fp16 split and
    accumulation order per (position, tile, channel) as the three-launch layer -- the GEMM part is the same bits -- and an output
    transform that combines the position COLUMNS in registers before the rows (the three-launch layer's wino_output_kernel does rows
    first): the two differ by float32 rounding of the 2 x 2 output sums only.  Held to the three-launch layer within 2e-6 of max|y| and to a
    float64 convolution at the three-launch layer's own bar (<= 1.5 x its error + 2e-7 of max|y|); deterministic."""
    import time
    lib = nv.lib()
    gen = torch.Generator().manual_seed(h * 1000 + w + cin)
    x = (torch.randn((n, h, w, cin), generator=gen) * torch.exp(torch.randn((1, 1, 1, cin), generator=gen))).cuda()
    wt = (torch.randn((cout, cin, 3, 3), generator=gen) * (2.0 / (9 * cin)) ** 0.5).cuda()
    b = (torch.randn((cout,), generator=gen) * 0.1).cuda()
    u = pack_x3(wt)
    want = run_x3(x if n > 1 else x[0].contiguous(), u, b, cout, relu, pool, n_maps=n)
    oh, ow = (h // 2, w // 2) if pool else (h, w)
    y = torch.full((n, oh, ow, cout), float("nan"), device="cuda")
    wsb = int(lib.frcnn_conv3x3_winograd_x3_fused_workspace_bytes(n, h, w))
    ws = torch.empty((wsb,), dtype=torch.uint8, device="cuda")
    flags = (nv.RELU if relu else 0) | (nv.POOL2 if pool else 0)
    call = lambda: nv.check(lib.frcnn_conv3x3_nhwc_winograd_x3_fused(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), n, h, w, cin, cout, flags,
                                                                      nv.ptr(ws), wsb, nv.stream_ptr()), "x3_fused")
    call()
    torch.cuda.synchronize()
    assert not torch.isnan(y).any()
    first = y.clone()
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), b.double(), padding=1)
    if relu:
        ref = ref.clamp(min=0)
    if pool:
        ref = F.max_pool2d(ref, 2)
    ref = ref.permute(0, 2, 3, 1).cpu()
    scale = float(ref.abs().max())
    e1 = float((y.cpu().double() - ref).abs().max()) / scale
    e3 = float((want.reshape(y.shape).cpu().double() - ref).abs().max()) / scale
    diff = float((y.reshape(want.shape) - want).abs().max()) / scale
    t0 = time.perf_counter()
    for _ in range(10):
        call()
    torch.cuda.synchronize()
    print("x3 one-launch %dx%dx%d %d->%d: vs float64 %.3g of max|y| (three-launch layer %.3g), vs the three-launch layer %.3g; %.1f us per call" % (
        n, h, w, cin, cout, e1, e3, diff, (time.perf_counter() - t0) / 10 * 1e6))
    assert diff <= 2e-6 and e1 <= 1.5 * e3 + 2e-7
    assert torch.equal(y, first)                                           # run-to-run identical


@pytest.mark.parametrize("n,h,w,cin,cout,relu,pool", [
    (1, 9, 11, 32, 64, True, False),          # 2 chunks: chunk 0 and one steady-state chunk, partial tiles
    (1, 38, 66, 64, 64, True, False),
    (1, 150, 250, 64, 128, True, False),
    (1, 75, 125, 256, 256, True, True),       # conv3_3 at half size, fused pool
    (2, 21, 35, 128, 192, False, False),      # two maps, three cout blocks, no ReLU
    (1, 37, 62, 512, 512, True, False),       # 32 chunks, 8 cout blocks, filter-resident order
    (1, 38, 62, 256, 512, True, True),
    (1, 21, 35, 256, 128, True, False),       # surplus blocks leave
    (1, 44, 70, 256, 64, True, True),
    (1, 16, 32, 32, 64, True, False),         # exactly one full tile block
])
def test_x3_one_launch_eight_wave_form_is_the_four_wave_form_bit_for_bit(n, h, w, cin, cout, relu, pool):
    """csrc/wino_x3e.hip (round 5): the one-launch f32x3 layer as EIGHT waves, two per SIMD -- wave (i, jp) owns position row i and the position
    columns 2 jp, 2 jp + 1 -- against csrc/wino_x3f.hip's four waves: the same operand formation (per value the same float32 operations), the
    same MFMA order per accumulator and the same summation order in the output transform (the column pass crosses the wave pair through
    LDS): EQUAL outputs and equal emitted channel maxima, forced through FRCNN_X3F_WAVES4 / _WAVES8 on every shape class."""
    lib = nv.lib()
    gen = torch.Generator().manual_seed(7 * h + w + cin + cout)
    x = (torch.randn((n, h, w, cin), generator=gen) * torch.exp(torch.randn((1, 1, 1, cin), generator=gen))).clamp(min=-0.5).cuda()
    wt = (torch.randn((cout, cin, 3, 3), generator=gen) * (2.0 / (9 * cin)) ** 0.5).cuda()
    b = (torch.randn((cout,), generator=gen) * 0.1).cuda()
    u = pack_x3(wt)
    oh, ow = (h // 2, w // 2) if pool else (h, w)
    wsb = int(lib.frcnn_conv3x3_winograd_x3_fused_workspace_bytes(n, h, w))
    ws = torch.empty((wsb,), dtype=torch.uint8, device="cuda")
    base = (nv.RELU if relu else 0) | (nv.POOL2 if pool else 0)
    out = {}
    # round 6: the eight-wave form ships in `make EXPERIMENTS=1` libraries only (FRCNN_LIB_PATH=build/libfrcnn_exp.so); it still forms its
    # operands with round 5's v_fma_mix pairs, so in such a build this test also holds round 6's v_cvt_pk formation of the four-wave
    # kernel to the old one bit for bit
    y = torch.empty((n, oh, ow, cout), device="cuda")
    if lib.frcnn_conv3x3_nhwc_winograd_x3_fused(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), n, h, w, cin, cout, base | nv.X3F_WAVES8,
                                                nv.ptr(ws), wsb, nv.stream_ptr()) == -4:
        pytest.skip("wino_x3e_kernel is not in this build (make EXPERIMENTS=1)")
    for name, force in (("four", nv.X3F_WAVES4), ("eight", nv.X3F_WAVES8)):
        y = torch.full((n, oh, ow, cout), float("nan"), device="cuda")
        nv.check(lib.frcnn_conv3x3_nhwc_winograd_x3_fused(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), n, h, w, cin, cout, base | force,
                                                          nv.ptr(ws), wsb, nv.stream_ptr()), "x3_fused " + name)
        torch.cuda.synchronize()
        assert not torch.isnan(y).any(), name
        out[name] = y
    assert torch.equal(out["four"], out["eight"]), float((out["four"] - out["eight"]).abs().max())
    if relu:    # the channel maxima the epilogue leaves for the next layer (frcnn_conv3x3_nhwc_winograd_x3_chain), given the input's own
        cm_in = torch.empty((n, h, w), device="cuda")
        nv.check(lib.frcnn_pixel_absmax(nv.ptr(x), nv.ptr(cm_in), n * h * w, cin, nv.stream_ptr()), "absmax")
        cms = {}
        for name, force in (("four", nv.X3F_WAVES4), ("eight", nv.X3F_WAVES8)):
            y = torch.empty((n, oh, ow, cout), device="cuda")
            cm = torch.zeros((n, oh, ow), device="cuda")
            nv.check(lib.frcnn_conv3x3_nhwc_winograd_x3_chain(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), n, h, w, cin, cout, base | force, 1,
                                                              nv.ptr(ws), wsb, nv.ptr(cm_in), nv.ptr(cm), nv.stream_ptr()), "x3_chain " + name)
            torch.cuda.synchronize()
            assert torch.equal(y, out[name])
            assert torch.equal(cm, y.amax(dim=3)), name
            cms[name] = cm
        assert torch.equal(cms["four"], cms["eight"])
    # the default choice is one of the two
    y = torch.empty((n, oh, ow, cout), device="cuda")
    nv.check(lib.frcnn_conv3x3_nhwc_winograd_x3_fused(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), n, h, w, cin, cout, base, nv.ptr(ws), wsb,
                                                      nv.stream_ptr()), "x3_fused")
    assert torch.equal(y, out["four"])
    # ... and twice the same
    y2 = torch.empty_like(y)
    nv.check(lib.frcnn_conv3x3_nhwc_winograd_x3_fused(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y2), n, h, w, cin, cout, base | nv.X3F_WAVES8,
                                                      nv.ptr(ws), wsb, nv.stream_ptr()), "x3_fused eight")
    assert torch.equal(y2, out["eight"])


@pytest.mark.parametrize("n,h,w,cin,cout,relu,pool", [
    (1, 9, 11, 64, 128, True, False),         # 4 chunks per pass: first, second, one loop pair (which is also the last), partial tiles
    (1, 38, 66, 64, 128, True, True),
    (1, 150, 250, 128, 128, True, True),      # conv2_2 at half size: 8 chunks, pool, XCD-grouped block order
    (1, 75, 125, 128, 256, True, False),      # conv3_1's class: two cout blocks
    (1, 75, 125, 256, 256, True, True),       # conv3_3 at half size: filter-resident order, fused pool
    (2, 21, 35, 128, 384, False, False),      # two maps, three cout blocks, no ReLU
    (1, 37, 62, 512, 512, True, False),       # conv5_x: 32 chunks per pass, four cout blocks
    (1, 21, 35, 256, 128, True, False),       # surplus blocks leave
    (1, 16, 32, 96, 128, True, False),        # exactly one full tile block, 6 chunks
])
def test_x3_one_launch_pair_form_is_the_four_wave_form_bit_for_bit(n, h, w, cin, cout, relu, pool):
    """csrc/wino_x3p.hip (round 6): the one-launch f32x3 layer with 128 output channels per block in two passes over the input channels -- a
    wave owns HALF a position row per pass, the first pass's accumulators rest in a block-private scratch -- against csrc/wino_x3f.hip: the
    same value per operand (r 2^e is exact: fma(r1, 2^e, r2 2^e) == r1 2^e + r2 2^e rounded once), the same MFMA order per accumulator,
    the same output transform: EQUAL outputs and equal emitted channel maxima (FRCNN_X3F_PAIR against FRCNN_X3F_WAVES4)."""
    lib = nv.lib()
    gen = torch.Generator().manual_seed(11 * h + w + cin + cout)
    x = (torch.randn((n, h, w, cin), generator=gen) * torch.exp(torch.randn((1, 1, 1, cin), generator=gen))).clamp(min=-0.5).cuda()
    wt = (torch.randn((cout, cin, 3, 3), generator=gen) * (2.0 / (9 * cin)) ** 0.5).cuda()
    b = (torch.randn((cout,), generator=gen) * 0.1).cuda()
    u = pack_x3(wt)
    oh, ow = (h // 2, w // 2) if pool else (h, w)
    wsb = int(lib.frcnn_conv3x3_winograd_x3_pair_workspace_bytes(n, h, w, cout))
    if wsb == 0:
        pytest.skip("wino_x3p_kernel is not in this build (make EXPERIMENTS=1; FRCNN_LIB_PATH=build/libfrcnn_exp.so)")
    assert wsb >= int(lib.frcnn_conv3x3_winograd_x3_fused_workspace_bytes(n, h, w)) + 256 * 1024
    ws = torch.empty((wsb,), dtype=torch.uint8, device="cuda")
    base = (nv.RELU if relu else 0) | (nv.POOL2 if pool else 0)
    out = {}
    for name, force in (("four", nv.X3F_WAVES4), ("pair", nv.X3F_PAIR)):
        y = torch.full((n, oh, ow, cout), float("nan"), device="cuda")
        nv.check(lib.frcnn_conv3x3_nhwc_winograd_x3_fused(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), n, h, w, cin, cout, base | force,
                                                          nv.ptr(ws), wsb, nv.stream_ptr()), "x3_fused " + name)
        torch.cuda.synchronize()
        assert not torch.isnan(y).any(), name
        out[name] = y
    assert torch.equal(out["four"], out["pair"]), float((out["four"] - out["pair"]).abs().max())
    if relu:
        cm_in = torch.empty((n, h, w), device="cuda")
        nv.check(lib.frcnn_pixel_absmax(nv.ptr(x), nv.ptr(cm_in), n * h * w, cin, nv.stream_ptr()), "absmax")
        y = torch.empty((n, oh, ow, cout), device="cuda")
        cm = torch.zeros((n, oh, ow), device="cuda")
        nv.check(lib.frcnn_conv3x3_nhwc_winograd_x3_chain(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), n, h, w, cin, cout, base | nv.X3F_PAIR, 1,
                                                          nv.ptr(ws), wsb, nv.ptr(cm_in), nv.ptr(cm), nv.stream_ptr()), "x3_chain pair")
        torch.cuda.synchronize()
        assert torch.equal(y, out["four"])
        assert torch.equal(cm, y.amax(dim=3))
    # twice the same, and a workspace one byte short is refused
    y2 = torch.empty_like(out["pair"])
    nv.check(lib.frcnn_conv3x3_nhwc_winograd_x3_fused(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y2), n, h, w, cin, cout, base | nv.X3F_PAIR,
                                                      nv.ptr(ws), wsb, nv.stream_ptr()), "x3_fused pair")
    assert torch.equal(y2, out["pair"])
    assert lib.frcnn_conv3x3_nhwc_winograd_x3_fused(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y2), n, h, w, cin, cout, base | nv.X3F_PAIR,
                                                    nv.ptr(ws), wsb - 256 * 1024, nv.stream_ptr()) != 0


@pytest.mark.parametrize("one_launch,h,w,cin,cout,pool", [
    (1, 38, 66, 64, 128, False),       # one-launch layer, two cout blocks
    (1, 75, 125, 128, 256, True),      # fused pool, XCD-grouped block order
    (0, 38, 62, 256, 256, False),      # three-launch layer (wino_output_kernel's wave reduction)
    (0, 75, 125, 256, 512, True),      # conv4_3's shape class: fused pool
])
def test_x3_layers_chain_their_channel_maxima(one_launch, h, w, cin, cout, pool):
    """Round 4: an f32x3 layer can leave the per-pixel channel maximum of its OUTPUT behind for the next f32x3 layer (atomic maxima from its
    epilogue into a zeroed buffer) and take its INPUT's maxima from its producer (frcnn_conv3x3_nhwc_winograd_x3_chain; what
    frcnn_vgg16_forward does for conv2_2 ... conv5_3, the RPN trunk and the RoI pooling).  The emitted maxima equal frcnn_pixel_absmax of the
    output bit for bit, and the output does not depend on which way the input's maxima arrived."""
    lib = nv.lib()
    gen = torch.Generator().manual_seed(h + w + cin + cout)
    x = torch.randn((h, w, cin), generator=gen).clamp(min=0).cuda()
    wt = (torch.randn((cout, cin, 3, 3), generator=gen) * (2.0 / (9 * cin)) ** 0.5).cuda()
    b = (torch.randn((cout,), generator=gen) * 0.1).cuda()
    u = pack_x3(wt)
    flags = nv.RELU | (nv.POOL2 if pool else 0)
    oh, ow = (h // 2, w // 2) if pool else (h, w)
    wsb = int(lib.frcnn_conv3x3_winograd_x3_fused_workspace_bytes(1, h, w) if one_launch else lib.frcnn_conv3x3_winograd_x3_workspace_bytes(1, h, w, cin, cout))
    ws = torch.empty((wsb,), dtype=torch.uint8, device="cuda")

    def run(cmax_in, cmax_out):
        y = torch.full((oh, ow, cout), float("nan"), device="cuda")
        nv.check(lib.frcnn_conv3x3_nhwc_winograd_x3_chain(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), 1, h, w, cin, cout, flags, one_launch, nv.ptr(ws), wsb,
                                                          nv.ptr(cmax_in), nv.ptr(cmax_out), nv.stream_ptr()), "x3_chain")
        torch.cuda.synchronize()
        return y
    y0 = run(None, None)
    assert not torch.isnan(y0).any()
    cin_max = torch.empty((h * w,), device="cuda")
    nv.check(lib.frcnn_pixel_absmax(nv.ptr(x), nv.ptr(cin_max), h * w, cin, nv.stream_ptr()), "pixel_absmax")
    cout_max = torch.zeros((oh * ow,), device="cuda")
    y1 = run(cin_max, cout_max)
    assert torch.equal(y0, y1)
    want = torch.empty((oh * ow,), device="cuda")
    nv.check(lib.frcnn_pixel_absmax(nv.ptr(y1), nv.ptr(want), oh * ow, cout, nv.stream_ptr()), "pixel_absmax")
    torch.cuda.synchronize()
    assert float(want.max()) > 0 and torch.equal(cout_max, want)
    assert torch.equal(want.reshape(oh, ow), y1.max(dim=2).values)


def test_inflight_slots_run_the_512_channel_layers_in_the_one_launch_form(gpu_model):
    """FasterRCNNModel.layer_tables: the in-flight slots of predict_async run conv4_1 .. conv5_3 and the RPN trunk as one-launch f32x3 layers;
    slot 0 (forward / predict) does too since round 5 (alone_winograd_x3f_layers: ONE table for every slot) -- this test puts slot 0 back on
    round 4's three launches to compare the two forms on the same blobs (13 launches of timing class 10, none of the x6 classes).  The
    two forms differ by the rounding order of the output transform only: feature maps within 2e-6 of the largest activation, the same
    proposals as rows -- apart by what two float32 evaluations that are each ~1.2e-4 px from the float64 truth differ by (measured: worst
    row 3.4e-4 px), inside north_star's 1e-3 px of each other."""
    from fasterrcnn_amd import synthetic
    img = synthetic.image(5).unsqueeze(0).cuda()
    assert gpu_model.layer_tables(0) == gpu_model.layer_tables(2) and gpu_model.layer_tables(2)[1] == ()          # one table for every slot
    assert set(gpu_model.layer_tables(2)[2]) == set(nv.DEFAULT_X3F_LAYERS_VGG16) | set(nv.DEFAULT_INFLIGHT_X3F_LAYERS_VGG16)
    saved_alone = gpu_model.alone_winograd_x3f_layers
    gpu_model.alone_winograd_x3f_layers = ()
    assert gpu_model.layer_tables(0)[1] == nv.DEFAULT_X3_LAYERS_VGG16
    try:
        _compare_three_launch_slot0_with_one_launch_slot2(gpu_model, img)
    finally:
        gpu_model.alone_winograd_x3f_layers = saved_alone
    # with the default tables the feature extractor of every slot computes the same bits (the slots still differ in the split granularity of
    # the fc GEMMs: inflight_conv_blocks_target)
    with torch.no_grad():
        gpu_model._enqueue(img, None, None, None, 0).result()
        f0 = gpu_model.context(0).tensor(0).clone()
        gpu_model._enqueue(img, None, None, None, 2).result()
        f2 = gpu_model.context(2).tensor(0).clone()
    assert torch.equal(f0, f2)


def _compare_three_launch_slot0_with_one_launch_slot2(gpu_model, img):
    with torch.no_grad():
        p0, c0, d0 = gpu_model._enqueue(img, None, None, None, 0).result()
        fm0 = gpu_model.context(0).tensor(0).clone()
        p1, c1, d1 = gpu_model._enqueue(img, None, None, None, 2).result()
        ctx = gpu_model.context(2)
        fm1 = ctx.tensor(0).clone()
        ctx.timing_enable(True)
        gpu_model._enqueue(img, None, None, None, 2).result()
        torch.cuda.synchronize()
        t = ctx.timing_read(reset=True)
        ctx.timing_enable(False)
    assert t["winograd_x3f"][1] == 13 and t["winograd_x6_gemm"][1] == 0 and t["winograd_x6_transforms"][1] == 0 and t["winograd_gemm"][1] == 0
    rel = float((fm0 - fm1).abs().max()) / float(fm0.abs().max())
    a, b = p0.cpu().numpy(), p1.cpu().numpy()
    assert a.shape == b.shape
    d = np.abs(a[:, None, :] - b[None, :, :]).max(axis=2).min(axis=1)
    print("in-flight table vs slot 0: feature map %.3g of max, proposals %d rows, worst nearest-row distance %.3g px, median %.3g" % (
        rel, len(a), float(d.max()), float(np.median(d))))
    assert rel <= 2e-6
    assert (d <= 1e-3).mean() >= 0.99 and float(np.median(d)) <= 2e-4
    # and the slot's results do not depend on what ran before it
    with torch.no_grad():
        p2, c2, d2 = gpu_model._enqueue(img, None, None, None, 2).result()
    assert torch.equal(p1, p2) and torch.equal(c1, c2) and torch.equal(d1, d2)
