"""
GPU parity tests of the ONE-launch Winograd F(2x2,3x3) float32 layer (csrc/winofused.hip: all 16 positions in MFMA
accumulators, input transform at operand-read time, output transform + bias + ReLU + pool on the accumulators).

Tolerances (float32 path): filter bank == the three-launch form's bank bit for bit (another order); one layer against a
float64 convolution <= 5x the direct exact-f32 kernel's own error + 3e-6 of max|y|, and within 4e-6 of max|y| of the
three-launch Winograd form (same transforms in the same float32 operation order; the channel sum is associated differently).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fasterrcnn_amd import _native as nv

pytestmark = pytest.mark.gpu


def fused_bank(w_oihw, scale=None):
    cout, cin = int(w_oihw.shape[0]), int(w_oihw.shape[1])
    u = torch.full((16 * cout * cin,), float("nan"), device=w_oihw.device)
    nv.check(nv.lib().frcnn_pack_conv3x3_winograd_fused(nv.ptr(w_oihw), nv.ptr(scale), nv.ptr(u), cout, cin, nv.stream_ptr()),
             "pack_winograd_fused")
    return u


def run_fused(x, w_oihw, b, relu, pool, u=None):
    lib = nv.lib()
    h, wd, cin = (int(v) for v in x.shape)
    cout = int(w_oihw.shape[0])
    oh, ow = (h // 2, wd // 2) if pool else (h, wd)
    y = torch.full((oh, ow, cout), float("nan"), device=x.device)
    if u is None:
        u = fused_bank(w_oihw)
    flags = (nv.RELU if relu else 0) | (nv.POOL2 if pool else 0)
    nv.check(lib.frcnn_conv3x3_nhwc_winograd_fused(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), h, wd, cin, cout, flags,
                                                   nv.stream_ptr()), "conv_winograd_fused")
    torch.cuda.synchronize()
    return y


def run_other(kind, x, w_oihw, b, relu, pool):
    lib = nv.lib()
    h, wd, cin = (int(v) for v in x.shape)
    cout = int(w_oihw.shape[0])
    s = nv.stream_ptr()
    oh, ow = (h // 2, wd // 2) if pool else (h, wd)
    y = torch.full((oh, ow, cout), float("nan"), device=x.device)
    flags = (nv.RELU if relu else 0) | (nv.POOL2 if pool else 0)
    if kind == "winograd":
        u = torch.empty((16, cout, cin), device=x.device)
        nv.check(lib.frcnn_pack_conv3x3_winograd(nv.ptr(w_oihw), None, nv.ptr(u), cout, cin, s), "pack_winograd")
        wsb = int(lib.frcnn_conv3x3_winograd_workspace_bytes(1, h, wd, cin, cout))
        ws = torch.empty((wsb // 4,), device=x.device)
        nv.check(lib.frcnn_conv3x3_nhwc_winograd(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), 1, h, wd, cin, cout, flags,
                                                 nv.ptr(ws), wsb, s), "conv_winograd")
    else:
        wp = torch.empty((9, cout, cin), device=x.device)
        nv.check(lib.frcnn_pack_conv3x3(nv.ptr(w_oihw), nv.ptr(wp), cout, cin, s), "pack")
        wsb = int(lib.frcnn_conv3x3_workspace_bytes(h, wd, cin, cout))
        ws = torch.empty((max(wsb, 4) // 4,), device=x.device)
        nv.check(lib.frcnn_conv3x3_nhwc(nv.ptr(x), nv.ptr(wp), nv.ptr(b), nv.ptr(y), h, wd, cin, cout, flags,
                                        nv.ptr(ws), wsb, s), "conv")
    torch.cuda.synchronize()
    return y


def test_fused_bank_is_a_permutation_of_the_three_launch_bank():
    lib = nv.lib()
    gen = torch.Generator().manual_seed(3)
    cout, cin = 192, 48
    w = torch.randn((cout, cin, 3, 3), generator=gen).cuda()
    scale = (torch.rand((cout,), generator=gen) + 0.5).cuda()
    for sc in (None, scale):
        u3 = torch.empty((16, cout, cin), device="cuda")
        # (the three-launch pack itself has no cout % 128 restriction; only its GEMM has)
        nv.check(lib.frcnn_pack_conv3x3_winograd(nv.ptr(w), nv.ptr(sc), nv.ptr(u3), cout, cin, nv.stream_ptr()), "pack_winograd")
        # [chunk][cout block][slot][64][16]; slot of position (i, j) = its stage (i & 1, j >> 1), then the wave group i >> 1, then j & 1
        uf = fused_bank(w, sc).reshape(cin // 16, cout // 64, 16, 64, 16)
        slot = [(((i & 1) * 2 + (j >> 1)) * 2 + (i >> 1)) * 2 + (j & 1) for i in range(4) for j in range(4)]
        want = torch.empty_like(uf)
        want[:, :, slot] = u3.reshape(16, cout // 64, 64, cin // 16, 16).permute(3, 1, 0, 2, 4)
        assert sorted(slot) == list(range(16)) and torch.equal(uf, want)
    # from the tap-major master pack (train step): the same bank; data-gradient bank = bank of the rotated, transposed filter
    wp = w.permute(2, 3, 0, 1).reshape(9, cout, cin).contiguous()
    ut = torch.full((16 * cout * cin,), float("nan"), device="cuda")
    nv.check(lib.frcnn_pack_conv3x3_winograd_fused_taps(nv.ptr(wp), nv.ptr(ut), cout, cin, 0, nv.stream_ptr()), "pack_taps")
    assert torch.equal(ut, fused_bank(w))
    cout2, cin2 = 64, 128                                                      # dgrad: output channels = cin2 (% 64), input = cout2 (% 16)
    w2 = torch.randn((cout2, cin2, 3, 3), generator=gen).cuda()
    wp2 = w2.permute(2, 3, 0, 1).reshape(9, cout2, cin2).contiguous()
    ud = torch.full((16 * cout2 * cin2,), float("nan"), device="cuda")
    nv.check(lib.frcnn_pack_conv3x3_winograd_fused_taps(nv.ptr(wp2), nv.ptr(ud), cout2, cin2, 1, nv.stream_ptr()), "pack_taps_dgrad")
    w_rot = w2.flip(2, 3).permute(1, 0, 2, 3).contiguous()                     # [cin2][cout2][3][3]
    assert torch.equal(ud, fused_bank(w_rot))
    # unsupported shapes are refused
    bad = torch.empty((16 * 40 * 48,), device="cuda")
    assert lib.frcnn_pack_conv3x3_winograd_fused(nv.ptr(w), None, nv.ptr(bad), 40, 48, nv.stream_ptr()) == -4
    assert lib.frcnn_pack_conv3x3_winograd_fused(nv.ptr(w), None, nv.ptr(bad), 96, 40, nv.stream_ptr()) == -4


@pytest.mark.parametrize("h,w,cin,cout,relu,pool", [
    (37, 62, 512, 512, True, False),      # block 5 / RPN trunk: 19 x 31 tiles (ragged in both block dimensions)
    (75, 125, 256, 512, True, False),     # conv4_1: odd height and width
    (75, 125, 512, 512, True, True),      # conv4_3 with the fused pool (floor: 37 x 62)
    (150, 250, 256, 256, True, True),     # conv3_3
    (300, 500, 64, 128, True, False),     # conv2_1: four chunks only
    (120, 200, 64, 64, True, True),       # conv1_2's channel shape, two cout blocks
    (9, 11, 256, 128, False, False),      # tiny, odd, no ReLU (negative values must survive)
    (2, 2, 16, 64, True, True),           # a single tile, a single pooled pixel, one chunk, one cout block
    (1, 5, 32, 64, True, False),          # one row
    (4, 32, 48, 192, False, True),        # exactly one block of tiles, three chunks (odd), three cout blocks
    (5, 33, 16, 64, True, False),         # one pixel more than a block in both directions
    (38, 63, 1024, 1024, True, False),    # ResNet's RPN trunk: 64 chunks, 16 cout blocks on four XCD groups
])
def test_layer_against_float64_direct_and_three_launch_form(h, w, cin, cout, relu, pool):
    gen = torch.Generator().manual_seed(h * 1000 + w + cin)
    x = torch.randn((h, w, cin), generator=gen)
    wt = torch.randn((cout, cin, 3, 3), generator=gen) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn((cout,), generator=gen) * 0.1
    ref = F.conv2d(x.permute(2, 0, 1).unsqueeze(0).double(), wt.double(), b.double(), padding=1)
    if relu:
        ref = F.relu(ref)
    if pool:
        ref = F.max_pool2d(ref, 2, 2)
    ref = ref[0].permute(1, 2, 0).numpy()
    xd, wd_, bd = x.cuda(), wt.cuda(), b.cuda()
    yf = run_fused(xd, wd_, bd, relu, pool).cpu().numpy().astype(np.float64)
    assert yf.shape == ref.shape and np.isfinite(yf).all()          # every output written
    scale = max(float(np.abs(ref).max()), 1e-30)
    ef = float(np.abs(yf - ref).max()) / scale
    ed = e3 = 0.0
    if cout % 64 == 0 and h >= 2 and w >= 2:
        yd = run_other("direct", xd, wd_, bd, relu, pool).cpu().numpy().astype(np.float64)
        ed = float(np.abs(yd - ref).max()) / scale
    if cout % 128 == 0:
        y3 = run_other("winograd", xd, wd_, bd, relu, pool).cpu().numpy().astype(np.float64)
        e3 = float(np.abs(y3 - yf).max()) / scale
        assert e3 <= 4e-6
    print("fused winograd %dx%d %d->%d pool=%d: max err / max|y| = %.3g (direct kernel %.3g, vs three-launch form %.3g)" % (
        h, w, cin, cout, pool, ef, ed, e3))
    assert ef <= 5 * ed + 3e-6
    # run-to-run identical; the output buffer is not read (NaN-filled before every run)
    yf2 = run_fused(xd, wd_, bd, relu, pool).cpu().numpy().astype(np.float64)
    assert np.array_equal(yf, yf2)


def test_zero_padding_and_locality():
    """A single non-zero input pixel only reaches its 3x3 output neighbourhood, at every image corner / block seam; the
    response equals the (flipped) filter taps up to the Winograd rounding."""
    gen = torch.Generator().manual_seed(11)
    h, w, cin, cout = 19, 41, 16, 64
    wt = (torch.randn((cout, cin, 3, 3), generator=gen) * 0.2).cuda()
    b = torch.zeros((cout,), device="cuda")
    u = fused_bank(wt)
    for (py, px) in [(0, 0), (0, 40), (18, 0), (18, 40), (3, 31), (4, 32), (5, 33), (3, 16), (7, 0)]:
        x = torch.zeros((h, w, cin), device="cuda")
        x[py, px, 5] = 1.0
        y = run_fused(x, wt, b, False, False, u=u).cpu()
        want = torch.zeros((h, w, cout))
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                oy, ox = py + dy, px + dx
                if 0 <= oy < h and 0 <= ox < w:
                    want[oy, ox] = wt[:, 5, 1 - dy, 1 - dx].cpu()
        assert float((y - want).abs().max()) <= 2e-6, (py, px)


def test_unsupported_shapes_and_arguments():
    lib = nv.lib()
    x = torch.zeros((8, 8, 24), device="cuda")
    u = torch.zeros((16 * 64 * 32,), device="cuda")
    b = torch.zeros((64,), device="cuda")
    y = torch.zeros((8, 8, 64), device="cuda")
    s = nv.stream_ptr()
    assert lib.frcnn_conv3x3_nhwc_winograd_fused(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), 8, 8, 24, 64, 0, s) == -4    # cin % 16
    assert lib.frcnn_conv3x3_nhwc_winograd_fused(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), 8, 8, 32, 96, 0, s) == -4    # cout % 64
    assert lib.frcnn_conv3x3_nhwc_winograd_fused(None, nv.ptr(u), nv.ptr(b), nv.ptr(y), 8, 8, 32, 64, 0, s) == -1
    assert lib.frcnn_conv3x3_nhwc_winograd_fused(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), 1, 8, 32, 64, nv.POOL2, s) == -1
    assert bool(lib.frcnn_conv3x3_uses_winograd_fused(64, 64)) and bool(lib.frcnn_conv3x3_uses_winograd_fused(512, 512))
    assert not lib.frcnn_conv3x3_uses_winograd_fused(3, 64) and not lib.frcnn_conv3x3_uses_winograd_fused(64, 48)
    for cin in (3, 16, 64, 120, 128, 1024):
        for cout in (32, 64, 80, 96, 128, 512):
            assert bool(lib.frcnn_conv3x3_uses_winograd_fused(cin, cout)) == nv.uses_winograd_fused(cin, cout)
