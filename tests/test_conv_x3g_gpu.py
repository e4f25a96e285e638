"""
frcnn_conv_nhwc_x3g / frcnn_tensor_absmax (csrc/conv_gather.hip conv_gather_x3_kernel): the Bottleneck convolutions of
models/resnet.py:38-46 (frozen BN folded) in the f32x3 arithmetic under ONE power-of-two scale per tensor, against the float64
convolution of the same operands and next to the exact-f32 kernel's own distance from it.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from fasterrcnn_amd import _native as nv
from fasterrcnn_amd.models import resnet as R


def pack(w):
    """[cout][cin][k][k] -> the gather kernel's [k*k][cout][cin] float32 pack"""
    cout, cin, k, _ = w.shape
    return w.permute(2, 3, 0, 1).reshape(k * k, cout, cin).contiguous()


def truth(x, w, b, stride, pad, relu, res):
    y = F.conv2d(x.double().permute(0, 3, 1, 2).cpu(), w.double().cpu(), b.double().cpu(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    if res is not None:
        y = y + res.double().cpu()
    return torch.relu(y) if relu else y


CASES = [
    # n, h, w, cin, cout, k, stride, relu, residual            what it covers
    (1, 38, 63, 256, 64, 1, 1, True, False),                   # cout <= 64 tile, split-K
    (1, 38, 63, 64, 256, 1, 1, True, True),                    # residual + relu, 128 x 128 tile
    (1, 37, 61, 128, 128, 3, 1, True, False),                  # 3x3, ragged M
    (1, 37, 61, 128, 128, 3, 2, True, False),                  # stride-2 3x3 (layer2.0 / layer3.0)
    (1, 38, 63, 256, 512, 1, 2, False, False),                 # downsample: stride-2 1x1, no relu
    (2, 150, 250, 64, 64, 1, 1, True, False),                  # many blocks, no split
    (3, 7, 7, 512, 512, 3, 1, True, False),                    # per-RoI maps of layer4, deep K
    (1, 5, 3, 32, 4, 1, 1, False, False),                      # smallest legal shape
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d_%dx%d_%d_%d_k%ds%d" % c[:7])
def test_conv_x3g_against_the_float64_convolution(case):
    n, h, w, cin, cout, k, stride, relu, with_res = case
    g = torch.Generator().manual_seed(hash(case) % (1 << 31))
    x = (torch.randn(n, h, w, cin, generator=g) * torch.exp(2.0 * torch.randn(1, 1, 1, cin, generator=g))).relu().cuda()   # channels of very different size
    wt = (torch.randn(cout, cin, k, k, generator=g) * (1.0 / np.sqrt(cin * k * k)) * torch.exp(1.5 * torch.randn(cout, 1, 1, 1, generator=g))).cuda()
    b = torch.randn(cout, generator=g).cuda()
    pad = 1 if k == 3 else 0
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = torch.randn(n, ho, wo, cout, generator=g).cuda() if with_res else None
    wp = pack(wt)
    xmax, wmax = R.tensor_absmax(x), R.tensor_absmax(wp)
    assert float(xmax) == float(x.abs().max()) and float(wmax) == float(wp.abs().max())
    ymax = torch.zeros(1, device="cuda")
    y, _, _ = R.conv_nhwc_x3g(x, wp, b, n, h, w, cin, cout, k, stride, pad, relu, xmax, wmax, ymax, residual=res)
    y32, _, _ = R.conv_nhwc(x, wp, b, n, h, w, cin, cout, k, stride, pad, relu, residual=res)
    torch.cuda.synchronize()
    assert float(ymax) == float(y.abs().max()), "the epilogue's maximum is the tensor's maximum, exactly"
    yt = truth(x, wt, b, stride, pad, relu, res)
    scale = float(yt.abs().max())
    e3 = float((y.double().cpu() - yt).abs().max()) / scale
    e32 = float((y32.double().cpu() - yt).abs().max()) / scale
    print("x3g %.3g  exact-f32 kernel %.3g  (relative to max|y|)" % (e3, e32))
    assert e3 <= 1e-6, "f32x3 under a tensor scale: ~2^-22 per operand + float32 accumulation"
    assert e3 <= 4.0 * e32 + 2e-7


def test_conv_x3g_upper_bounds_scale_the_same_result():
    """any upper bound of the maxima is a valid scale: a bound 3x the maximum (a different power of two) moves the result by rounding only"""
    g = torch.Generator().manual_seed(5)
    n, h, w, cin, cout = 1, 19, 31, 128, 64
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wp = pack((torch.randn(cout, cin, 1, 1, generator=g) / 11.0).cuda())
    b = torch.zeros(cout).cuda()
    xm, wm = R.tensor_absmax(x), R.tensor_absmax(wp)
    y0, _, _ = R.conv_nhwc_x3g(x, wp, b, n, h, w, cin, cout, 1, 1, 0, False, xm, wm)
    y1, _, _ = R.conv_nhwc_x3g(x, wp, b, n, h, w, cin, cout, 1, 1, 0, False, xm * 3.0, wm * 5.0)
    assert float((y0 - y1).abs().max()) <= 2e-6 * float(y0.abs().max())
    assert torch.isfinite(y1).all()


def test_conv_x3g_saturates_on_a_bound_that_is_not_one():
    """a caller's "maximum" 64x too small must not become hi = inf, lo = -inf -> NaN in the output: the operand saturates at the largest
    finite fp16 hi term (the result is then wrong in value, finite, and the true maximum the epilogue leaves behind shows it)"""
    g = torch.Generator().manual_seed(9)
    n, h, w, cin, cout = 1, 9, 11, 64, 64
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wp = pack((torch.randn(cout, cin, 1, 1, generator=g) / 8.0).cuda())
    b = torch.zeros(cout).cuda()
    ymax = torch.zeros(1, device="cuda")
    before = nv.x3_saturation_count()
    y, _, _ = R.conv_nhwc_x3g(x, wp, b, n, h, w, cin, cout, 1, 1, 0, False, R.tensor_absmax(x), R.tensor_absmax(wp), ymax)
    assert nv.x3_saturation_count() == before, "a true maximum never saturates"
    y, _, _ = R.conv_nhwc_x3g(x, wp, b, n, h, w, cin, cout, 1, 1, 0, False, R.tensor_absmax(x) / 64.0, R.tensor_absmax(wp), ymax)
    assert torch.isfinite(y).all() and torch.isfinite(ymax).all()
    # ... and it is COUNTED (frcnn_x3_saturation_events, ABI 12): the clamp is a reported event, not a silent one
    assert nv.x3_saturation_count() > before


def test_conv_x3g_rejects_missing_scales():
    x = torch.zeros(1, 4, 4, 16).cuda()
    wp = torch.zeros(1, 4, 16).cuda()
    b = torch.zeros(4).cuda()
    y = torch.zeros(1, 4, 4, 4).cuda()
    rc = nv.lib().frcnn_conv_nhwc_x3g(nv.ptr(x), nv.ptr(wp), nv.ptr(b), None, nv.ptr(y), 1, 4, 4, 16, 4, 1, 1, 0, 0, None, None, None, None, 0, nv.stream_ptr())
    assert rc == -1          # FRCNN_EINVAL


@pytest.mark.parametrize("shape", [(1, 38, 63, 256, 1024, 1, 1, 0), (1, 38, 63, 1024, 256, 1, 1, 0), (1, 38, 63, 256, 256, 3, 1, 1),
                                   (1, 75, 125, 512, 256, 3, 2, 1), (2, 19, 31, 512, 128, 1, 1, 0), (1, 9, 11, 2048, 64, 1, 1, 0)])
def test_conv_x3g_split_reduction_finished_in_the_kernel_is_the_same_bits(shape):
    """frcnn_conv_nhwc_x3g_tickets (ABI 15, round 6): the last block of a tile to arrive sums the partial planes in ascending order --
    the bits of frcnn_conv_nhwc_x3g's separate finishing pass, output and emitted maximum alike; the ticket array is zero again
    afterwards, call after call (models/resnet.py:38-46: the small-map Bottleneck convolutions are the ones that split)."""
    n, h, w, cin, cout, k, stride, pad = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wp = pack((torch.randn(cout, cin, k, k, generator=g) / (3.0 * (cin * k * k) ** 0.5)).cuda())
    b = torch.randn(cout, generator=g).cuda()
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = torch.randn(n, ho, wo, cout, generator=g).cuda()
    xm, wm = R.tensor_absmax(x), R.tensor_absmax(wp)
    tickets = torch.zeros(nv.X3G_TILE_COUNTERS, dtype=torch.int32, device="cuda")
    for relu, residual in ((True, res), (False, None)):
        ym0, ym1 = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
        y0, _, _ = R.conv_nhwc_x3g(x, wp, b, n, h, w, cin, cout, k, stride, pad, relu, xm, wm, ym0, residual)
        for _ in range(3):
            ym1.zero_()
            y1, _, _ = R.conv_nhwc_x3g(x, wp, b, n, h, w, cin, cout, k, stride, pad, relu, xm, wm, ym1, residual, tickets=tickets)
            assert torch.equal(y0, y1)
            assert float(ym0) == float(ym1)
            assert int(tickets.abs().max()) == 0


@pytest.mark.parametrize("shape", [(1, 38, 63, 256, 1024, 1, 1, 0), (1, 75, 125, 128, 128, 3, 2, 1), (2, 19, 31, 512, 128, 1, 1, 0), (1, 150, 250, 64, 256, 1, 1, 0)])
def test_conv_x3g_pre_split_weights_are_the_same_bits(shape):
    """frcnn_pack_conv_x3g_weights + FRCNN_X3G_WSPLIT (ABI 15, round 6): the weight pack split once, at pack time, into the kernel's operand
    format -- the convolution then copies a weight piece into LDS instead of splitting it in every block; output and emitted maximum are
    the float32 pack's, bit for bit (models/resnet.py:38-46, frozen BN folded: the weights are constants)."""
    n, h, w, cin, cout, k, stride, pad = shape
    g = torch.Generator().manual_seed(7 + sum(shape))
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wp = pack((torch.randn(cout, cin, k, k, generator=g) / (3.0 * (cin * k * k) ** 0.5)).cuda())
    b = torch.randn(cout, generator=g).cuda()
    xm, wm = R.tensor_absmax(x), R.tensor_absmax(wp)
    ws = R.pack_x3g_weights(wp, wm)
    assert ws.shape == wp.shape and not torch.equal(ws, wp)
    tickets = torch.zeros(nv.X3G_TILE_COUNTERS, dtype=torch.int32, device="cuda")
    for relu in (True, False):
        ym0, ym1 = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
        y0, _, _ = R.conv_nhwc_x3g(x, wp, b, n, h, w, cin, cout, k, stride, pad, relu, xm, wm, ym0)
        y1, _, _ = R.conv_nhwc_x3g(x, ws, b, n, h, w, cin, cout, k, stride, pad, relu, xm, wm, ym1, tickets=tickets, wsplit=True)
        assert torch.equal(y0, y1)
        assert float(ym0) == float(ym1)
