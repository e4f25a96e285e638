"""
CPU checks around the Winograd path (no GPU): the float32 CPU model of csrc/winograd.hip that the parity probe uses
(tests/winograd_parity_probe.py) is a convolution, and bench.py's FLOP accounting of the f32_winograd mode is consistent.
"""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("h,w,cin,cout", [(8, 8, 8, 4), (7, 9, 5, 3), (1, 6, 4, 2), (12, 5, 16, 8)])
def test_cpu_model_of_the_winograd_layers_is_a_convolution(h, w, cin, cout):
    probe = load("winograd_parity_probe", os.path.join(ROOT, "tests", "winograd_parity_probe.py"))
    g = torch.Generator().manual_seed(h * 100 + w)
    x = torch.randn((1, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, 3, 3), generator=g) * 0.2
    b = torch.randn((cout,), generator=g)
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    scale = float(ref.abs().max())
    for fn, tol in ((probe.winograd_conv3x3, 2e-6), (probe.winograd4_conv3x3, 2e-5), (probe.pertap_conv3x3, 2e-6)):
        y = fn(x, wt, b)
        assert y.shape == ref.shape and y.dtype == torch.float32
        assert float((y.double() - ref).abs().max()) <= tol * scale, fn.__name__


def test_transform_matrices_satisfy_the_winograd_identity():
    """A^T [(G g) .* (B^T d)] == correlation of d with g, for the 1-D F(2,3) and F(4,3) matrices in exact rational arithmetic."""
    probe = load("winograd_parity_probe", os.path.join(ROOT, "tests", "winograd_parity_probe.py"))
    rng = np.random.RandomState(3)
    for bt, gm, at, m in ((probe.BT, probe.G, probe.AT, 2), (probe.BT4, probe.G4, probe.AT4, 4)):
        bt, gm, at = (np.asarray(v, dtype=np.float64) for v in (bt.numpy(), gm.numpy(), at.numpy()))
        d = rng.randint(-8, 9, size=(m + 2,)).astype(np.float64)
        g = rng.randint(-8, 9, size=(3,)).astype(np.float64)
        y = at @ ((gm @ g) * (bt @ d))
        ref = np.array([np.dot(d[i:i + 3], g) for i in range(m)])
        assert np.abs(y - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1.0)


def test_bench_flop_accounting():
    bench = load("bench_module", os.path.join(ROOT, "bench.py"))
    from fasterrcnn_amd import _native as nv
    total = bench.total_flops_per_image()
    assert abs(total - 4.4922e11) <= 2e7                                    # 4.4922e11 FLOP per image
    conv1_1 = 2.0 * 27 * 64 * 600 * 1000
    assert abs(bench.executed_mfma_flops_per_image("f32") - (total - conv1_1)) <= 1.0
    wl, dl = bench.winograd_layers("f32_winograd"), bench.direct_layers("f32_winograd")
    # round 2: the one-launch Winograd layer has no V / M traffic, so every 3x3 layer from conv1_2 on is a Winograd layer
    assert len(wl) == 13 and len(dl) == 0 and len(bench.direct_layers("f32")) == 13 and not bench.winograd_layers("f32")
    for ci, co, h, w in bench._MFMA_CONVS:
        assert bench.uses_winograd(ci, co) == nv.uses_winograd_fused(ci, co)
    # the Winograd GEMMs execute 16 multiplies per 2x2 outputs instead of 36: 2.25x less, minus the padding of odd maps
    direct = sum(2.0 * 9 * ci * co * h * w for ci, co, h, w in wl)
    wino = sum(bench.winograd_gemm_flops(*l) for l in wl)
    assert 2.1 <= direct / wino <= 2.25
    assert bench.executed_mfma_flops_per_image("f32_winograd") < 0.54 * bench.executed_mfma_flops_per_image("f32")
