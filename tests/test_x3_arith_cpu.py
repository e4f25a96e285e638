"""
CPU statement of the "f32x3" arithmetic (csrc/x3t.h, csrc/gemm_x3t.hip) in numpy: the row scale, the two-term fp16 split and the
three-product GEMM with float32 accumulation.  Runs without a GPU; tests/test_gemm_x3t_gpu.py checks the kernels against the same bounds.
"""
import numpy as np


def row_scale(mx):
    """(mult, inv) of hx_row_scale: mult = 2^e with mx 2^e in [2^14, 2^15) (e clamped to +-100; 1 for mx == 0)."""
    mx = np.asarray(mx, dtype=np.float32)
    bits = mx.view(np.uint32)
    E = ((bits >> 23) & 0xFF).astype(np.int64) - 127
    e = np.where(mx > 0, np.clip(14 - E, -100, 100), 0)
    return np.exp2(e).astype(np.float32), np.exp2(-e).astype(np.float32)


def split(x):
    """x [R][K] float32 -> hi, lo (float16 values as float32), inv [R]."""
    mult, inv = row_scale(np.abs(x).max(axis=1))
    xs = (x * mult[:, None]).astype(np.float32)                 # exact: a power of two
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32), inv


def gemm_x3(a, w):
    """c[m][n] = (hh + hl + lh) accumulated per 16-k chunk in float32, then * 2^-e(m) * 2^-e(n)."""
    ah, al, ai = split(a)
    wh, wl, wi = split(w)
    acc = np.zeros((a.shape[0], w.shape[0]), dtype=np.float32)
    for k0 in range(0, a.shape[1], 16):
        s = slice(k0, k0 + 16)
        for x, y in ((al, wh), (ah, wl), (ah, wh)):             # the kernel's order within a chunk: lo*hi, hi*lo, hi*hi
            acc = (acc.astype(np.float64) + x[:, s].astype(np.float64) @ y[:, s].astype(np.float64).T).astype(np.float32)
    return (acc * ai[:, None]) * wi[None, :]


def test_row_scale_and_split_bounds():
    rng = np.random.RandomState(0)
    x = (rng.randn(64, 96) * np.exp2(rng.randint(-40, 41, (64, 1)))).astype(np.float32)
    x[3] = 0
    x[5, :4] = [65504.0, -1e-30, 0.0, 3.0]
    hi, lo, inv = split(x)
    e = np.log2(inv.astype(np.float64))
    assert np.array_equal(e, np.round(e)) and inv[3] == 1.0
    scaled = x.astype(np.float64) / inv[:, None]
    mx = np.abs(scaled).max(axis=1)
    live = mx > 0
    assert (mx[live] >= 2.0 ** 14).all() and (mx[live] < 2.0 ** 15).all()
    assert np.isfinite(hi).all() and np.abs(hi).max() <= 2.0 ** 15
    err = np.abs(hi.astype(np.float64) + lo - scaled)
    assert (err <= np.maximum(2.0 ** -22 * np.abs(scaled), 2.0 ** -25)).all()
    # clamped exponents: magnitudes beyond 2^+-100 of the fp16 window keep a finite, power-of-two scale
    m, i = row_scale(np.array([1e-38, 3e38, 0.0], dtype=np.float32))
    assert np.isfinite(m).all() and np.isfinite(i).all() and (m * i == 1.0).all() and m[2] == 1.0


def test_three_product_gemm_is_float32_class():
    rng = np.random.RandomState(1)
    for (M, N, K) in ((40, 48, 512), (17, 32, 2048)):
        a = np.maximum(rng.randn(M, K), 0).astype(np.float32) * np.exp2(rng.randint(-12, 13, (M, 1))).astype(np.float32)
        w = (rng.randn(N, K) * (2.0 / K) ** 0.5).astype(np.float32)
        ref = a.astype(np.float64) @ w.astype(np.float64).T
        c3 = gemm_x3(a, w)
        # float32 GEMM with the same 16-k chunking
        acc = np.zeros((M, N), dtype=np.float32)
        for k0 in range(0, K, 16):
            acc = (acc.astype(np.float64) + a[:, k0:k0 + 16].astype(np.float64) @ w[:, k0:k0 + 16].astype(np.float64).T).astype(np.float32)
        rs = np.abs(ref).max(axis=1, keepdims=True)
        e3 = (np.abs(c3 - ref) / rs).max()
        e32 = (np.abs(acc - ref) / rs).max()
        # operands are held to 2^-22 of their row maximum: the representation error alone is below 3e-7 of a row's largest output,
        # the total within 2x the float32 chain's own error + that
        assert e3 <= 2.0 * e32 + 3e-7, (M, N, K, e3, e32)
        assert e3 <= 2e-6
