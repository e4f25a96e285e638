"""
GPU parity tests of the "f32x6" dense layers (csrc/linear_x6.hip): fc1 / fc2 of models/vgg16.py:129-133 with exactly split
bf16x3 operands, six bf16 MFMAs per product and f32 accumulation.

Tolerances: the split is exact (hi + mid + lo == x bit for bit); a layer against float64 truth must be no worse than 1.5x the
exact-f32 MFMA kernel's own error + 2e-7 of max|y| (dropped terms <= 2^-24 relative per product), and within 4e-6 * sqrt(K) of
max|y| like every fp32 GEMM-class kernel here; end to end the model reproduces the reference's golden vectors in either fc mode.
"""
import os

import numpy as np
import pytest
import torch

from fasterrcnn_amd import _native as nv
from fasterrcnn_amd import synthetic
from fasterrcnn_amd.models import vgg16 as V

pytestmark = pytest.mark.gpu


def records_to_planes(rec, rows, k):
    """uint8 record array (chunk-major: [k/16][rows][hi, mid, lo][16] bf16) -> three float32 (rows, k) matrices (hi, mid, lo)."""
    r = rec.cpu().numpy().view(np.uint16).reshape(k // 16, rows, 3, 16).transpose(1, 0, 2, 3)
    f = (r.astype(np.uint32) << 16).view(np.float32)
    return [f[:, :, p, :].reshape(rows, k) for p in range(3)]


def test_split_is_exact_and_padded_rows_are_zero():
    gen = torch.Generator().manual_seed(1)
    a = torch.randn((37, 160), generator=gen) * torch.exp(torch.randn((37, 160), generator=gen) * 3)     # wide dynamic range
    a[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e-30, 65504.0, 1e-3, 0.1])
    rec = V.split_rows_x6(a.cuda(), rows_out=128)
    assert rec.numel() == 128 * 10 * 96
    hi, mid, lo = records_to_planes(rec, 128, 160)
    x = a.numpy()
    assert np.array_equal((hi[:37].astype(np.float64) + mid[:37] + lo[:37]).astype(np.float32), x)        # hi + mid + lo == x
    assert np.abs(mid[:37]).max() <= np.abs(hi[:37]).max() * 2.0 ** -7 and (hi[37:] == 0).all() and (lo[37:] == 0).all()
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.abs(lo[:37]) / np.abs(x)
    assert np.nanmax(rel[np.abs(x) > 1e-20]) <= 2.0 ** -15


@pytest.mark.parametrize("M,N,K,relu", [(300, 4096, 25088, True), (300, 4096, 4096, True), (137, 256, 512, False), (1, 128, 32, True),
                                        (320, 132, 80, False), (7, 4, 48, True)])
def test_linear_x6_against_float64_and_the_exact_f32_kernel(M, N, K, relu):
    gen = torch.Generator().manual_seed(M + N + K)
    a = torch.randn((M, K), generator=gen).clamp(min=0)          # post-ReLU features
    w = torch.randn((N, K), generator=gen) * (2.0 / K) ** 0.5
    b = torch.randn((N,), generator=gen) * 0.1
    ref = a.double() @ w.double().t() + b.double()
    if relu:
        ref = ref.clamp(min=0)
    ad, wd, bd = a.cuda(), w.cuda(), b.cuda()
    npad = (N + 127) // 128 * 128
    a_rec, w_rec = V.split_rows_x6(ad), V.split_rows_x6(wd, rows_out=npad)
    y = V.linear_x6(a_rec, w_rec, bd, M, N, K, relu, want="float32")
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    e6 = float((y.cpu().double() - ref).abs().max()) / scale
    wpad = torch.zeros((npad, K), device="cuda")
    wpad[:N] = wd
    e32 = float((V.linear(ad, wpad, bd, N, relu).cpu().double() - ref).abs().max()) / scale
    print("linear_x6 M=%d N=%d K=%d: max err / max|y| = %.3g (exact-f32 MFMA kernel %.3g)" % (M, N, K, e6, e32))
    assert e6 <= 1.5 * e32 + 2e-7 and e6 <= 4e-6 * np.sqrt(K)
    # run-to-run identical (deterministic split-K)
    y2 = V.linear_x6(a_rec, w_rec, bd, M, N, K, relu, want="float32")
    assert torch.equal(y, y2)
    if N % 16 == 0:
        # the records the reduction emits for the next layer == the split of its float32 output
        y_rec = V.linear_x6(a_rec, w_rec, bd, M, N, K, relu, want="records")
        assert torch.equal(y_rec, V.split_rows_x6(y))


def test_linear_x6_rejects_bad_arguments():
    lib = nv.lib()
    s = nv.stream_ptr()
    x = torch.zeros((1 << 20,), device="cuda")
    p = nv.ptr(x)
    assert lib.frcnn_linear_x6(None, p, p, p, 128, None, 8, 128, 64, 0, p, 1 << 22, s) == -1
    assert lib.frcnn_linear_x6(p, p, p, p, 128, None, 321, 128, 64, 0, p, 1 << 22, s) == -4        # M > 320
    assert lib.frcnn_linear_x6(p, p, p, p, 128, None, 8, 128, 40, 0, p, 1 << 22, s) == -4         # K % 16
    assert lib.frcnn_linear_x6(p, p, p, None, 128, None, 8, 128, 64, 0, p, 1 << 22, s) == -1      # no output
    assert lib.frcnn_linear_x6(p, p, p, p, 128, None, 8, 128, 64, 0, p, 16, s) == -1              # scratch too small
    assert lib.frcnn_linear_x6_workspace_bytes(300, 4096, 25088) == 8 * 300 * 4096 * 4
    assert lib.frcnn_split_rows_x6(p, 40, p, 4, 4, 40, s) == -1                                    # K % 16


def test_model_fc_modes_agree_and_both_reproduce_the_golden_vectors(gpu_model, golden_dir):
    """fc_math_mode "f32x6" (default) vs "f32": same proposals bit for bit (the RPN does not depend on fc1 / fc2), class
    probabilities within 1e-5, the same detections; and both reproduce the reference's golden detections."""
    assert gpu_model.fc_math_mode == "f32x6"
    g = np.load(os.path.join(golden_dir, "vgg16_600x1000_s0.npz"))
    img = synthetic.image(0).unsqueeze(0).cuda()
    out = {}
    try:
        for mode in ("f32x6", "f32"):
            gpu_model.fc_math_mode = mode
            out[mode] = (gpu_model(image_data=img), gpu_model.predict(image_data=img, score_threshold=0.05))
    finally:
        gpu_model.fc_math_mode = "f32x6"
    (p6, c6, d6), det6 = out["f32x6"]
    (p32, c32, d32), det32 = out["f32"]
    assert torch.equal(p6, p32)
    assert float((c6 - c32).abs().max()) <= 1e-5 and float((d6 - d32).abs().max()) <= 2e-5 * max(1.0, float(d32.abs().max()))
    ref = g["detections"]
    for mode, det in (("f32x6", det6), ("f32", det32)):
        n_ok = 0
        for c in range(1, 21):
            r = ref[ref[:, 0] == c][:, 1:]
            if len(r) and len(det[c]):
                d = np.abs(det[c][:, None, :4] - r[None, :, :4]).max(axis=2)
                j = d.argmin(axis=0)
                n_ok += int(((d[j, np.arange(len(r))] <= 1e-3) & (np.abs(det[c][j, 4] - r[:, 4]) <= 1e-4)).sum())
        print("fc_math_mode %s: %d/%d reference detections reproduced" % (mode, n_ok, len(ref)))
        assert n_ok == len(ref)
    with pytest.raises(ValueError):
        gpu_model.fc_math_mode = "bf16"
