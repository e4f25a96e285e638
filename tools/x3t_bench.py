"""
tools/x3t_bench.py -- frcnn_gemm_x3t (f32x3: two fp16 terms per row-scaled operand, three MFMAs per product) against frcnn_gemm_x6t
(f32x6) and float64: time and error (development aid).

  python tools/x3t_bench.py [--reps 20]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from fasterrcnn_amd import _native as nv  # noqa: E402
from tools.layer_bench import timeit      # noqa: E402

SHAPES = [("conv4 GEMMs", 2394, 512, 512, 16), ("conv5 GEMMs", 589, 512, 512, 16), ("conv4_1 GEMMs", 2394, 512, 256, 16),
          ("fc1", 300, 4096, 25088, 1), ("fc2", 300, 4096, 4096, 1), ("ragged", 137, 260, 96, 3)]


def pad(v, m):
    return (v + m - 1) // m * m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--range", type=float, default=0.0, help="log2 spread of per-row magnitudes of A (0 = none)")
    args = ap.parse_args()
    nv.require_gpu()
    lib = nv.lib()
    dev = "cuda:0"
    s = nv.stream_ptr()
    for name, M, N, K, B in SHAPES:
        if args.only and args.only not in name:
            continue
        g = torch.Generator(device=dev).manual_seed(M + N + K)
        a = torch.randn((B, M, K), device=dev, generator=g).clamp(min=0)
        if args.range:
            a = a * torch.exp2((torch.rand((B, M, 1), device=dev, generator=g) - 0.5) * args.range)
        w = torch.randn((B, N, K), device=dev, generator=g) * (2.0 / K) ** 0.5
        bias = torch.randn((N,), device=dev, generator=g) * 0.1
        Mp, Np = pad(M, nv.X6T_ROW_TILE), pad(N, nv.X6T_COL_TILE)
        ref = torch.matmul(a.double(), w.double().transpose(1, 2)) + bias.double()
        scale = float(ref.abs().max())
        # x6t
        a6, b6 = int(lib.frcnn_x6t_record_bytes(Mp, K)), int(lib.frcnn_x6t_record_bytes(Np, K))
        ar6 = torch.zeros((B * a6,), dtype=torch.uint8, device=dev)
        br6 = torch.zeros((B * b6,), dtype=torch.uint8, device=dev)
        nv.check(lib.frcnn_split_rows_x6t(nv.ptr(a), K, M * K, nv.ptr(ar6), M, Mp, K, B, s), "split a6")
        nv.check(lib.frcnn_split_rows_x6t(nv.ptr(w), K, N * K, nv.ptr(br6), N, Np, K, B, s), "split b6")
        c6 = torch.empty((B, M, N), device=dev)
        w6b = int(lib.frcnn_gemm_x6t_workspace_bytes(M, N, K, B))
        ws6 = torch.empty((max(w6b, 4),), dtype=torch.uint8, device=dev)
        f6 = lambda: nv.check(lib.frcnn_gemm_x6t(nv.ptr(ar6), Mp, a6, nv.ptr(br6), Np, b6, nv.ptr(bias), None, nv.ptr(c6), N, M * N, M, N, K, B, 0,
                                                 nv.ptr(ws6), w6b, s), "gemm_x6t")
        us6 = timeit(f6, args.reps, ramp_s=0.5)
        # x3t
        a3, b3 = int(lib.frcnn_x3t_record_bytes(Mp, K)), int(lib.frcnn_x3t_record_bytes(Np, K))
        ar3 = torch.zeros((B * a3,), dtype=torch.uint8, device=dev)
        br3 = torch.zeros((B * b3,), dtype=torch.uint8, device=dev)
        ai = torch.empty((B, Mp), device=dev)
        bi = torch.empty((B, Np), device=dev)
        nv.check(lib.frcnn_rows_scale_x3t(nv.ptr(a), K, M * K, nv.ptr(ai), M, Mp, K, B, s), "scale a")
        nv.check(lib.frcnn_rows_scale_x3t(nv.ptr(w), K, N * K, nv.ptr(bi), N, Np, K, B, s), "scale b")
        fa = lambda: nv.check(lib.frcnn_split_rows_x3t(nv.ptr(a), K, M * K, nv.ptr(ai), nv.ptr(ar3), M, Mp, K, B, s), "split a3")
        fa()
        nv.check(lib.frcnn_split_rows_x3t(nv.ptr(w), K, N * K, nv.ptr(bi), nv.ptr(br3), N, Np, K, B, s), "split b3")
        c3 = torch.empty((B, M, N), device=dev)
        w3b = int(lib.frcnn_gemm_x3t_workspace_bytes(M, N, K, B))
        ws3 = torch.empty((max(w3b, 4),), dtype=torch.uint8, device=dev)
        f3 = lambda: nv.check(lib.frcnn_gemm_x3t(nv.ptr(ar3), nv.ptr(ai), Mp, a3, Mp, nv.ptr(br3), nv.ptr(bi), Np, b3, Np, nv.ptr(bias), None, nv.ptr(c3),
                                                 N, M * N, M, N, K, B, 0, nv.ptr(ws3), w3b, s), "gemm_x3t")
        us3 = timeit(f3, args.reps, ramp_s=0.5)
        # exact-f32 MFMA kernel (frcnn_linear), batch 0 only
        npad = pad(N, 128)
        wp = torch.zeros((npad, K), device=dev)
        wp[:N] = w[0]
        y32 = torch.empty((M, N), device=dev)
        lwb = int(lib.frcnn_linear_workspace_bytes(M, N, K))
        lws = torch.empty((max(lwb, 4),), dtype=torch.uint8, device=dev)
        nv.check(lib.frcnn_linear(nv.ptr(a[0].contiguous()), K, nv.ptr(wp), nv.ptr(bias), nv.ptr(y32), N, M, N, K, 0, nv.ptr(lws), lwb, s), "linear")
        torch.cuda.synchronize()
        e = lambda y, r: (float((y.double() - r).abs().max()) / scale, float(((y.double() - r) ** 2).mean().sqrt()) / scale)
        e6, e3, e32 = e(c6, ref), e(c3, ref), e(y32, ref[0])
        fl = 2.0 * M * N * K * B
        print("%-14s M=%d N=%d K=%d x%d | x6t %7.1f us (%.3f of bf16 peak) | x3t %7.1f us (%.3f of fp16 peak, %.2fx) | max err / scale: x6t %.2e  x3t %.2e  "
              "f32 %.2e | rms: x6t %.2e  x3t %.2e  f32 %.2e" % (name, M, N, K, B, us6, 6 * fl / us6 / 1e6 / 2500.0, us3, 3 * fl / us3 / 1e6 / 2500.0,
                                                                us6 / us3, e6[0], e3[0], e32[0], e6[1], e3[1], e32[1]))


if __name__ == "__main__":
    main()
