"""tools/exp_fc1_ablate.py -- what bounds fc1's GEMM (gemm_x3t_kernel, M = 300, N = 4096, K = 25088)?  Times frcnn_gemm_x3t (+ its split
reduction) with the library FRCNN_LIB_PATH points at; run once per ablation build (SRC=gemm_x3t tools/build_ablate.sh hxa1 -DHX_ABLATE=1 ...).
HX_ABLATE: 1 no LDS-DMA after the prologue, 2 no MFMAs, 4 no epilogue stores, 16 no fragment reads after the first stage (results wrong)."""
import os
import sys

sys.path.insert(0, ".")
import torch

from fasterrcnn_amd import _native as nv
from tools.layer_bench import timeit


def main():
    nv.require_gpu()
    lib = nv.lib()
    dev = "cuda:0"
    s = nv.stream_ptr()
    for (M, N, K) in ((300, 4096, 25088), (300, 4096, 4096)):
        B = 1
        pad = lambda v, m: (v + m - 1) // m * m
        Mp, Np = pad(M, nv.X6T_ROW_TILE), pad(N, nv.X6T_COL_TILE)
        a = torch.randn((B, M, K), device=dev).clamp(min=0)
        w = torch.randn((B, N, K), device=dev) * (2.0 / K) ** 0.5
        bias = torch.zeros((N,), device=dev)
        a3, b3 = int(lib.frcnn_x3t_record_bytes(Mp, K)), int(lib.frcnn_x3t_record_bytes(Np, K))
        ar3 = torch.zeros((B * a3,), dtype=torch.uint8, device=dev)
        br3 = torch.zeros((B * b3,), dtype=torch.uint8, device=dev)
        ai, bi = torch.empty((B, Mp), device=dev), torch.empty((B, Np), device=dev)
        nv.check(lib.frcnn_rows_scale_x3t(nv.ptr(a), K, M * K, nv.ptr(ai), M, Mp, K, B, s), "scale a")
        nv.check(lib.frcnn_rows_scale_x3t(nv.ptr(w), K, N * K, nv.ptr(bi), N, Np, K, B, s), "scale b")
        nv.check(lib.frcnn_split_rows_x3t(nv.ptr(a), K, M * K, nv.ptr(ai), nv.ptr(ar3), M, Mp, K, B, s), "split a")
        nv.check(lib.frcnn_split_rows_x3t(nv.ptr(w), K, N * K, nv.ptr(bi), nv.ptr(br3), N, Np, K, B, s), "split b")
        del a, w
        c3 = torch.empty((B, M, N), device=dev)
        wb = int(lib.frcnn_gemm_x3t_workspace_bytes(M, N, K, B))
        ws = torch.empty((max(wb, 4),), dtype=torch.uint8, device=dev)
        f = lambda: nv.check(lib.frcnn_gemm_x3t(nv.ptr(ar3), nv.ptr(ai), Mp, a3, Mp, nv.ptr(br3), nv.ptr(bi), Np, b3, Np, nv.ptr(bias), None, nv.ptr(c3),
                                                N, M * N, M, N, K, B, 0, nv.ptr(ws), wb, s), "gemm_x3t")
        us = timeit(f, 20, ramp_s=0.3)
        print("%s  M %d N %d K %d: %.1f us per call (GEMM + split reduction; workspace %.1f MB)" % (os.path.basename(nv.LIB_PATH), M, N, K, us, wb / 1e6))


if __name__ == "__main__":
    main()
