"""
tools/x6t_bench.py -- timing of round 3's f32x6 Winograd layers and of the tile-record GEMM underneath (development aid).

  python tools/x6t_bench.py [--reps 20] [--only conv4]

Per layer shape: the x6 Winograd layer (three launches), its GEMM alone (frcnn_gemm_x6t on prepared records), and the one-launch
float32 Winograd layer it replaces, events on torch's current stream, random operands.  TF columns: executed Winograd-GEMM FLOP
(2 x 16 x tiles x cin x cout) / time; for the x6 GEMM additionally x 6 against the bf16 pipe's 2500 TFLOP/s.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from fasterrcnn_amd import _native as nv  # noqa: E402
from tools.layer_bench import timeit      # noqa: E402

LAYERS = [("conv3_2", 256, 256, 150, 250, False), ("conv4_1", 256, 512, 75, 125, False), ("conv4_2", 512, 512, 75, 125, False),
          ("conv4_3", 512, 512, 75, 125, True), ("conv5_x", 512, 512, 37, 62, False)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--gemm", type=str, action="append", default=[], help="time frcnn_gemm_x6t alone: M,N,K,batches (repeatable); random records")
    args = ap.parse_args()
    nv.require_gpu()
    lib = nv.lib()
    dev = "cuda:0"
    s = nv.stream_ptr()
    for spec in args.gemm:
        M, N, K, B = (int(v) for v in spec.split(","))
        Mp = (M + nv.X6T_ROW_TILE - 1) // nv.X6T_ROW_TILE * nv.X6T_ROW_TILE
        Np = (N + nv.X6T_COL_TILE - 1) // nv.X6T_COL_TILE * nv.X6T_COL_TILE
        a = torch.randn((B, M, K), device=dev)
        wt = torch.randn((B, N, K), device=dev) * 0.02
        a_per, b_per = int(lib.frcnn_x6t_record_bytes(Mp, K)), int(lib.frcnn_x6t_record_bytes(Np, K))
        ar = torch.zeros((B * a_per,), dtype=torch.uint8, device=dev)
        br = torch.zeros((B * b_per,), dtype=torch.uint8, device=dev)
        nv.check(lib.frcnn_split_rows_x6t(nv.ptr(a), K, M * K, nv.ptr(ar), M, Mp, K, B, s), "split a")
        nv.check(lib.frcnn_split_rows_x6t(nv.ptr(wt), K, N * K, nv.ptr(br), N, Np, K, B, s), "split b")
        c = torch.empty((B, M, N), device=dev)
        gwsb = int(lib.frcnn_gemm_x6t_workspace_bytes(M, N, K, B))
        gws = torch.empty((max(gwsb, 4),), dtype=torch.uint8, device=dev)
        us = timeit(lambda: nv.check(lib.frcnn_gemm_x6t(nv.ptr(ar), Mp, a_per, nv.ptr(br), Np, b_per, None, None, nv.ptr(c), N, M * N, M, N, K, B, 0,
                                                        nv.ptr(gws), gwsb, s), "gemm_x6t"), args.reps)
        fl = 2.0 * M * N * K * B
        print("gemm_x6t M=%d N=%d K=%d x%d: %8.1f us = %6.1f TF f32-equivalent, %.3f of the bf16 peak (splitws %d B)" % (
            M, N, K, B, us, fl / us / 1e6, 6 * fl / us / 1e6 / 2500.0, gwsb))
    if args.gemm and not args.only:
        return
    for name, cin, cout, h, w, pool in LAYERS:
        if args.only and args.only not in name:
            continue
        x = torch.randn((h, w, cin), device=dev)
        w_oihw = torch.randn((cout, cin, 3, 3), device=dev) * 0.02
        b = torch.zeros((cout,), device=dev)
        oh, ow = (h // 2, w // 2) if pool else (h, w)
        y = torch.empty((oh, ow, cout), device=dev)
        flags = nv.RELU | (nv.POOL2 if pool else 0)
        T = ((h + 1) // 2) * ((w + 1) // 2)
        gfl = 2.0 * 16 * T * cin * cout
        # float32 one-launch layer
        wf = torch.empty((16 * cout * cin,), device=dev)
        nv.check(lib.frcnn_pack_conv3x3_winograd_fused(nv.ptr(w_oihw), None, nv.ptr(wf), cout, cin, s), "pack_fused")
        us32 = timeit(lambda: nv.check(lib.frcnn_conv3x3_nhwc_winograd_fused(nv.ptr(x), nv.ptr(wf), nv.ptr(b), nv.ptr(y), h, w, cin, cout,
                                                                             flags, s), "fused"), args.reps)
        # x6 layer
        u = torch.empty((int(lib.frcnn_conv3x3_winograd_x6_pack_bytes(cout, cin)),), dtype=torch.uint8, device=dev)
        nv.check(lib.frcnn_pack_conv3x3_winograd_x6(nv.ptr(w_oihw), None, nv.ptr(u), cout, cin, s), "pack_x6")
        wsb = int(lib.frcnn_conv3x3_winograd_x6_workspace_bytes(1, h, w, cin, cout))
        ws = torch.zeros((wsb,), dtype=torch.uint8, device=dev)
        us6 = timeit(lambda: nv.check(lib.frcnn_conv3x3_nhwc_winograd_x6(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), 1, h, w, cin, cout, flags,
                                                                         nv.ptr(ws), wsb, s), "x6"), args.reps)
        # its GEMM alone: V records = the head of the workspace the layer just filled
        Tp = (T + nv.X6T_ROW_TILE - 1) // nv.X6T_ROW_TILE * nv.X6T_ROW_TILE
        Np = (cout + nv.X6T_COL_TILE - 1) // nv.X6T_COL_TILE * nv.X6T_COL_TILE
        a_per, b_per = int(lib.frcnn_x6t_record_bytes(Tp, cin)), int(lib.frcnn_x6t_record_bytes(Np, cin))
        m = torch.empty((16, T, cout), device=dev)
        gwsb = int(lib.frcnn_gemm_x6t_workspace_bytes(T, cout, cin, 16))
        gws = torch.empty((max(gwsb, 4),), dtype=torch.uint8, device=dev)
        usg = timeit(lambda: nv.check(lib.frcnn_gemm_x6t(nv.ptr(ws), Tp, a_per, nv.ptr(u), Np, b_per, None, None, nv.ptr(m), cout, T * cout, T,
                                                         cout, cin, 16, 0, nv.ptr(gws), gwsb, s), "gemm_x6t"), args.reps)
        print("%-8s %4d->%4d %4dx%-4d pool=%d  f32 one-launch %7.1f us (%.3f of 157.3) | x6 layer %7.1f us | x6 GEMM alone %7.1f us "
              "= %6.1f TF f32-equivalent, %.3f of the bf16 peak (splitws %d B)" % (
                  name, cin, cout, h, w, pool, us32, gfl / us32 / 1e6 / 157.3, us6, usg, gfl / usg / 1e6, 6 * gfl / usg / 1e6 / 2500.0, gwsb))


if __name__ == "__main__":
    main()
