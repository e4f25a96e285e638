"""Timing experiment for gemm_x3t_kernel (needs a library built with -DHX_CLOCKS: SRC=gemm_x3t tools/build_ablate.sh hxclk -DHX_CLOCKS,
run with FRCNN_LIB_PATH=build/libfrcnn_hxclk.so).  Prints the shader clock the K loop ran at and the shader cycles one 16-k stage
took (60 MFMAs per SIMD = 1920 matrix-pipe cycles for the 320 x 256 tile, two waves per SIMD)."""
import sys
import numpy as np
import torch as t
sys.path.insert(0, ".")
from fasterrcnn_amd import _native as nv


def run(M, N, K, B, reps=300):
    lib = nv.lib()
    dev = t.device("cuda:0")
    s = nv.stream_ptr()
    Mp = (M + 319) // 320 * 320
    Np = (N + 255) // 256 * 256
    a = t.randn((B, M, K), device=dev)
    w = t.randn((B, N, K), device=dev) * 0.02
    a_per, b_per = int(lib.frcnn_x3t_record_bytes(Mp, K)), int(lib.frcnn_x3t_record_bytes(Np, K))
    ar = t.zeros((B * a_per,), dtype=t.uint8, device=dev)
    br = t.zeros((B * b_per,), dtype=t.uint8, device=dev)
    ai, bi = t.empty((B, Mp), device=dev), t.empty((B, Np), device=dev)
    nv.check(lib.frcnn_rows_scale_x3t(nv.ptr(a), K, M * K, nv.ptr(ai), M, Mp, K, B, s), "scale a")
    nv.check(lib.frcnn_rows_scale_x3t(nv.ptr(w), K, N * K, nv.ptr(bi), N, Np, K, B, s), "scale b")
    nv.check(lib.frcnn_split_rows_x3t(nv.ptr(a), K, M * K, nv.ptr(ai), nv.ptr(ar), M, Mp, K, B, s), "split a")
    nv.check(lib.frcnn_split_rows_x3t(nv.ptr(w), K, N * K, nv.ptr(bi), nv.ptr(br), N, Np, K, B, s), "split b")
    c = t.empty((B, M, N), device=dev)
    assert int(lib.frcnn_gemm_x3t_workspace_bytes(M, N, K, B)) == 0, "needs an unsplit shape"
    nblk = (Mp // 320) * (Np // 256) * B
    dbg = t.zeros((nblk * 8 * 8,), dtype=t.float32, device=dev)
    e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    for rep in range(reps):
        if rep == reps - 1:
            e0.record()
        nv.check(lib.frcnn_gemm_x3t(nv.ptr(ar), nv.ptr(ai), Mp, a_per, Mp, nv.ptr(br), nv.ptr(bi), Np, b_per, Np, None, None, nv.ptr(c), N, M * N, M, N, K,
                                    B, 0, nv.ptr(dbg), dbg.numel() * 4, s), "gemm_x3t")
    e1.record()
    t.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    o = dbg.view(nblk, 8, 8).cpu().numpy().astype(np.float64)
    assert (o[:, :, 7] == 1.0).all()
    cyc, real, pro, epi, nst = o[..., 0], o[..., 1], o[..., 2], o[..., 3], o[0, 0, 4]
    mhz = cyc / real * 100.0
    span = ((o[..., 6].max() - o[..., 5].min()) % (1 << 24)) / 100.0
    print("gemm_x3t M=%d N=%d K=%d x%d: %d blocks | launch %.1f us (events), first entry -> last exit %.1f us | sclk %.0f MHz | K loop %.0f "
          "cycles/stage (p10 %.0f, p90 %.0f; 1920 = both waves of a SIMD back to back) = %.3f of the pipe in shader cycles | per block: before "
          "the loop %.2f us, loop %.2f us, after %.2f us" % (
              M, N, K, B, nblk, us, span, mhz.mean(), (cyc / nst).mean(), np.percentile(cyc / nst, 10), np.percentile(cyc / nst, 90),
              1920.0 / (cyc / nst).mean(), pro.mean() / 100.0, real.mean() / 100.0, epi.mean() / 100.0))


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    for shp in [(2394, 512, 512, 16), (2394, 512, 2048, 16), (589, 512, 512, 16)]:
        run(*shp, reps=reps)
