"""Timing experiment for wino_fused_kernel (needs a library built with -DWF_CLOCKS: tools/build_ablate.sh clk -DWF_CLOCKS,
run with FRCNN_LIB_PATH=build/libfrcnn_clk.so).  Prints, per layer shape, the shader clock the K loop ran at and the
shader cycles one stage (32 MFMAs per wave = 1024 matrix-pipe cycles, two waves per SIMD) took."""
import sys
import numpy as np
import torch as t
sys.path.insert(0, ".")
from fasterrcnn_amd import _native as nv
from fasterrcnn_amd.models import vgg16 as V

def run(h, w, cin, cout, reps=30):
    dev = t.device("cuda:0")
    x = t.randn(h, w, cin, device=dev)
    conv = t.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
    b = t.zeros(cout, device=dev)
    wp = V.pack_conv3x3(conv, "f32_winograd")
    tb = ((h + 1) // 2 + 1) // 2 * (((w + 1) // 2 + 15) // 16)       # tile blocks of 2 x 16 tiles
    nblk = tb * (cout // 64)                                          # x cout blocks of 64
    nalloc = 2 * nblk + 64                                            # the grid is padded per XCD group; blocks that ran set a marker
    y = t.zeros((h * w * cout + 32 * nalloc,), dtype=t.float32, device=dev)
    e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    for rep in range(reps):
        if rep == reps - 1:
            e0.record()
        nv.check(nv.lib().frcnn_conv3x3_nhwc_winograd_fused(nv.ptr(x), nv.ptr(wp), nv.ptr(b), nv.ptr(y), h, w, cin, cout, nv.RELU,
                                                            nv.stream_ptr()), "frcnn_conv3x3_nhwc_winograd_fused")
    e1.record()
    t.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    o = y[h * w * cout:].view(nalloc, 4, 8).cpu().numpy().astype(np.float64)
    o = o[o[:, 0, 7] == 1.0]
    assert o.shape[0] == nblk, (o.shape, nblk)
    cyc, real, pro, epi, t_in, t_out, nch = o[..., 0], o[..., 1], o[..., 2], o[..., 3], o[..., 4], o[..., 5], o[0, 0, 6]
    stages = 4 * nch
    mhz = cyc / real * 100.0
    span = ((t_out.max() - t_in.min()) % (1 << 24)) / 100.0
    busy = (((t_out - t_in) % (1 << 24)) / 100.0)[:, 0].sum()          # block residency, wave 0 of each block
    flop = nblk * stages * 4 * 32 * 2048.0                         # 4 waves x 32 MFMAs per stage
    print("%4dx%-4d %3d->%3d  blocks %5d | launch %.1f us (events), first entry -> last exit %.1f us | sclk %.0f MHz | K loop %.0f cycles/stage "
          "(p10 %.0f, p90 %.0f; 2048 = both waves of a SIMD back to back) | per block: before the loop %.2f us, loop %.2f us, after %.2f us | "
          "(sum %.2f us = %.0f cycles per stage pair) | slot occupancy %.2f of 512 | executed %.1f TF"
          % (h, w, cin, cout, nblk, us, span, mhz.mean(), (cyc / stages).mean(), np.percentile(cyc / stages, 10), np.percentile(cyc / stages, 90),
             pro.mean() / 100.0, real.mean() / 100.0, epi.mean() / 100.0, (pro + real + epi).mean() / 100.0,
             (pro + real + epi).mean() / 100.0 * mhz.mean() / stages, busy / (512.0 * span), flop / us / 1e6))
    return o


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30      # a few thousand: the clocks of a sustained (power-limited) run
    for shp in [(512, 512, 256, 256), (150, 250, 256, 256), (300, 500, 128, 128), (75, 125, 512, 512), (37, 62, 512, 512)]:
        run(*shp, reps=reps)
