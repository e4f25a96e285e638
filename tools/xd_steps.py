"""tools/xd_steps.py -- shader cycles per STEP (h, j) of a chunk of wino_x3d_kernel, summed over a block's chunks (build: SRC=wino_x3f
tools/build_ablate.sh xdsteps -DXD_CLOCKS -DXD_STEPS; FRCNN_LIB_PATH=build/libfrcnn_xdsteps.so).  Step 3 holds the chunk's barrier, steps 4-5
the halo DMA, every step two filter-piece loads."""
import sys
import numpy as np
import torch as t
sys.path.insert(0, ".")
from fasterrcnn_amd import _native as nv


def run(name, cin, cout, h, w, pool):
    dev = t.device("cuda:0")
    lib = nv.lib()
    s = nv.stream_ptr()
    x = t.randn((h, w, cin), device=dev).clamp(min=0)
    wt = t.randn((cout, cin, 3, 3), device=dev) * 0.02
    b = t.zeros((cout,), device=dev)
    bank = t.empty((16, cout, cin), device=dev)
    u = t.empty((int(lib.frcnn_conv3x3_winograd_x3_pack_bytes(cout, cin)),), dtype=t.int8, device=dev)
    nv.check(lib.frcnn_pack_conv3x3_winograd(nv.ptr(wt), None, nv.ptr(bank), cout, cin, s), "pack")
    nv.check(lib.frcnn_pack_conv3x3_winograd_x3(nv.ptr(bank), nv.ptr(u), cout, cin, s), "pack_x3")
    oh, ow = (h // 2, w // 2) if pool else (h, w)
    th, tw = (h + 1) // 2, (w + 1) // 2
    nblk = ((th + 3) // 4) * ((tw + 15) // 16) * (cout // 64)
    y = t.zeros((oh * ow * cout + 16 * nblk,), device=dev)
    wsb = int(lib.frcnn_conv3x3_winograd_x3_fused_workspace_bytes(1, h, w))
    ws = t.empty((wsb,), dtype=t.uint8, device=dev)
    flags = nv.RELU | (nv.POOL2 if pool else 0) | nv.X3F_WAVES4
    for rep in range(5):
        nv.check(lib.frcnn_conv3x3_nhwc_winograd_x3_fused(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), 1, h, w, cin, cout, flags, nv.ptr(ws), wsb, s), "x3f")
    t.cuda.synchronize()
    o = y[oh * ow * cout:].view(nblk, 16).cpu().numpy().astype(np.float64)
    k16 = o[0, 6]
    steps = o[:, 8:16].mean(axis=0) / (k16 - 0.125)          # (the first step of the first chunk has no predecessor)
    print("%-8s cycles per chunk %.0f (instrumented) | per step (h, j) = (0,0) .. (1,3): %s" % (name, o[:, 3].mean() / k16, " ".join("%.0f" % v for v in steps)))


if __name__ == "__main__":
    for a in [("conv2_2", 128, 128, 300, 500, True), ("conv3_2", 256, 256, 150, 250, False), ("conv4_2", 512, 512, 75, 125, False)]:
        run(*a)
