"""tools/xd_clocks.py -- where a block of the one-launch f32x3 Winograd kernel (csrc/wino_x3f.hip) spends its time.
Needs a library built with -DXD_CLOCKS:  SRC=wino_x3f tools/build_ablate.sh xdclk -DXD_CLOCKS ; FRCNN_LIB_PATH=build/libfrcnn_xdclk.so"""
import sys
import numpy as np
import torch as t
sys.path.insert(0, ".")
from fasterrcnn_amd import _native as nv


def run(name, cin, cout, h, w, pool, reps=20, force=0):
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
    pair = force == nv.X3F_PAIR
    nblk = ((th + 3) // 4) * ((tw + 15) // 16) * (cout // (128 if pair else 64))
    y = t.zeros((oh * ow * cout + 16 * (nblk + 8),), device=dev)
    wsb = int(lib.frcnn_conv3x3_winograd_x3_pair_workspace_bytes(1, h, w, cout)) if pair else int(lib.frcnn_conv3x3_winograd_x3_fused_workspace_bytes(1, h, w))
    ws = t.empty((wsb,), dtype=t.uint8, device=dev)
    flags = nv.RELU | (nv.POOL2 if pool else 0)
    e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    for rep in range(reps):
        if rep == reps - 1:
            e0.record()
        nv.check(lib.frcnn_conv3x3_nhwc_winograd_x3_fused(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), 1, h, w, cin, cout, flags | force, nv.ptr(ws), wsb, s), "x3f")
    e1.record()
    t.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    o = y[oh * ow * cout:].view(nblk + 8, 16).cpu().numpy().astype(np.float64)
    o = o[o[:, 7] == 1.0]
    assert len(o) == nblk, (len(o), nblk)
    pro, loop, epi, cyc, t_in, t_out, k16 = (o[:, i] for i in range(7))
    k16 = k16[0]
    span = ((t_out.max() - t_in.min()) % (1 << 24)) / 100.0
    start = np.sort(((t_in - t_in.min()) % (1 << 24)) / 100.0)
    print("%-8s %s %3d->%3d %4dx%-4d blocks %4d | launch (with the channel-maximum pass) %.1f us, first entry -> last exit %.1f us | per block: before the "
          "loop %.2f us, loop %.2f us = %.3f us / chunk (%.0f cycles / chunk at %.0f MHz; 1536 = the MFMAs alone), after %.2f us | block starts: "
          "p25 %.1f p50 %.1f p75 %.1f p100 %.1f us" % (name, {0: "auto ", nv.X3F_WAVES4: "four ", nv.X3F_WAVES8: "eight", nv.X3F_PAIR: "pair "}[force], cin, cout, h, w, nblk, us, span, pro.mean() / 100, loop.mean() / 100, loop.mean() / 100 / k16,
                                                       cyc.mean() / k16, (cyc / loop).mean() * 100, epi.mean() / 100,
                                                       np.percentile(start, 25), np.percentile(start, 50), np.percentile(start, 75), start.max()))
    if "chunks" not in sys.argv[1:]:
      print("         before the loop: loads issued after %.2f us, landed + barrier %.2f us later, first operand %.2f us | after: wait for the "
          "other waves %.2f us, column pass + LDS %.2f us, row pass + stores %.2f us" % tuple(o[:, i].mean() / 100 for i in range(8, 14))
          + (" | pair form: a 'chunk' is one pass over 16 input channels (48 MFMAs per wave, 64 tiles x 128 channels); the spill between the passes %.2f us (inside the loop time)" % (o[:, 14].mean() / 100) if pair else ""))
    if "chunks" in sys.argv[1:]:
        print("         cycles: chunk 0 %.0f, chunk 1 %.0f, steady-state chunks %.0f each (%d of them), chunk K16-2 %.0f, last chunk %.0f" % (
            o[:, 8].mean(), o[:, 9].mean(), o[:, 10].mean() / max(1, k16 - 4), k16 - 4, o[:, 11].mean(), o[:, 12].mean()))
    return cyc.mean() / k16


if __name__ == "__main__":
    for a in [("conv1_2", 64, 64, 600, 1000, True), ("conv2_2", 128, 128, 300, 500, True), ("conv3_1", 128, 256, 150, 250, False),
              ("conv3_2", 256, 256, 150, 250, False), ("conv3_3", 256, 256, 150, 250, True), ("conv4_2", 512, 512, 75, 125, False),
              ("conv5_x", 512, 512, 37, 62, False)]:
        c4 = run(*a, force=nv.X3F_WAVES4)
        if "pair" in sys.argv[1:] and a[1] >= 64 and a[2] >= 128:
            run(*a, force=nv.X3F_PAIR)
        if "four" in sys.argv[1:] or "pair" in sys.argv[1:]:          # (the four-wave kernel only: ablation builds of csrc/wino_x3f.hip)
            continue
        c8 = run(*a, force=nv.X3F_WAVES8)
        print("   => cycles per chunk: four waves %.0f, eight waves %.0f per wave pair (1536 = the MFMAs alone)" % (c4, c8))
