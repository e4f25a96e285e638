"""tools/x3f_bench.py -- the one-launch f32x3 Winograd layer (csrc/wino_x3f.hip) against the float32 one-launch layer
(csrc/winofused.hip) on the six VGG-16 layers the latter still owns."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from fasterrcnn_amd import _native as nv  # noqa: E402
from tools.layer_bench import timeit      # noqa: E402

LAYERS = [("conv1_2", 64, 64, 600, 1000, True), ("conv2_1", 64, 128, 300, 500, False), ("conv2_2", 128, 128, 300, 500, True),
          ("conv3_1", 128, 256, 150, 250, False), ("conv3_2", 256, 256, 150, 250, False), ("conv3_3", 256, 256, 150, 250, True),
          ("conv4_1", 256, 512, 75, 125, False), ("conv4_2", 512, 512, 75, 125, False), ("conv4_3", 512, 512, 75, 125, True),
          ("conv5_x", 512, 512, 37, 62, False)]


def main():
    nv.require_gpu()
    lib = nv.lib()
    dev = "cuda:0"
    s = nv.stream_ptr()
    for name, cin, cout, h, w, pool in LAYERS:
        x = torch.randn((h, w, cin), device=dev).clamp(min=0)
        wt = torch.randn((cout, cin, 3, 3), device=dev) * 0.02
        b = torch.zeros((cout,), device=dev)
        oh, ow = (h // 2, w // 2) if pool else (h, w)
        y = torch.empty((oh, ow, cout), device=dev)
        flags = nv.RELU | (nv.POOL2 if pool else 0)
        wf = torch.empty((16 * cout * cin,), device=dev)
        nv.check(lib.frcnn_pack_conv3x3_winograd_fused(nv.ptr(wt), None, nv.ptr(wf), cout, cin, s), "pack_fused")
        us32 = timeit(lambda: nv.check(lib.frcnn_conv3x3_nhwc_winograd_fused(nv.ptr(x), nv.ptr(wf), nv.ptr(b), nv.ptr(y), h, w, cin, cout, flags, s),
                                       "fused"), 10, ramp_s=0.3)
        bank = torch.empty((16, cout, cin), device=dev)
        u = torch.empty((int(lib.frcnn_conv3x3_winograd_x3_pack_bytes(cout, cin)),), dtype=torch.int8, device=dev)
        nv.check(lib.frcnn_pack_conv3x3_winograd(nv.ptr(wt), None, nv.ptr(bank), cout, cin, s), "pack")
        nv.check(lib.frcnn_pack_conv3x3_winograd_x3(nv.ptr(bank), nv.ptr(u), cout, cin, s), "pack_x3")
        pair_ok = cin >= 64 and cout % 128 == 0 and int(lib.frcnn_conv3x3_winograd_x3_pair_workspace_bytes(1, h, w, cout)) > 0   # (make EXPERIMENTS=1 builds only)
        wsb = int(lib.frcnn_conv3x3_winograd_x3_pair_workspace_bytes(1, h, w, cout)) if pair_ok else int(lib.frcnn_conv3x3_winograd_x3_fused_workspace_bytes(1, h, w))
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        # the channel maxima computed once outside the timed calls (the forward chains them through the layers: no pass over the input)
        cm = torch.empty((h, w), device=dev)
        nv.check(lib.frcnn_pixel_absmax(nv.ptr(x), nv.ptr(cm), h * w, cin, s), "absmax")
        us = {}
        forms = [("four", nv.X3F_WAVES4)]
        if lib.frcnn_conv3x3_nhwc_winograd_x3_chain(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), 1, h, w, cin, cout, flags | nv.X3F_WAVES8, 1, nv.ptr(ws), wsb, nv.ptr(cm), None, s) == 0:
            forms.append(("eight", nv.X3F_WAVES8))        # (make EXPERIMENTS=1 builds only)
        if pair_ok:
            forms.append(("pair", nv.X3F_PAIR))           # csrc/wino_x3p.hip: two passes, 128 output channels per block
        for form, force in forms:
            us[form] = timeit(lambda: nv.check(lib.frcnn_conv3x3_nhwc_winograd_x3_chain(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), 1, h, w, cin, cout,
                                                                                     flags | force, 1, nv.ptr(ws), wsb, nv.ptr(cm), None, s), "x3_chain"), 10, ramp_s=0.3)
        gfl = 2.0 * 16 * ((h + 1) // 2) * ((w + 1) // 2) * cin * cout
        print("%-8s %4d->%4d %4dx%-4d pool=%d | float32 one-launch %7.1f us (%.2f of 157.3) | f32x3 one-launch, channel maxima given: " % (
              name, cin, cout, h, w, pool, us32, gfl / us32 / 1e6 / 157.3) + ", ".join(
              "%s %7.1f us (%.2f of the fp16 peak%s)" % (f, us[f], 3 * gfl / us[f] / 1e6 / 2500.0, "" if f == "four" else "; %.2fx" % (us["four"] / us[f])) for f, _ in forms))


if __name__ == "__main__":
    main()
