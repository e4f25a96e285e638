"""
tools/layer_bench.py -- per-layer timing of the HIP kernels through the C ABI (development aid).

  python tools/layer_bench.py [--reps 20]

Times every VGG-16 conv layer shape (600x1000 input), the FC layers and the proposal kernels in
isolation with events on torch's current stream (the kernels are launched on that stream) and
prints microseconds and TFLOP/s per layer.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from fasterrcnn_amd import _native as nv  # noqa: E402

LAYERS = [("conv1_2", 64, 64, 600, 1000, True), ("conv2_1", 64, 128, 300, 500, False),
          ("conv2_2", 128, 128, 300, 500, True), ("conv3_1", 128, 256, 150, 250, False),
          ("conv3_2", 256, 256, 150, 250, False), ("conv3_3", 256, 256, 150, 250, True),
          ("conv4_1", 256, 512, 75, 125, False), ("conv4_2", 512, 512, 75, 125, False),
          ("conv4_3", 512, 512, 75, 125, True), ("conv5_x", 512, 512, 37, 62, False)]


def timeit(fn, reps, ramp_s=1.0):
    """Median of 5 batches of `reps` launches after `ramp_s` seconds of the same load (the GPU needs ~1.5 s to leave its idle
    power state; a cold 20-launch measurement reads 10-20 % slow and noisy)."""
    import time
    fn()
    torch.cuda.synchronize()
    t_end = time.perf_counter() + ramp_s
    while time.perf_counter() < t_end:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    out = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / reps)   # us
    return sorted(out)[2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--winograd", action="store_true", help="time the Winograd F(2x2,3x3) path for layers with cout %% 128 == 0 "
                    "(TF column = direct-convolution FLOP / time)")
    ap.add_argument("--fused", action="store_true", help="time the ONE-launch Winograd layer (csrc/winofused.hip); TF column = "
                    "direct-convolution FLOP / time, second figure = executed Winograd GEMM FLOP / time")
    ap.add_argument("--shape", type=str, action="append", default=[], help="extra layer: cin,cout,h,w,pool (repeatable)")
    args = ap.parse_args()
    for i, sh in enumerate(args.shape):
        cin, cout, h, w, pool = (int(v) for v in sh.split(","))
        LAYERS.append(("custom%d" % i, cin, cout, h, w, bool(pool)))
    if args.shape and not args.only:
        args.only = "custom"
    nv.require_gpu()
    lib = nv.lib()
    dev = "cuda:0"
    s = nv.stream_ptr()
    total_us, total_fl = 0.0, 0.0
    mult = {"conv5_x": 4}
    for name, cin, cout, h, w, pool in LAYERS:
        if args.only and args.only not in name:
            continue
        x = torch.randn((h, w, cin), device=dev)
        wp = torch.randn((9, cout, cin), device=dev) * 0.02
        b = torch.zeros((cout,), device=dev)
        oh, ow = (h // 2, w // 2) if pool else (h, w)
        y = torch.empty((oh, ow, cout), device=dev)
        wino = args.winograd and cout % 128 == 0
        fused = args.fused and cout % 64 == 0 and cin % 16 == 0
        if fused:
            w_oihw = torch.randn((cout, cin, 3, 3), device=dev) * 0.02
            wf = torch.empty((16 * cout * cin,), device=dev)
            nv.check(lib.frcnn_pack_conv3x3_winograd_fused(nv.ptr(w_oihw), None, nv.ptr(wf), cout, cin, s), "pack_winograd_fused")
            wsb = 0
        elif wino:
            w_oihw = torch.randn((cout, cin, 3, 3), device=dev) * 0.02
            wu = torch.empty((16, cout, cin), device=dev)
            nv.check(lib.frcnn_pack_conv3x3_winograd(nv.ptr(w_oihw), None, nv.ptr(wu), cout, cin, s), "pack_winograd")
            wsb = int(lib.frcnn_conv3x3_winograd_workspace_bytes(1, h, w, cin, cout))
        else:
            wsb = int(lib.frcnn_conv3x3_workspace_bytes(h, w, cin, cout))
        ws = torch.empty((max(wsb, 4) // 4,), device=dev)
        flags = nv.RELU | (nv.POOL2 if pool else 0)

        def run():
            if fused:
                nv.check(lib.frcnn_conv3x3_nhwc_winograd_fused(nv.ptr(x), nv.ptr(wf), nv.ptr(b), nv.ptr(y), h, w, cin, cout, flags, s),
                         "conv_winograd_fused")
            elif wino:
                nv.check(lib.frcnn_conv3x3_nhwc_winograd(nv.ptr(x), nv.ptr(wu), nv.ptr(b), nv.ptr(y), 1, h, w, cin, cout, flags,
                                                         nv.ptr(ws), wsb, s), "conv_winograd")
            else:
                nv.check(lib.frcnn_conv3x3_nhwc(nv.ptr(x), nv.ptr(wp), nv.ptr(b), nv.ptr(y), h, w, cin, cout, flags,
                                                nv.ptr(ws), wsb, s), "conv")
        us = timeit(run, args.reps)
        fl = 2.0 * 9 * cin * cout * h * w
        k = mult.get(name, 1)
        total_us += us * k
        total_fl += fl * k
        extra = ""
        if fused:
            gfl = 2.0 * 16 * ((h + 1) // 2) * ((w + 1) // 2) * cin * cout
            extra = "  (executed %.1f TF = %.3f of 157.3)" % (gfl / us / 1e6, gfl / us / 1e6 / 157.3)
        print("%-8s %4d->%4d %4dx%-4d pool=%d splitws=%9d  %8.1f us  %6.1f TF%s" % (name, cin, cout, h, w, pool, wsb, us, fl / us / 1e6, extra))
    if not args.only:
        print("all MFMA convs of one image (conv5_x x4 incl. RPN trunk): %.1f us, %.1f TF" % (total_us, total_fl / total_us / 1e6))
    if not args.only or "c3" in args.only:
        x = torch.randn((3, 600, 1000), device=dev)
        wp = torch.randn((27, 64), device=dev) * 0.1
        b = torch.zeros((64,), device=dev)
        y = torch.empty((600, 1000, 64), device=dev)

        def run_c3():
            nv.check(lib.frcnn_conv3x3_c3(nv.ptr(x), nv.ptr(wp), nv.ptr(b), nv.ptr(y), 600, 1000, 64, nv.RELU, s), "c3")
        us = timeit(run_c3, args.reps)
        print("conv1_1 (c3) 600x1000: %8.1f us  %.2f TB/s of output writes" % (us, 153.6e6 / us / 1e6))
    # FC layers
    for name, m, n, k in (("fc1", 300, 4096, 25088), ("fc2", 300, 4096, 4096), ("heads", 300, 101, 4096), ("rpn1x1", 2294, 45, 512)):
        if args.only and args.only not in name:
            continue
        a = torch.randn((m, k), device=dev)
        wt = torch.randn(((n + 127) // 128 * 128, k), device=dev) * 0.01
        b = torch.zeros((n,), device=dev)
        y = torch.empty((m, n), device=dev)
        wsb = int(lib.frcnn_linear_workspace_bytes(m, n, k))
        ws = torch.empty((max(wsb, 4) // 4,), device=dev)

        def run():
            nv.check(lib.frcnn_linear(nv.ptr(a), k, nv.ptr(wt), nv.ptr(b), nv.ptr(y), n, m, n, k, nv.RELU, nv.ptr(ws), wsb, s), "linear")
        us = timeit(run, args.reps)
        print("%-8s M=%d N=%d K=%d  %8.1f us  %6.1f TF" % (name, m, n, k, us, 2.0 * m * n * k / us / 1e6))


if __name__ == "__main__":
    main()
