"""
tools/conv1x1_bench.py -- the 1x1 (and stride-2 3x3) convolutions of the ResNet-50 bottlenecks at 600x1000 through frcnn_conv_nhwc
(csrc/conv_gather.hip), for a batch of N images (development aid).

  python tools/conv1x1_bench.py [--batch 8] [--reps 10]

Prints microseconds, TFLOP/s and the algorithmic HBM rate (input + output + residual + weights, each once).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from fasterrcnn_amd import _native as nv  # noqa: E402
from tools.layer_bench import timeit      # noqa: E402

# name, h, w, cin, cout, k, stride, residual
SHAPES = [("l1.conv1", 150, 250, 256, 64, 1, 1, False), ("l1.conv3", 150, 250, 64, 256, 1, 1, True),
          ("l2.conv1", 75, 125, 512, 128, 1, 1, False), ("l2.conv3", 75, 125, 128, 512, 1, 1, True),
          ("l3.conv1", 38, 63, 1024, 256, 1, 1, False), ("l3.conv3", 38, 63, 256, 1024, 1, 1, True),
          ("l2.down", 150, 250, 256, 512, 1, 2, False), ("l3.down", 75, 125, 512, 1024, 1, 2, False),
          ("l2.0.conv2", 150, 250, 128, 128, 3, 2, False), ("l3.0.conv2", 75, 125, 256, 256, 3, 2, False)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", type=str, default="")
    args = ap.parse_args()
    nv.require_gpu()
    lib = nv.lib()
    dev = "cuda:0"
    s = nv.stream_ptr()
    n = args.batch
    for name, h, w, cin, cout, k, stride, res in SHAPES:
        if args.only and args.only not in name:
            continue
        pad = k // 2
        ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        x = torch.randn((n, h, w, cin), device=dev)
        wp = torch.randn((k * k, cout, cin), device=dev) * 0.05
        b = torch.randn((cout,), device=dev)
        r = torch.randn((n, ho, wo, cout), device=dev) if res else None
        y = torch.empty((n, ho, wo, cout), device=dev)
        wsb = int(lib.frcnn_conv_workspace_bytes(n, h, w, cin, cout, k, stride, pad))
        ws = torch.empty((max(wsb, 4) // 4,), device=dev)
        us = timeit(lambda: nv.check(lib.frcnn_conv_nhwc(nv.ptr(x), nv.ptr(wp), nv.ptr(b), nv.ptr(r), nv.ptr(y), n, h, w, cin, cout, k, stride, pad,
                                                         nv.RELU, nv.ptr(ws), wsb, s), "conv"), args.reps, ramp_s=0.3)
        m = n * ho * wo
        fl = 2.0 * m * cout * cin * k * k
        by = 4.0 * (m * cin * (1 if stride == 1 or k == 3 else 1) * (stride * stride if k == 1 else 1) * 0 + n * h * w * cin + m * cout * (2 if res else 1) + k * k * cout * cin)
        print("%-11s N=%d %3dx%-3d %4d->%4d k%d s%d res=%d splitws %9d B: %7.1f us  %6.1f TFLOP/s  %5.2f TB/s algorithmic" % (
            name, n, h, w, cin, cout, k, stride, int(res), wsb, us, fl / us / 1e6, by / us / 1e6))


if __name__ == "__main__":
    main()
