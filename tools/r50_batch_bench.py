"""
tools/r50_batch_bench.py -- ResNet-50 at 600x1000: images in flight one by one vs true batches through the feature extractor
(development aid; bench.py carries the judged legs).

  python tools/r50_batch_bench.py [--steps 64] [--arch ResNet50]
"""
import argparse
import ctypes as C
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from fasterrcnn_amd import _native as nv, synthetic  # noqa: E402
from fasterrcnn_amd.models import resnet  # noqa: E402
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel  # noqa: E402


def timed(fn, n, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return n / best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--arch", type=str, default="ResNet50")
    ap.add_argument("--modes", type=str, default="head,all")
    ap.add_argument("--backbone-only", type=int, default=0, help="only loop the feature extractor over a batch of this many images (for rocprofv3)")
    args = ap.parse_args()
    nv.require_gpu()
    dev = "cuda:0"
    m = FasterRCNNModel(num_classes=21, backbone=resnet.ResNetBackbone(getattr(resnet.Architecture, args.arch)))
    m.load_state_dict(synthetic.resnet_state_dict(1234, args.arch), strict=True)
    m = m.cuda().eval()
    pool = [synthetic.image_rgb(s).unsqueeze(0).to(dev) for s in range(8)]

    def inflight(n):
        pend = []
        for i in range(n):
            if len(pend) == 8:
                pend.pop(0).result()
            pend.append(m.predict_async(pool[i % 8], 0.05, slot=1 + (i % 8)))
        while pend:
            pend.pop(0).result()

    def batched(bsz, lanes):
        batch = torch.cat(pool[:bsz], dim=0)

        def run(n):
            pend, lane = [], 0
            for _ in range((n + bsz - 1) // bsz):
                if len(pend) == lanes:
                    for h in pend.pop(0):
                        h.result()
                pend.append(m.predict_batch_async(batch, 0.05, lane=lane))
                lane = (lane + 1) % lanes
            while pend:
                for h in pend.pop(0):
                    h.result()
        return run

    lib = nv.lib()
    if args.backbone_only:
        from fasterrcnn_amd import runtime as rt
        bsz = args.backbone_only
        m.x6_conv1x1 = args.modes.split(",")[0]
        w, p = m._weights(), m._forward_params(1)
        batch = torch.cat((pool * ((bsz + 7) // 8))[:bsz], dim=0)
        lane = rt.BackboneLane(dev, 608, 1008, bsz, 1024)
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(args.steps):
            nv.check(lib.frcnn_resnet_backbone(lane.handle, C.byref(w), C.byref(p), nv.ptr(batch), bsz, 600, 1000, nv.ptr(lane.features), s), "backbone")
        torch.cuda.synchronize()
        return
    for mode in args.modes.split(","):
        m.x6_conv1x1 = mode
        m.winograd_x6_layers = ("rpn_trunk",) if mode == "all" else ()
        inflight(16)
        print("x6_conv1x1=%-4s 8 batch-1 images in flight: %7.1f images/sec" % (mode, timed(inflight, args.steps)))
        for bsz, lanes in ((8, 1), (8, 2), (4, 2), (4, 4), (2, 4)):
            m._lanes.clear()
            m._slots.clear()
            run = batched(bsz, lanes)
            run(2 * bsz * lanes)
            print("x6_conv1x1=%-4s batches of %d, %d in flight:   %7.1f images/sec" % (mode, bsz, lanes, timed(run, args.steps)))
        # the feature extractor alone: per-image time of one pass over B images on one stream
        w = m._weights()
        p = m._forward_params(1)
        for bsz in (1, 2, 4, 8):
            m._lanes.clear()
            batch = torch.cat(pool[:bsz], dim=0)
            from fasterrcnn_amd import runtime as rt
            lane = rt.BackboneLane(dev, 608, 1008, bsz, 1024)
            s = torch.cuda.current_stream().cuda_stream

            def bb(n):
                for _ in range(n):
                    nv.check(lib.frcnn_resnet_backbone(lane.handle, C.byref(w), C.byref(p), nv.ptr(batch), bsz, 600, 1000, nv.ptr(lane.features), s), "backbone")
            bb(3)
            ips = timed(bb, 20) * bsz
            print("x6_conv1x1=%-4s feature extractor alone, batch %d: %6.3f ms per image" % (mode, bsz, 1e3 / ips))
            del lane


if __name__ == "__main__":
    main()
