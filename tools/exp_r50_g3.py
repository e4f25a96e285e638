"""tools/exp_r50_g3.py -- ResNet predict() throughput for the bottleneck_g3 settings (off / backbone / all) in ONE process.

  python tools/exp_r50_g3.py [--arch ResNet50] [--steps 100] [--modes off,backbone,all]

8 batch-1 images in flight (bench.py's resnet50 leg), one image at a time, and batches of 8.  Development aid; the numbers that count are
bench.py's."""
import argparse
import sys
import time

sys.path.insert(0, ".")
import torch

from fasterrcnn_amd import synthetic
from fasterrcnn_amd.models import resnet
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="ResNet50")
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--modes", default="off,backbone,all,off")
    ap.add_argument("--inflight", default="8", help="comma-separated numbers of batch-1 images in flight to time")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    m = FasterRCNNModel(num_classes=21, backbone=resnet.ResNetBackbone(getattr(resnet.Architecture, args.arch)))
    m.load_state_dict(synthetic.resnet_state_dict(1234, args.arch), strict=True)
    m = m.to(dev).eval()
    pool = [synthetic.image_rgb(100 + i).unsqueeze(0).to(dev) for i in range(8)]
    batch = torch.cat(pool, dim=0)

    def inflight(n_steps, n=8):
        pend = []
        for i in range(n_steps):
            if len(pend) == n:
                pend.pop(0).result()
            pend.append(m.predict_async(pool[i % 8], 0.05, slot=1 + (i % n)))
        while pend:
            pend.pop(0).result()

    def single(n_steps):
        for i in range(n_steps):
            m.predict(pool[i % 8], score_threshold=0.05)

    def batches(n_steps):
        pend, lane = [], 0
        for _ in range((n_steps + 7) // 8):
            if len(pend) == 2:
                for h in pend.pop(0):
                    h.result()
            pend.append(m.predict_batch_async(batch, 0.05, lane=lane))
            lane ^= 1
        while pend:
            for h in pend.pop(0):
                h.result()

    def rate(fn, steps):
        fn(16)
        best = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(steps)
            torch.cuda.synchronize()
            best.append(steps / (time.perf_counter() - t0))
        return sorted(best)[1]

    for mode in args.modes.split(","):
        m.bottleneck_g3 = mode
        fl = "  ".join("x%d: %7.1f" % (n, rate(lambda k, n=n: inflight(k, n), args.steps)) for n in (int(v) for v in args.inflight.split(",")))
        print("%-9s in flight %s img/s   one at a time: %7.1f   batches of 8: %7.1f" % (
            mode, fl, rate(single, args.steps // 2), rate(batches, args.steps)), flush=True)


if __name__ == "__main__":
    main()
