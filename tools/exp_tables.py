"""tools/exp_tables.py -- the headline loop (bench.py: predict_async, N images in flight) timed for several arithmetic tables in ONE process.

  python tools/exp_tables.py [--inflight 3] [--steps 200] [--tables default,x3f_all,...]

Development aid (which layers belong on the one-launch f32x3 kernel at a given number of images in flight); the numbers that count are
bench.py's.  A table is a dict of FasterRCNNModel attributes; "default" leaves the model as constructed."""
import argparse
import sys
import time

sys.path.insert(0, ".")
import torch

from fasterrcnn_amd import synthetic
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models.vgg16 import VGG16Backbone

X6 = ("conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3", "rpn_trunk")
X3F = ("conv2_2", "conv3_1", "conv3_2", "conv3_3")
X3F_ALL = ("conv1_2", "conv2_1") + X3F
C4 = ("conv4_1", "conv4_2", "conv4_3")
C5 = ("conv5_1", "conv5_2", "conv5_3")


def without(names, drop):
    return tuple(n for n in names if n not in drop)


TABLES = {
    "default": {},
    "no_x3f": {"winograd_x3f_layers": ()},
    "no_inflight_x3f": {"inflight_winograd_x3f_layers": ()},
    "inflight_conv4": {"inflight_winograd_x3f_layers": C4},
    "inflight_conv5": {"inflight_winograd_x3f_layers": C5 + ("rpn_trunk",)},
    "no_conv1_2": {"winograd_x3f_layers": ("conv2_1",) + X3F},
    "no_conv2_1": {"winograd_x3f_layers": X3F},
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inflight", type=int, default=3)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--tables", default="default,no_inflight_x3f,no_conv1_2,no_x3f,default")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    model = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
    model.load_state_dict(synthetic.vgg16_state_dict(0), strict=True)
    model = model.to(dev).eval()
    pool = [synthetic.image(100 + i, 600, 1000).unsqueeze(0).to(dev) for i in range(8)]
    base = {k: getattr(model, k) for k in ("winograd_x6_layers", "winograd_x3_layers", "winograd_x3f_layers", "inflight_winograd_x3f_layers", "fc_math_mode")}
    n = max(1, args.inflight)

    def run(steps):
        pending = []
        for i in range(steps):
            if len(pending) == n:
                pending.pop(0).result()
            pending.append(model.predict_async(pool[i % len(pool)], 0.05, slot=0 if n == 1 else 1 + (i % n)))
        while pending:
            pending.pop(0).result()

    for name in args.tables.split(","):
        for k, v in base.items():
            setattr(model, k, v)
        for k, v in TABLES[name].items():
            setattr(model, k, v)
        run(30)
        rates = []
        for _ in range(args.repeats):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(args.steps)
            torch.cuda.synchronize()
            rates.append(args.steps / (time.perf_counter() - t0))
        print("%-12s inflight %d: %s img/s (median %.1f)" % (name, n, " ".join("%.1f" % r for r in rates), sorted(rates)[len(rates) // 2]), flush=True)


if __name__ == "__main__":
    main()
