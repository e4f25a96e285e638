"""
Times FasterRCNNModel.train_step (VGG-16, one 600x1000 synthetic sample per step) on cuda:0 and prints one JSON line.
Informational: BASELINE.json's headline metric is inference throughput (bench.py); this reports the cost of
SURVEY.md section 8 row f3 on the same hardware.
  python tools/train_bench.py [--steps 20] [--warmup 3] [--height 600 --width 1000] [--lr 1e-6]
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasterrcnn_amd import synthetic, training                                   # noqa: E402
from fasterrcnn_amd.datasets.training_sample import Box                          # noqa: E402
from fasterrcnn_amd.models import anchors                                        # noqa: E402
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel                    # noqa: E402
from fasterrcnn_amd.models.vgg16 import VGG16Backbone                            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--width", type=int, default=1000)
    ap.add_argument("--lr", type=float, default=1e-6)
    ap.add_argument("--pool", type=int, default=4, help="distinct synthetic samples cycled through")
    ap.add_argument("--backbone", type=str, default="vgg16", choices=["vgg16", "resnet50", "resnet101", "resnet152"])
    ap.add_argument("--math", type=str, default=None, choices=["f32", "f32_winograd"],
                    help="default: the model's (f32_winograd: forward / data-gradient convolutions of the wide layers as Winograd layers)")
    ap.add_argument("--grad-math", type=str, default="f32", choices=["f32", "bf16"],
                    help="arithmetic of the gradient GEMMs (FasterRCNNModel.grad_math); bf16 = BASELINE configs[4]'s reduced-precision step")
    ap.add_argument("--roi", type=str, default="pool", choices=["pool", "align"])
    ap.add_argument("--host-clocks", action="store_true", help="also report what the host needs to ENQUEUE a step (training.HOST_CLOCKS)")
    args = ap.parse_args()
    h, w = args.height, args.width
    if args.backbone == "vgg16":
        model = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0), roi_pooling=args.roi)
        model.load_state_dict(synthetic.vgg16_state_dict(1234), strict=True)
        make_image = synthetic.image
    else:
        from fasterrcnn_amd.models import resnet
        arch = {"resnet50": "ResNet50", "resnet101": "ResNet101", "resnet152": "ResNet152"}[args.backbone]
        model = FasterRCNNModel(num_classes=21, backbone=resnet.ResNetBackbone(getattr(resnet.Architecture, arch)), roi_pooling=args.roi)
        model.load_state_dict(synthetic.resnet_state_dict(1234, arch), strict=True)
        make_image = synthetic.image_rgb
    model = model.cuda()
    model.grad_math = args.grad_math
    if args.math is not None:
        model.math_mode = args.math
    am, vm = anchors.generate_anchor_maps((3, h, w), model.backbone.compute_feature_map_shape((3, h, w)), 16)
    samples = []
    for seed in range(args.pool):
        gts = synthetic.ground_truth(seed, h, w)
        boxes = [Box(c, "x", k) for c, k in gts]
        rmap, obj, bg = anchors.generate_rpn_map(am, vm, boxes)
        samples.append((make_image(seed, h, w).unsqueeze(0).cuda(), torch.from_numpy(rmap).unsqueeze(0).cuda(), obj, bg, boxes))
    opt = training.create_optimizer(model, learning_rate=args.lr)
    random.seed(0); torch.manual_seed(0)
    losses = []

    def step(i):
        img, rmap, obj, bg, boxes = samples[i % len(samples)]
        return model.train_step(opt, img, am, vm, rmap, [obj], [bg], [boxes])

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if args.host_clocks:
        training.HOST_CLOCKS = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        losses.append(step(i).total)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    host = {}
    if args.host_clocks:
        hc = np.array(training.HOST_CLOCKS)
        host = {"host_ms_per_step": {"enqueue_until_sync_1": round(1e3 * float(np.median(hc[:, 1] - hc[:, 0])), 3),
                                     "wait_sync_1": round(1e3 * float(np.median(hc[:, 2] - hc[:, 1])), 3),
                                     "enqueue_until_sync_2": round(1e3 * float(np.median(hc[:, 3] - hc[:, 2])), 3),
                                     "wait_sync_2": round(1e3 * float(np.median(hc[:, 4] - hc[:, 3])), 3)}}
    print(json.dumps({**host, "metric": "train_step (%s Faster R-CNN, %dx%d, batch 1)" % (args.backbone, h, w), "ms_per_step": 1e3 * dt / args.steps,
                      "steps_per_sec": args.steps / dt, "steps": args.steps, "warmup": args.warmup, "dtype": "f32", "grad_math": model.grad_math, "roi": args.roi, "math": model.math_mode,
                      "first_total_loss": losses[0], "last_total_loss": losses[-1], "data": "synthetic"}))


if __name__ == "__main__":
    main()
