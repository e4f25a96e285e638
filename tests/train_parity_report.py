"""
Stage-by-stage parity report of FasterRCNNModel.train_step (GPU, through the C ABI) against
oracle/train_oracle.py run on this machine's CPU with the same seeds.  Not a test (no assertions): it prints the
relative error of every intermediate and gradient so a discrepancy can be localised.
  python tests/train_parity_report.py [height width seed]
"""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fasterrcnn_amd import synthetic, training as T                      # noqa: E402
from fasterrcnn_amd.datasets.training_sample import Box                   # noqa: E402
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel             # noqa: E402
from fasterrcnn_amd.models.vgg16 import VGG16Backbone, _LAYERS            # noqa: E402
from oracle import frcnn_oracle as O                                      # noqa: E402
from oracle import train_oracle as TO                                     # noqa: E402


def rel(a, b):
    a = a.detach().cpu().double().reshape(-1)
    b = b.detach().cpu().double().reshape(-1)
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def hwc(x):          # oracle (1,C,H,W) -> [H][W][C]
    return x[0].permute(1, 2, 0)


def main():
    h, w, seed = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (352, 480, 4)
    lr, mom, wd = 1e-6, 0.9, 5e-4
    sd = synthetic.vgg16_state_dict(1234)
    img = synthetic.image(seed, h, w).unsqueeze(0)
    gts = synthetic.ground_truth(seed, h, w)
    am, vm = O.generate_anchor_maps((3, h, w), (512, h // 16, w // 16), 16)
    gc = np.stack([k for _, k in gts]); gcls = np.array([c for c, _ in gts])
    rmap, obj, bg = O.generate_rpn_map(am, vm, gc)
    random.seed(100 + seed); torch.manual_seed(100 + seed)
    od = {}
    ol, og, onew, _ = TO.train_step(sd, img, am, vm, torch.from_numpy(rmap).unsqueeze(0), obj, bg, gc, gcls, 21, lr, mom, wd, None, detail=od)
    model = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
    model.load_state_dict(sd, strict=True)
    model = model.cuda()
    random.seed(100 + seed); torch.manual_seed(100 + seed)
    d = {}
    loss = T.train_step(model, T.SGD(lr, mom, wd), img.cuda(), am, vm, torch.from_numpy(rmap).unsqueeze(0), [obj], [bg],
                        [[Box(c, "x", k) for c, k in gts]], detail=d)
    print("losses gpu   ", loss)
    print("losses oracle", ol)
    print("sample idx equal:", np.array_equal(d["sample_idx"].numpy(), od["proposal_sample_indices"]),
          " rpn sample equal:", np.array_equal(d["rpn_sample"].cpu().numpy(), od["rpn_sample_flat"]))
    go = od["grad_of"]
    rows = [("fm", d["fm"], hwc(od["feature_map"])), ("trunk", d["trunk"], hwc(od["rpn_trunk"])),
            ("roi_out", d["roi_out"].reshape(-1, 7, 7, 512), od["pooled"].permute(0, 2, 3, 1)),
            ("h1", d["h1"], od["fc1"]), ("h2", d["h2"], od["fc2"]), ("classes", d["classes"], od["classes"]),
            ("deltas", d["deltas"], od["deltas"]),
            ("dfm_roi", d["dfm_roi"], None), ("dfm(total)", d["dfm"], hwc(go["feature_map"]))]
    rows += [("dh2 (pre-mask)", d["dh2"], go["fc2"]), ("dh1 (pre-mask)", d["dh1"], go["fc1"]),
             ("droi", d["droi"].reshape(-1, 7, 7, 512), go["pooled"].permute(0, 2, 3, 1))]
    for nm, a, b in (("h1", d["h1"], od["fc1"]), ("h2", d["h2"], od["fc2"]), ("fm", d["fm"], hwc(od["feature_map"]))):
        print("ReLU mask flips in %-3s: %d of %d" % (nm, int(((a.cpu() > 0) != (b > 0)).sum()), b.numel()))
    for name, a, b in rows:
        if b is not None:
            print("%-14s rel err %.3g" % (name, rel(a, b)))
    from tests.test_train_gpu import canonical_grads
    cg = canonical_grads(d["grads"])
    for k in og:
        a = cg[k].detach().cpu().double().reshape(-1); b = og[k].double().reshape(-1)
        dd = (a - b).abs() / float(b.abs().max())
        print("grad %-50s max %.3g  median %.3g  p99 %.3g  L2 %.3g   |g| %.4g" % (
            k[-50:], float(dd.max()), float(dd.median()), float(dd.kthvalue(int(0.99 * dd.numel())).values),
            float((a - b).norm() / b.norm()), float(b.norm())))


if __name__ == "__main__":
    main()
