"""tools/x3_compare.py -- VGG-16: golden parity and images/sec of the f32x6 / f32x3 arithmetic choices (development aid)."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
import numpy as np
import torch

from fasterrcnn_amd import synthetic
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models.vgg16 import VGG16Backbone

dev = torch.device("cuda", 0)
sd = synthetic.vgg16_state_dict(1234)
models = {}
for edge in (True, False):
    m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0), allow_edge_proposals=edge)
    m.load_state_dict(sd, strict=True)
    models[edge] = m.cuda(dev).eval()
pool = [synthetic.image(s).unsqueeze(0).to(dev) for s in range(8)]
CASES = [("600x1000_s0", True), ("224x320_s3", True), ("333x517_s5_noedge", False)]


def match(ours, ref):
    if len(ours) == 0 or len(ref) == 0:
        return np.full((len(ref),), np.inf), np.zeros((len(ref),), int)
    d = np.abs(ours[:, None, :4] - ref[None, :, :4]).max(axis=2)
    j = d.argmin(axis=0)
    return d[j, np.arange(len(ref))], j


def run(m, n, nslots=3):
    pend = []
    for i in range(n):
        if len(pend) == nslots:
            pend.pop(0).result()
        pend.append(m.predict_async(pool[i % 8], 0.05, slot=1 + (i % nslots)))
    while pend:
        pend.pop(0).result()


X6 = ("conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3", "rpn_trunk")
CONFIGS = [("x6 conv, x6 fc (round 3)", (), "f32x6"), ("x6 conv, x3 fc", (), "f32x3"),
           ("x3 conv5 + trunk, x3 fc", X6[3:], "f32x3"), ("x3 conv4_1 + conv5 + trunk, x3 fc", (X6[0],) + X6[3:], "f32x3"), ("x3 all, x3 fc", X6, "f32x3")]
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
for name, x3, fc in CONFIGS:
    line = []
    for tag, edge in CASES:
        m = models[edge]
        m.winograd_x3_layers, m.fc_math_mode = x3, fc
        g = np.load("tests/golden/vgg16_%s.npz" % tag)
        img = synthetic.image(int(g["seed"]), int(g["height"]), int(g["width"])).unsqueeze(0).to(dev)
        p, c, d = m(image_data=img)
        err, j = match(p.cpu().numpy(), g["proposals"])
        ok = err <= 1e-3
        cerr = float(np.abs(c.cpu().numpy()[j[ok]] - g["classes"][ok]).max())
        det = m.predict(image_data=img, score_threshold=0.05)
        ref = g["detections"]
        n_ok = 0
        for cc in range(1, 21):
            r = ref[ref[:, 0] == cc][:, 1:]
            if len(r) and len(det[cc]):
                e2, j2 = match(det[cc], r)
                n_ok += int(((e2 <= 1e-3) & (np.abs(det[cc][j2, 4] - r[:, 4]) <= 2e-4)).sum())
        line.append("%s: props %d/%d, |dprob| %.1e, dets %d/%d (ours %d)" % (tag, int(ok.sum()), len(ok), cerr, n_ok, len(ref), sum(len(v) for v in det.values())))
    m = models[True]
    if quick:
        print("%-34s\n    %s" % (name, "\n    ".join(line)))
        continue
    t_end = time.perf_counter() + 1.5
    while time.perf_counter() < t_end:
        run(m, 12)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); run(m, 90); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    for i in range(40):
        m.predict(pool[i % 8], score_threshold=0.05)
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    print("%-34s | 3 in flight %.1f img/s | one at a time %.1f img/s\n    %s" % (name, 90 / sorted(ts)[2], 40 / t1, "\n    ".join(line)))
