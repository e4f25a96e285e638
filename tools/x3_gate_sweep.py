"""tools/x3_gate_sweep.py -- which f32x3 layer tables keep every exact gate of the GPU tests (three VGG-16 golden fixtures, the predict_one
fixture against the oracle).  Development aid: the default table (nv.DEFAULT_X3_LAYERS_VGG16) was chosen from this output."""
import itertools
import os
import sys

sys.path.insert(0, ".")
import numpy as np
import torch

from fasterrcnn_amd import evaluate as E
from fasterrcnn_amd import synthetic
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models.vgg16 import VGG16Backbone
from oracle import frcnn_oracle as O

dev = torch.device("cuda", 0)
sd = synthetic.vgg16_state_dict(1234)
models = {}
for edge in (True, False):
    m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0), allow_edge_proposals=edge)
    m.load_state_dict(sd, strict=True)
    models[edge] = m.cuda(dev).eval()
CASES = [("600x1000_s0", True), ("224x320_s3", True), ("333x517_s5_noedge", False)]
X6 = ("conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3", "rpn_trunk")


def match(ours, ref):
    if len(ours) == 0 or len(ref) == 0:
        return np.full((len(ref),), np.inf), np.zeros((len(ref),), int)
    d = np.abs(ours[:, None, :4] - ref[None, :, :4]).max(axis=2)
    j = d.argmin(axis=0)
    return d[j, np.arange(len(ref))], j


# the predict_one fixture of tests/test_harness_gpu.py
from PIL import Image
rng = np.random.RandomState(3)
low = rng.randint(0, 256, (12, 16, 3)).astype(np.uint8)
rgb = np.array(Image.fromarray(low, mode="RGB").resize((500, 375), resample=Image.BICUBIC))
path = "/tmp/x3_sweep_image.png"
Image.fromarray(rgb, mode="RGB").save(path)
data = O.preprocess_image(rgb, True, 1.0, [103.939, 116.779, 123.680], [1, 1, 1], 600, False)
ref1 = O.predict(sd, torch.from_numpy(data).unsqueeze(0), 0.3)
n_ref1 = sum(len(v) for v in ref1.values())


def gates(x3, fc):
    out = []
    for tag, edge in CASES:
        m = models[edge]
        m.winograd_x3_layers, m.fc_math_mode = x3, fc
        g = np.load("tests/golden/vgg16_%s.npz" % tag)
        img = synthetic.image(int(g["seed"]), int(g["height"]), int(g["width"])).unsqueeze(0).to(dev)
        p, c, d = m(image_data=img)
        err, j = match(p.cpu().numpy(), g["proposals"])
        det = m.predict(image_data=img, score_threshold=0.05)
        ref = g["detections"]
        n_ok = 0
        for cc in range(1, 21):
            r = ref[ref[:, 0] == cc][:, 1:]
            if len(r) and len(det[cc]):
                e2, j2 = match(det[cc], r)
                n_ok += int(((e2 <= 1e-3) & (np.abs(det[cc][j2, 4] - r[:, 4]) <= 2e-4)).sum())
        out.append("%d/%d %d/%d" % (int((err <= 1e-3).sum()), len(err), n_ok, len(ref)))
    m = models[True]
    det, pil, scale = E.predict_one(m, path, score_threshold=0.3)
    n_ok = 0
    for c in ref1:
        if len(ref1[c]) and len(det[c]):
            e2, j2 = match(det[c], ref1[c])
            n_ok += int(((e2 <= 1e-3) & (np.abs(det[c][j2, 4] - ref1[c][:, 4]) <= 1e-4)).sum())
    out.append("predict_one %d/%d (ours %d)" % (n_ok, n_ref1, sum(len(v) for v in det.values())))
    return out


base = ("conv5_1", "conv5_2", "conv5_3", "rpn_trunk")
configs = [(base, "f32x3")]
if len(sys.argv) > 1 and sys.argv[1] == "wide":
    rest = ("conv4_1", "conv4_2", "conv4_3")
    for r in range(1, 4):
        for extra in itertools.combinations(rest, r):
            configs.append((tuple(extra) + base, "f32x3"))
    # tables without some of the conv5 / trunk layers but with conv4_2 / conv4_3
    for drop in base:
        configs.append((("conv4_2", "conv4_3") + tuple(n for n in base if n != drop), "f32x3"))
        configs.append((("conv4_1", "conv4_2", "conv4_3") + tuple(n for n in base if n != drop), "f32x3"))
    configs.append((("conv4_2", "conv4_3"), "f32x3"))
    configs.append((("conv4_1", "conv4_2", "conv4_3"), "f32x3"))
else:
    configs = [((), "f32x6"), ((), "f32x3")] + [((n,), "f32x3") for n in X6] + configs + [(("conv4_1",) + base, "f32x3"), (X6, "f32x3"), (X6, "f32x6")]
for x3, fc in configs:
    print("x3 = %-60s fc %s | %s" % (",".join(x3) or "-", fc, " | ".join(gates(x3, fc))))
