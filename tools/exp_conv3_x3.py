"""tools/exp_conv3_x3.py -- does it pay to run conv3_2 / conv3_3 (256 -> 256 at 150 x 250: V + M = 614 MB of scratch in the three-launch form)
as f32x3 Winograd layers?  images/sec at 3 images in flight, and the golden gates of the 600x1000 fixture."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
import numpy as np
import torch

from fasterrcnn_amd import _native as nv, synthetic
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models.vgg16 import VGG16Backbone

dev = torch.device("cuda", 0)
m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
m.load_state_dict(synthetic.vgg16_state_dict(1234), strict=True)
m = m.cuda(dev).eval()
pool = [synthetic.image(s).unsqueeze(0).to(dev) for s in range(8)]
g = np.load("tests/golden/vgg16_600x1000_s0.npz")
img = synthetic.image(0, 600, 1000).unsqueeze(0).to(dev)


def run(n, nslots=3):
    pend = []
    for i in range(n):
        if len(pend) == nslots:
            pend.pop(0).result()
        pend.append(m.predict_async(pool[i % 8], 0.05, slot=1 + (i % nslots)))
    while pend:
        pend.pop(0).result()


X6, X3 = nv.DEFAULT_X6_LAYERS_VGG16, nv.DEFAULT_X3_LAYERS_VGG16
for name, extra in (("default", ()), ("+ conv3_3", ("conv3_3",)), ("+ conv3_2, conv3_3", ("conv3_2", "conv3_3"))):
    m.winograd_x6_layers = extra + X6
    m.winograd_x3_layers = extra + X3
    p, c, d = m(image_data=img)
    err = np.abs(p.cpu().numpy() - g["proposals"]).max(axis=1)
    t_end = time.perf_counter() + 1.0
    while time.perf_counter() < t_end:
        run(12)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); run(90); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    for i in range(30):
        m.predict(pool[i % 8], score_threshold=0.05)
    torch.cuda.synchronize()
    print("%-22s 3 in flight %.1f img/s | one at a time %.1f | proposals within 1e-3 px %d/300 (max %.3g px)" % (
        name, 90 / sorted(ts)[2], 30 / (time.perf_counter() - t0), int((err <= 1e-3).sum()), err.max()))
