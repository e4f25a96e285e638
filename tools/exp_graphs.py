"""Experiment: one image at a time with and without hipGraph replay (development aid)."""
import os, sys, time
sys.path.insert(0, ".")
import torch
from fasterrcnn_amd import synthetic
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models.vgg16 import VGG16Backbone
dev = torch.device("cuda", 0)
m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
m.load_state_dict(synthetic.vgg16_state_dict(1234), strict=True)
m = m.cuda(dev).eval()
pool = [synthetic.image(s).unsqueeze(0).to(dev) for s in range(4)]
def measure():
    t_end = time.perf_counter() + 1.5
    while time.perf_counter() < t_end:
        m.predict(pool[0], 0.05)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(100):
            m.predict(pool[i % 4], 0.05)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return 100 / sorted(ts)[2]
for g in (False, True, False, True):
    m.use_hip_graphs = g
    d0 = m.predict(pool[1], 0.05)
    print("graphs", g, "%.1f img/s" % measure(), sum(len(v) for v in d0.values()))
