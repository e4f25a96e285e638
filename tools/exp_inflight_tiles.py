"""tools/exp_inflight_tiles.py -- images/sec at 3 images in flight for the three block-tile choices of the split-operand GEMMs
(frcnn_forward_params.x6_gemm_tiles: 0 cost model, 1 = 320 x 256, 2 = 160 x 128) with the default f32x6 / f32x3 tables."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
import torch

from fasterrcnn_amd import synthetic
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models.vgg16 import VGG16Backbone

dev = torch.device("cuda", 0)
m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
m.load_state_dict(synthetic.vgg16_state_dict(1234), strict=True)
m = m.cuda(dev).eval()
pool = [synthetic.image(s).unsqueeze(0).to(dev) for s in range(8)]


def run(n, nslots=3):
    pend = []
    for i in range(n):
        if len(pend) == nslots:
            pend.pop(0).result()
        pend.append(m.predict_async(pool[i % 8], 0.05, slot=1 + (i % nslots)))
    while pend:
        pend.pop(0).result()


for rep in range(2):
    for tiles in (2, 0, 1):
        m.inflight_x6_gemm_tiles = tiles
        t_end = time.perf_counter() + 1.0
        while time.perf_counter() < t_end:
            run(12)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); run(90); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print("x6_gemm_tiles=%d: %.1f images/sec" % (tiles, 90 / sorted(ts)[2]))
