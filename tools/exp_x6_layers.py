"""Experiment: in-flight images/sec of VGG-16 for several x6 layer tables (development aid)."""
import os, sys, time
sys.path.insert(0, ".")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from fasterrcnn_amd import synthetic, _native as nv
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models.vgg16 import VGG16Backbone
dev = torch.device("cuda", 0)
m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
m.load_state_dict(synthetic.vgg16_state_dict(1234), strict=True)
m = m.cuda(dev).eval()
pool = [synthetic.image(s).unsqueeze(0).to(dev) for s in range(8)]
def run(n, nslots):
    pend = []
    for i in range(n):
        if len(pend) == nslots:
            pend.pop(0).result()
        pend.append(m.predict_async(pool[i % 8], 0.05, slot=1 + (i % nslots)))
    while pend:
        pend.pop(0).result()
def measure(nslots=3):
    t_end = time.perf_counter() + 1.5
    while time.perf_counter() < t_end:
        run(12, nslots)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); run(150, nslots); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return 150 / sorted(ts)[3]
D = nv.DEFAULT_X6_LAYERS_VGG16
tables = [("default", D), ("+conv3_3", D + ("conv3_3",)), ("+conv3_2,3_3", D + ("conv3_2", "conv3_3")), ("none", ()), ("default again", D)]
for name, tab in tables:
    m.winograd_x6_layers = tab
    for tiles in (2,):
        m.inflight_x6_gemm_tiles = tiles
        print("%-16s tiles %d: if3 %.1f  if4 %.1f img/s" % (name, tiles, measure(3), measure(4)), flush=True)
