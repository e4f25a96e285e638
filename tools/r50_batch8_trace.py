"""tools/r50_batch8_trace.py -- bench.py's configs[2] leg (ResNet-50, true batches of 8 through the feature extractor, two batches in flight) for a
kernel trace:  rocprofv3 --kernel-trace --stats -d DIR -- python tools/r50_batch8_trace.py [seconds]"""
import sys
import time
import torch
sys.path.insert(0, ".")
from fasterrcnn_amd import _native as nv, synthetic
from fasterrcnn_amd.models import resnet
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel

nv.require_gpu()
dev = torch.device("cuda", 0)
m = FasterRCNNModel(num_classes=21, backbone=resnet.ResNetBackbone(resnet.Architecture.ResNet50))
m.load_state_dict(synthetic.resnet_state_dict(1234, "ResNet50"), strict=True)
m = m.cuda(dev).eval()
batch = torch.cat([synthetic.image_rgb(s).unsqueeze(0).to(dev) for s in range(8)], dim=0)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
if len(sys.argv) > 2:
    m.bottleneck_g3 = sys.argv[2]            # "off" | "backbone" (default) | "all"
if len(sys.argv) > 3:
    m.batch_head = sys.argv[3] == "1"       # the per-RoI head of the whole batch as one set of launches (default) / per image
pend, lane, n = [], 0, 0
t0 = time.perf_counter()
while time.perf_counter() - t0 < secs:
    if len(pend) == 2:
        for h in pend.pop(0):
            h.result()
    pend.append(m.predict_batch_async(batch, 0.05, lane=lane))
    lane ^= 1
    n += 8
while pend:
    for h in pend.pop(0):
        h.result()
torch.cuda.synchronize()
print("images", n, "images/sec %.1f" % (n / (time.perf_counter() - t0)))
