"""tools/exp_r50_lane_streams.py -- ResNet-50 as batches of 8 (predict_batch_async, two batches in flight): which streams the 8 per-image tails of a
batch run on (FRCNN_LANE_STREAMS = own | slots | n: a knob runtime.slot_stream had for this measurement; "slots" is what it does since).  Bursts of 24 images (bench.py's driver form) and of 200.
(development aid; python tools/exp_r50_lane_streams.py)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
import torch
from fasterrcnn_amd import synthetic
from fasterrcnn_amd.models import resnet
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel

dev = torch.device("cuda", 0)
m = FasterRCNNModel(num_classes=21, backbone=resnet.ResNetBackbone(resnet.Architecture.ResNet50))
m.load_state_dict(synthetic.resnet_state_dict(1234, "ResNet50"), strict=True)
m = m.cuda(dev).eval()
if "all" in sys.argv[1:]:
    m.bottleneck_g3 = "all"
if "batch_head" in sys.argv[1:]:
    m.batch_head = True
batch = torch.cat([synthetic.image_rgb(s).unsqueeze(0).to(dev) for s in range(8)], dim=0)


def run(n):
    pend, lane = [], 0
    for _ in range((n + 7) // 8):
        if len(pend) == 2:
            for h in pend.pop(0):
                h.result()
        pend.append(m.predict_batch_async(batch, 0.05, lane=lane))
        lane ^= 1
    while pend:
        for h in pend.pop(0):
            h.result()


def measure(n, reps):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(n)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return n / sorted(ts)[len(ts) // 2]


t_end = time.perf_counter() + 2.0
while time.perf_counter() < t_end:
    run(16)
print("FRCNN_LANE_STREAMS=%s g3=%s batch_head=%s: bursts of 24 %.1f images/sec, bursts of 200 %.1f" % (os.environ.get("FRCNN_LANE_STREAMS", "own"), m.bottleneck_g3, m.batch_head, measure(24, 15), measure(200, 5)))
