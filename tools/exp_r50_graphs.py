"""tools/exp_r50_graphs.py -- is ResNet-50 (BASELINE configs[2]: ~200 launches per image) bound by the host's launch rate?  8 batch-1 images in
flight from one host thread, eager launches against hipGraph replay (model.use_hip_graphs), with the host thread's CPU time per predict_async."""
import sys, time
sys.path.insert(0, ".")
import torch
from fasterrcnn_amd import synthetic
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models import resnet as _resnet
from fasterrcnn_amd.models.vgg16 import VGG16Backbone

dev = torch.device("cuda", 0)
arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
nfl = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if arch == "vgg16":
    m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
    m.load_state_dict(synthetic.vgg16_state_dict(1234), strict=True)
    pool = [synthetic.image(s).unsqueeze(0).to(dev) for s in range(8)]
else:
    m = FasterRCNNModel(num_classes=21, backbone=_resnet.ResNetBackbone(_resnet.Architecture.ResNet50))
    m.load_state_dict(synthetic.resnet_state_dict(1234, "ResNet50"), strict=True)
    pool = [synthetic.image_rgb(s).unsqueeze(0).to(dev) for s in range(8)]
m = m.cuda(dev).eval()


def run(n):
    pend, cpu = [], 0.0
    for i in range(n):
        if len(pend) == nfl:
            pend.pop(0).result()
        c0 = time.thread_time()
        pend.append(m.predict_async(pool[i % len(pool)], 0.05, slot=1 + (i % nfl)))
        cpu += time.thread_time() - c0
    while pend:
        pend.pop(0).result()
    return cpu


for g in (False, True, False, True):
    m.use_hip_graphs = g
    run(4 * nfl)
    t_end = time.perf_counter() + 1.5
    while time.perf_counter() < t_end:
        run(2 * nfl)
    res = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cpu = run(200)
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0, cpu))
    dt, cpu = sorted(res)[2]
    print("%s, %d in flight, hip graphs %-5s: %.1f images/sec, %.0f us wall per image, %.0f us of host-thread CPU per predict_async" % (arch, nfl, g, 200 / dt, 1e6 * dt / 200, 1e6 * cpu / 200), flush=True)
