"""tools/exp_evaluate_h2d.py -- the upload inside evaluate() (fasterrcnn_amd/evaluate.py; the reference's `t.from_numpy(image).unsqueeze(0).cuda()`,
__main__.py:78-86): preprocessed float32 (3, 600, 1000) numpy images, 7.2 MB each, (a) `.to(device)` from pageable memory as evaluate() did,
(b) from a worker thread, a few images ahead (evaluate.BackgroundUploader: evaluate()'s form since; a ring of pinned staging buffers measured 100 images/sec:
the host's copy into pinned memory).  (development aid)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
import numpy as np
import torch
from fasterrcnn_amd import synthetic
from fasterrcnn_amd import evaluate as ev
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models.vgg16 import VGG16Backbone

dev = torch.device("cuda", 0)
m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
m.load_state_dict(synthetic.vgg16_state_dict(1234), strict=True)
m = m.cuda(dev).eval()
images = [synthetic.image(s).numpy().copy() for s in range(8)]       # (3, 600, 1000) float32, pageable


def pageable(n):
    for i in range(n):
        yield i, torch.from_numpy(images[i % 8]).unsqueeze(0).to(dev), None


def staged(n):
    return ev.BackgroundUploader(dev, depth=4).iterate((i, images[i % 8], None) for i in range(n))


def measure(gen, n=200):
    ev.evaluate_stream(m, gen(16), inflight=4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ev.evaluate_stream(m, gen(n), inflight=4)
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


t_end = time.perf_counter() + 2.0
while time.perf_counter() < t_end:
    ev.evaluate_stream(m, pageable(8), inflight=4)
for rep in range(2):
    print("pageable .to(device): %.1f images/sec" % measure(pageable))
    print("background uploader thread: %.1f images/sec" % measure(staged))
