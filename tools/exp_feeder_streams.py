"""tools/exp_feeder_streams.py -- the headline loop fed from pinned host memory (HostFeeder.submit: H2D of the uint8 frame + device preprocess +
predict) with 1 / 2 / 4 / 8 feeder streams beside the 4 in-flight slots: do the feeder's streams take a hardware pipe away from the slots?
(development aid; python tools/exp_feeder_streams.py)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
import torch
from fasterrcnn_amd import synthetic
from fasterrcnn_amd.evaluate import HostFeeder
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models.vgg16 import VGG16Backbone

dev = torch.device("cuda", 0)
m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
m.load_state_dict(synthetic.vgg16_state_dict(1234), strict=True)
m = m.cuda(dev).eval()
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pool = [synthetic.image(s).unsqueeze(0).to(dev) for s in range(8)]
host_u8 = [synthetic.image_u8(s).pin_memory() for s in range(8)]


def loop(submit, frames, n):
    pend = []
    for i in range(n):
        if len(pend) == NS:
            pend.pop(0).result()
        pend.append(submit(frames[i % len(frames)], 0.05, 1 + (i % NS)))
    while pend:
        pend.pop(0).result()


def measure(fn, n=200, reps=5):
    fn(3 * NS)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn(n)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return n / sorted(ts)[len(ts) // 2]


t_end = time.perf_counter() + 2.0
while time.perf_counter() < t_end:
    loop(lambda f, thr, slot: m.predict_async(f, thr, slot=slot), pool, NS)
print("resident images, %d in flight: %.1f images/sec" % (NS, measure(lambda n: loop(lambda f, thr, slot: m.predict_async(f, thr, slot=slot), pool, n))))
for look in (8, 4, 2, 1, 8):
    feeder = HostFeeder(m, lookahead=look)
    print("feeder streams %d: %.1f images/sec (bursts of 200), %.1f (bursts of 20)" % (look, measure(lambda n: loop(feeder.submit, host_u8, n)), measure(lambda n: loop(feeder.submit, host_u8, n), n=20, reps=15)))
    del feeder
print("resident images again: %.1f images/sec" % measure(lambda n: loop(lambda f, thr, slot: m.predict_async(f, thr, slot=slot), pool, n)))
