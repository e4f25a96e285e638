"""tools/exp_r50_threads.py -- ResNet-50 (configs[2]) with 8 batch-1 images in flight driven by 1, 2 or 4 host threads (ctypes releases the GIL
inside the C call that enqueues an image's ~110 launches): is the one submitting thread the bound?  Also prints where the thread's wall time goes."""
import sys, time, threading
sys.path.insert(0, ".")
import torch
from fasterrcnn_amd import synthetic
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models import resnet as _resnet

dev = torch.device("cuda", 0)
m = FasterRCNNModel(num_classes=21, backbone=_resnet.ResNetBackbone(_resnet.Architecture.ResNet50))
m.load_state_dict(synthetic.resnet_state_dict(1234, "ResNet50"), strict=True)
m = m.cuda(dev).eval()
pool = [synthetic.image_rgb(s).unsqueeze(0).to(dev) for s in range(8)]


def worker(slots, n, stats):
    torch.cuda.set_device(dev)
    pend, t_sub, t_res = [], 0.0, 0.0
    for i in range(n):
        if len(pend) == len(slots):
            t0 = time.perf_counter(); pend.pop(0).result(); t_res += time.perf_counter() - t0
        t0 = time.perf_counter()
        pend.append(m.predict_async(pool[i % len(pool)], 0.05, slot=slots[i % len(slots)]))
        t_sub += time.perf_counter() - t0
    while pend:
        t0 = time.perf_counter(); pend.pop(0).result(); t_res += time.perf_counter() - t0
    stats.append((t_sub / n, t_res / n))


def run(nthreads, total=192):
    per = 8 // nthreads
    groups = [[1 + g * per + j for j in range(per)] for g in range(nthreads)]
    stats = []
    ths = [threading.Thread(target=worker, args=(g, total // nthreads, stats)) for g in groups]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return total / dt, stats


for nt in (1, 2, 4, 1, 2, 4):
    run(nt, 64)
    rate, stats = run(nt)
    print("%d host thread(s), 8 images in flight: %.1f images/sec | per image and thread: predict_async %.0f us wall, result() %.0f us wall" % (
        nt, rate, 1e6 * sum(s[0] for s in stats) / len(stats), 1e6 * sum(s[1] for s in stats) / len(stats)), flush=True)
