import sys, json, os, time
sys.path.insert(0, ".")
import torch
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from fasterrcnn_amd import synthetic
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models import resnet
import numpy as np
dev = torch.device("cuda", 0)
m = FasterRCNNModel(num_classes=21, backbone=resnet.ResNetBackbone(resnet.Architecture.ResNet50))
m.load_state_dict(synthetic.resnet_state_dict(1234, "ResNet50"), strict=True)
m = m.cuda(dev).eval()
pool = [synthetic.image_rgb(s).unsqueeze(0).to(dev) for s in range(8)]
g = np.load("tests/golden/resnet50_600x1000_s0.npz")
def match(a, b):
    d = np.abs(a[:, None, :4] - b[None, :, :4]).max(axis=2)
    return d.min(axis=0)
def run(n, nslots=8):
    pend = []
    for i in range(n):
        if len(pend) == nslots:
            pend.pop(0).result()
        pend.append(m.predict_async(pool[i % 8], 0.05, slot=1 + (i % nslots)))
    while pend:
        pend.pop(0).result()
for name, x6c, layers, arith, x3l in (("x6 head (default)", "head", (), "f32x6", ()), ("x3 head", "head", (), "f32x3", ()),
                                      ("x3 head + x3 trunk", "head", ("rpn_trunk",), "f32x3", ("rpn_trunk",)),
                                      ("x6 all + x6 trunk", "all", ("rpn_trunk",), "f32x6", ()),
                                      ("x3 all", "all", (), "f32x3", ()),
                                      ("x3 all + x3 trunk", "all", ("rpn_trunk",), "f32x3", ("rpn_trunk",))):
    m.x6_conv1x1_arith = arith
    m.x6_conv1x1 = x6c
    m.winograd_x6_layers = layers
    m.winograd_x3_layers = x3l
    p, c, d = m(image_data=pool[0])
    err = match(p.cpu().numpy(), g["proposals"])
    det = m.predict(image_data=pool[0], score_threshold=0.05)
    ref = g["detections"]
    n_ok = 0
    for cc in range(1, 21):
        r = ref[ref[:, 0] == cc][:, 1:]
        if len(r) and len(det[cc]):
            dd = np.abs(det[cc][:, None, :4] - r[None, :, :4]).max(axis=2)
            j = dd.argmin(axis=0)
            n_ok += int(((dd[j, np.arange(len(r))] <= 1e-3) & (np.abs(det[cc][j, 4] - r[:, 4]) <= 2e-4)).sum())
    t_end = time.perf_counter() + 1.5
    while time.perf_counter() < t_end:
        run(16)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); run(100); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    for i in range(40):
        m.predict(pool[i % 8], score_threshold=0.05)
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    pe = np.abs(p.cpu().numpy() - g["proposals"]).max(axis=1) if p.shape[0] == g["proposals"].shape[0] else np.array([np.inf])
    print("%-22s max row-by-row proposal error %.3g px |" % (name, pe.max()), end=" ")
    print("%-18s proposals %d/%d  detections %d/%d (ours %d) | 8 in flight %.1f img/s | one at a time %.1f img/s" % (
        name, int((err <= 1e-3).sum()), len(err), n_ok, len(ref), sum(len(v) for v in det.values()), 100 / sorted(ts)[2], 40 / t1))
