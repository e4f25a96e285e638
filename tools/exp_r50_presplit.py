"""tools/exp_r50_presplit.py -- ResNet-50, 8 images in flight: the bottleneck weight packs as float32 (g3 = 1: split in every block of every launch) against
pre-split at pack time (g3 = 2, frcnn_pack_conv_x3g_weights); same detections required."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from fasterrcnn_amd import synthetic
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models import resnet as _resnet

dev = torch.device("cuda", 0)
pool = [synthetic.image_rgb(s).unsqueeze(0).to(dev) for s in range(8)]
res = {}
for rep in range(2):
    for pre in ((True, False) if "rev" in sys.argv else (False, True)):
        _resnet.G3_PRESPLIT = pre
        m = FasterRCNNModel(num_classes=21, backbone=_resnet.ResNetBackbone(_resnet.Architecture.ResNet50))
        m.load_state_dict(synthetic.resnet_state_dict(1234, "ResNet50"), strict=True)
        m = m.cuda(dev).eval()
        if len(sys.argv) > 1 and sys.argv[1] != "rev":
            m.bottleneck_g3 = sys.argv[1]
        det = m.predict(pool[0], 0.05)
        res[pre] = det

        def run(n):
            pend = []
            for i in range(n):
                if len(pend) == 8:
                    pend.pop(0).result()
                pend.append(m.predict_async(pool[i % 8], 0.05, slot=1 + (i % 8)))
            while pend:
                pend.pop(0).result()
        run(32)
        t_end = time.perf_counter() + 1.0
        while time.perf_counter() < t_end:
            run(16)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); run(200); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print("pre-split weight packs %-5s (g3 %s): %.1f images/sec" % (pre, getattr(m, "bottleneck_g3", "?"), 200 / sorted(ts)[2]), flush=True)
        del m
same = all(np.array_equal(res[False][c], res[True][c]) for c in res[False])
print("detections identical:", same)
