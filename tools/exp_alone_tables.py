"""tools/exp_alone_tables.py -- one image at a time (slot 0: forward / predict): which of the 512-channel f32x3 layers should run in the
one-launch form there?  Times predict() loops for alone_winograd_x3f_layers = (), the 37x62 layers, all seven -- interleaved, same box."""
import sys
import time

sys.path.insert(0, ".")
import torch

from fasterrcnn_amd import _native as nv, synthetic
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models.vgg16 import VGG16Backbone


def main():
    nv.require_gpu()
    m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
    m.load_state_dict(synthetic.vgg16_state_dict(1234), strict=True)
    m = m.cuda().eval()
    imgs = [synthetic.image(s).unsqueeze(0).cuda() for s in range(4)]
    tables = {"none": (), "conv5+rpn": nv.DEFAULT_ALONE_X3F_LAYERS_VGG16, "all seven": nv.DEFAULT_INFLIGHT_X3F_LAYERS_VGG16}
    res = {k: [] for k in tables}
    for rep in range(4):
        for name, tab in tables.items():
            m.alone_winograd_x3f_layers = tab
            for i in range(6):
                m.predict(imgs[i % 4], 0.05)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 60
            for i in range(n):
                m.predict(imgs[i % 4], 0.05)
            torch.cuda.synchronize()
            res[name].append(n / (time.perf_counter() - t0))
    for name, v in res.items():
        print("%-10s images/sec one at a time: %s  median %.1f" % (name, " ".join("%.1f" % x for x in v), sorted(v)[len(v) // 2]))


if __name__ == "__main__":
    main()
