"""tools/exp_roi_cells.py -- how many feature-map cells RoI pooling reads on the workload's proposals: per bin window (what roi_pool_x3t_rows_kernel
reads: adjacent bins overlap by the ceil / floor of their edges) against the RoI's own area (what a separable row-then-column maximum would read)."""
import sys, math
sys.path.insert(0, ".")
import numpy as np
import torch
from fasterrcnn_amd import synthetic
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models.vgg16 import VGG16Backbone

dev = torch.device("cuda", 0)
m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
m.load_state_dict(synthetic.vgg16_state_dict(1234), strict=True)
m = m.cuda(dev).eval()
fh, fw, pooled, scale = 37, 62, 7, 1.0 / 16.0
for seed in range(4):
    img = synthetic.image(seed).unsqueeze(0).to(dev)
    p, c, d = m(image_data=img)
    rois = p.cpu().numpy().astype(np.float32)
    tot_bins = tot_area = tot_rowsep = 0
    for y1, x1, y2, x2 in rois:
        rs_h, rs_w = int(round(float(np.float32(y1) * np.float32(scale)))), int(round(float(np.float32(x1) * np.float32(scale))))
        re_h, re_w = int(round(float(np.float32(y2) * np.float32(scale)))), int(round(float(np.float32(x2) * np.float32(scale))))
        roi_h, roi_w = max(re_h - rs_h + 1, 1), max(re_w - rs_w + 1, 1)
        bh, bw = np.float32(roi_h) / np.float32(pooled), np.float32(roi_w) / np.float32(pooled)
        hs = [min(max(int(math.floor(float(np.float32(i) * bh))) + rs_h, 0), fh) for i in range(pooled)]
        he = [min(max(int(math.ceil(float(np.float32(i + 1) * bh))) + rs_h, 0), fh) for i in range(pooled)]
        ws = [min(max(int(math.floor(float(np.float32(i) * bw))) + rs_w, 0), fw) for i in range(pooled)]
        we = [min(max(int(math.ceil(float(np.float32(i + 1) * bw))) + rs_w, 0), fw) for i in range(pooled)]
        sh = sum(max(0, e - s) for s, e in zip(hs, he)); sw = sum(max(0, e - s) for s, e in zip(ws, we))
        tot_bins += sh * sw
        uh = max(0, max(he) - min(hs)); uw = max(0, max(we) - min(ws))
        tot_area += uh * uw
        tot_rowsep += uh * sw
    n = len(rois)
    print("image %d: %d RoIs | cells read per bin window: %.0f per RoI = %.0f MB per image at 2 KB per cell | RoI area: %.0f cells per RoI = %.0f MB (x%.2f less) | rows once, columns per bin: %.0f MB" % (
        seed, n, tot_bins / n, tot_bins * 2048 / 1e6, tot_area / n, tot_area * 2048 / 1e6, tot_bins / max(1, tot_area), tot_rowsep * 2048 / 1e6))
