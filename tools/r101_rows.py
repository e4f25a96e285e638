"""tools/r101_rows.py -- ResNet-101 600x1000 fixture: are the proposals that miss the 1e-3 px gate the same rows a little off, or other rows?"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from fasterrcnn_amd import synthetic
from fasterrcnn_amd.models import resnet
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel

g = np.load("tests/golden/resnet101_600x1000_s2.npz")
m = FasterRCNNModel(num_classes=21, backbone=resnet.ResNetBackbone(resnet.Architecture.ResNet101))
m.load_state_dict(synthetic.resnet_state_dict(1234, "ResNet101"), strict=True)
m = m.cuda().eval()
img = synthetic.image_rgb(int(g["seed"]), 600, 1000).unsqueeze(0).cuda()
p, c, d = m(image_data=img)
ours, ref = p.cpu().numpy().astype(np.float64), g["proposals"].astype(np.float64)
dist = np.abs(ours[:, None, :] - ref[None, :, :]).max(axis=2)
e = dist.min(axis=0)
j = dist.argmin(axis=0)
print("golden proposals matched within 1e-3 px: %d, within 1e-2: %d, within 0.1: %d, within 1 px: %d of %d" % (
    (e <= 1e-3).sum(), (e <= 1e-2).sum(), (e <= 0.1).sum(), (e <= 1.0).sum(), len(ref)))
print("matched at the SAME row index (within 1 px): %d" % int(((j == np.arange(len(ref))) & (e <= 1.0)).sum()))
miss = np.nonzero(e > 1e-3)[0]
for k in miss[:16]:
    print("  golden row %3d: nearest ours row %3d at %.3g px  %s" % (k, j[k], e[k], np.round(ref[k], 2).tolist()))
