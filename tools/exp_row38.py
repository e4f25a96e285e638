"""tools/exp_row38.py -- which layer's rounding order moved golden proposal 38 of the 600x1000 fixture from 0.85e-3 px (round 4's slot-0 table: the
512-channel layers in the THREE-launch f32x3 form, output transform rows first) to 1.04e-3 px (round 5: every layer in the ONE-launch form,
columns first)?  The two forms share operands, products and accumulation order (tests/test_gemm_x3t_gpu.py); they differ in the float32
rounding order of the 2 x 2 output sums only.  Each of the seven layers is switched back to the three-launch form alone, and all seven."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from fasterrcnn_amd import _native as nv, synthetic
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models.vgg16 import VGG16Backbone

nv.require_gpu()
g = np.load("tests/golden/vgg16_600x1000_s0.npz")
model = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
model.load_state_dict(synthetic.vgg16_state_dict(1234), strict=True)
model = model.cuda().eval()
img = synthetic.image(int(g["seed"]), int(g["height"]), int(g["width"])).unsqueeze(0).cuda()
ALL = tuple(nv.DEFAULT_ALONE_X3F_LAYERS_VGG16)


def measure(one_launch):
    model.alone_winograd_x3f_layers = one_launch
    props, _, _ = model(image_data=img)
    err = np.abs(props.cpu().numpy().astype(np.float64) - g["proposals"].astype(np.float64)).max(axis=1)
    return err


base = measure(ALL)
print("all seven one-launch (the default):   rows within 1e-3 px %d / %d, row 38 at %.4e px, worst row %d at %.4e" % ((base <= 1e-3).sum(), len(base), base[38], base.argmax(), base.max()))
for n in ALL:
    e = measure(tuple(x for x in ALL if x != n))
    print("%-10s three-launch, six one-launch: rows within 1e-3 px %d / %d, row 38 at %.4e px, worst row %d at %.4e" % (n, (e <= 1e-3).sum(), len(e), e[38], e.argmax(), e.max()))
e = measure(())
print("all seven three-launch (round 4):     rows within 1e-3 px %d / %d, row 38 at %.4e px, worst row %d at %.4e" % ((e <= 1e-3).sum(), len(e), e[38], e.argmax(), e.max()))
ref = g["proposals"][38]
print("golden proposal 38:", ref, "side lengths", ref[2] - ref[0], ref[3] - ref[1])
