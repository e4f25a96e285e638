import sys; sys.path.insert(0, ".")
import numpy as np, torch
from fasterrcnn_amd import synthetic, _native as nv
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models.vgg16 import VGG16Backbone
m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
m.load_state_dict(synthetic.vgg16_state_dict(1234), strict=True)
m = m.cuda().eval()
img = synthetic.image(0, 600, 1000).unsqueeze(0).cuda()
for name, layers in (("all_x3", nv.DEFAULT_X6_LAYERS_VGG16), ("conv4_2_only", ("conv4_2",)), ("conv5_2_only", ("conv5_2",)), ("default", nv.DEFAULT_X3_LAYERS_VGG16)):
    m.winograd_x3_layers = layers
    p, c, d = m(image_data=img)
    np.save("gpurun_out/props_%s.npy" % name, p.cpu().numpy())
    sc = m.context(0).tensor(2)
    np.save("gpurun_out/scores_%s.npy" % name, sc.cpu().numpy())
