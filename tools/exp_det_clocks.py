import sys; sys.path.insert(0,'.')
import torch
from fasterrcnn_amd import synthetic
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models.vgg16 import VGG16Backbone
m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
m.load_state_dict(synthetic.vgg16_state_dict(1234), strict=True)
m = m.cuda().eval()
img = synthetic.image(0).unsqueeze(0).cuda()
for _ in range(3):
    d = m.predict(img, 0.05)
torch.cuda.synchronize()
