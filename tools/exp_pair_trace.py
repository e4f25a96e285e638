"""tools/exp_pair_trace.py SET NSLOTS -- runs bench.py's loop with pair set SET (tools/exp_pair.py) for a kernel trace:
rocprofv3 --kernel-trace --stats -d DIR -- python tools/exp_pair_trace.py conv4 1"""
import sys
import time
import torch
sys.path.insert(0, ".")
from fasterrcnn_amd import _native as nv, synthetic
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models.vgg16 import VGG16Backbone
from tools.exp_pair import SETS

nv.require_gpu()
dev = torch.device("cuda", 0)
model = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
model.load_state_dict(synthetic.vgg16_state_dict(1234), strict=True)
model = model.cuda(dev).eval()
pool = [synthetic.image(s).unsqueeze(0).to(dev) for s in range(8)]
model.inflight_pair_layers = SETS[sys.argv[1]]
model.alone_pair_layers = SETS[sys.argv[1]]
nslots = int(sys.argv[2])
pending = []
t_end = time.perf_counter() + 4.0
i = 0
while time.perf_counter() < t_end:
    if len(pending) == nslots:
        pending.pop(0).result()
    pending.append(model.predict_async(pool[i % len(pool)], 0.05, slot=0 if nslots == 1 else 1 + (i % nslots)))
    i += 1
while pending:
    pending.pop(0).result()
print("images", i)
