"""tools/nms_clocks.py -- where nms_reduce_kernel (csrc/proposals.hip) spends its time on the workload's candidate lists.
Needs a library built with -DNMS_CLOCKS:  SRC=proposals tools/build_ablate.sh nmsclk -DNMS_CLOCKS ; FRCNN_LIB_PATH=build/libfrcnn_nmsclk.so
(the kernel leaves its clocks in the last three proposals: the forward's results are wrong in that build)."""
import sys
import torch
sys.path.insert(0, ".")
from fasterrcnn_amd import synthetic
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models.vgg16 import VGG16Backbone

model = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
model.load_state_dict(synthetic.vgg16_state_dict(1234), strict=True)
model = model.cuda().eval()
for seed in range(4):
    img = synthetic.image(seed).unsqueeze(0).cuda()
    for rep in range(3):
        p, _, _ = model(image_data=img)
    torch.cuda.synchronize()
    a, b, c = p[-1].tolist(), p[-2].tolist(), p[-3].tolist()
    tick = 0.01      # us per tick of s_memrealtime (100 MHz)
    print("image %d: %d candidates, %d kept, %d chunks resolved | kernel %.1f us: slot wait + 25 reads %.1f, ORs + the chunk's word %.1f, serial resolution %.1f, "
          "next word from the band %.1f, 25 DMA issues %.1f, surplus rows %.1f, between chunks %.1f, after the loop %.1f"
          % (seed, c[2], c[3], b[3], b[2] * tick, a[0] * tick, a[1] * tick, a[2] * tick, a[3] * tick, b[0] * tick, b[1] * tick, c[0] * tick, c[1] * tick))
