"""tools/exp_pair.py -- images/sec of the VGG-16 forward (bench.py's loop: predict_async, 3 images in flight / one at a time) with different
sets of one-launch f32x3 layers in the TWO-PASS form (csrc/wino_x3p.hip).  Sets are alternated A B A B so that box drift shows.
Also checks that detections are identical to the empty set's (the form is bit-identical)."""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from fasterrcnn_amd import _native as nv, synthetic
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models.vgg16 import VGG16Backbone

SETS = {
    "none": (),
    "conv4": ("conv4_1", "conv4_2", "conv4_3"),
    "conv4+5": ("conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3", "rpn_trunk"),
    "conv3+4": ("conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3"),
    "all": ("conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3", "rpn_trunk"),
}


def main():
    nv.require_gpu()
    dev = torch.device("cuda", 0)
    model = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
    model.load_state_dict(synthetic.vgg16_state_dict(1234), strict=True)
    model = model.cuda(dev).eval()
    pool = [synthetic.image(s).unsqueeze(0).to(dev) for s in range(8)]

    def run(n_steps, nslots):
        pending = []
        for i in range(n_steps):
            if len(pending) == nslots:
                pending.pop(0).result()
            pending.append(model.predict_async(pool[i % len(pool)], 0.05, slot=0 if nslots == 1 else 1 + (i % nslots)))
        last = None
        while pending:
            last = pending.pop(0).result()
        return last

    def rate(nslots, steps):
        run(3 * nslots + 10, nslots)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps, nslots)
        torch.cuda.synchronize()
        return steps / (time.perf_counter() - t0)

    t_end = time.perf_counter() + 2.0
    while time.perf_counter() < t_end:
        run(3, 3)
    names = [a for a in sys.argv[1:] if a in SETS] or list(SETS)
    ref = None
    res = {n: ([], []) for n in names}
    for rep in range(3):
        for n in names:
            model.inflight_pair_layers = SETS[n]
            model.alone_pair_layers = SETS[n]
            det = model.predict(pool[0], 0.05)
            key = np.concatenate([det[c].ravel() for c in sorted(det)])
            if ref is None:
                ref = key
            assert key.shape == ref.shape and np.array_equal(key, ref), "detections differ with pair set %s" % n
            res[n][0].append(rate(3, 400))
            res[n][1].append(rate(1, 150))
    for n in names:
        print("pair layers %-8s | 3 images in flight: %s images/sec | one at a time: %s" % (
            n, " ".join("%.1f" % v for v in res[n][0]), " ".join("%.1f" % v for v in res[n][1])))


if __name__ == "__main__":
    main()
