"""tools/soak_inflight.py -- determinism of the in-flight slots under load: the same images through predict_async with 1 .. 8 images in
flight, many times over; every result of an image must be bit-identical to its first result in an in-flight slot (the one-launch kernel's
LDS-DMA ring is ordered by s_waitcnt counts and barriers: a race would show up here as a differing bit under some interleaving), and slot 0
(since round 5 the same table as the in-flight slots) must agree with itself.  Development aid."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from fasterrcnn_amd import synthetic
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models.vgg16 import VGG16Backbone


def main():
    dev = torch.device("cuda:0")
    model = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
    model.load_state_dict(synthetic.vgg16_state_dict(0), strict=True)
    model = model.to(dev).eval()
    imgs = [synthetic.image(300 + i, 600, 1000).unsqueeze(0).to(dev) for i in range(5)]
    imgs += [synthetic.image(400 + i, h, w).unsqueeze(0).to(dev) for i, (h, w) in enumerate([(224, 320), (333, 517), (600, 901)])]
    first = {}
    n_checked = 0
    t0 = time.perf_counter()
    for rounds, n in ((3, 1), (6, 2), (8, 3), (8, 5), (8, 8), (6, 3)):
        for rep in range(rounds):
            pending = []
            order = np.random.RandomState(rep * 10 + n).permutation(len(imgs) * 3) % len(imgs)
            for k, ii in enumerate(order):
                if len(pending) == n:
                    j, p = pending.pop(0)
                    check(first, j, p.result(), n)
                    n_checked += 1
                pending.append((int(ii), model.predict_async(imgs[int(ii)], 0.05, slot=1 + (k % n))))
            for j, p in pending:
                check(first, j, p.result(), n)
                n_checked += 1
    base0 = [model.predict(image_data=im, score_threshold=0.05) for im in imgs]
    for im, b in zip(imgs, base0):
        again = model.predict(image_data=im, score_threshold=0.05)
        assert all(np.array_equal(b[c], again[c]) for c in b)
    print("soak: %d in-flight results of %d images, 1..8 in flight: every one bit-identical to the image's first; slot 0 repeatable; %.1f s"
          % (n_checked, len(imgs), time.perf_counter() - t0))


def check(first, j, res, n):
    if j not in first:
        first[j] = res
        return
    for c in res:
        assert np.array_equal(first[j][c], res[c]), "image %d differs with %d in flight (class %d)" % (j, n, c)


if __name__ == "__main__":
    main()
