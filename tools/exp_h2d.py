"""tools/exp_h2d.py -- where the host -> device leg of bench.py (h2d_preprocess_images_per_sec) spends its time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402

from fasterrcnn_amd import synthetic  # noqa: E402
from fasterrcnn_amd.datasets import image as I  # noqa: E402
from fasterrcnn_amd.evaluate import HostFeeder  # noqa: E402
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel  # noqa: E402
from fasterrcnn_amd.models.vgg16 import VGG16Backbone  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    model = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
    model.load_state_dict(synthetic.vgg16_state_dict(1234), strict=True)
    model = model.cuda(dev).eval()
    params = model.backbone.image_preprocessing_params
    frames = [synthetic.image_u8(s).pin_memory() for s in range(8)]
    dframes = [f.cuda() for f in frames]
    pool = [synthetic.image(s).unsqueeze(0).to(dev) for s in range(8)]
    n = 200

    def timed(fn, reps=3):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best / n * 1e6

    def prep_only():
        for i in range(n):
            I.preprocess_image(dframes[i % 8], params, 600, False)
    print("preprocess_image on a device-resident frame: %.1f us per image (host + device, back to back)" % timed(prep_only))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    I.preprocess_image(dframes[0], params, 600, False)
    torch.cuda.synchronize()
    ev0.record()
    for i in range(20):
        I.preprocess_image(dframes[i % 8], params, 600, False)
    ev1.record()
    torch.cuda.synchronize()
    print("   device time by events: %.1f us per image" % (ev0.elapsed_time(ev1) / 20 * 1e3))

    def host_only():
        t0 = time.perf_counter()
        for i in range(n):
            I.preprocess_image(dframes[i % 8], params, 600, False)
        return (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    print("   host time of the call (enqueue only): %.1f us" % host_only())
    torch.cuda.synchronize()

    nslots = 3

    def loop(submit, frames_):
        def fn():
            pend = []
            for i in range(n):
                if len(pend) == nslots:
                    pend.pop(0).result()
                pend.append(submit(frames_[i % 8], 0.05, 1 + i % nslots))
            while pend:
                pend.pop(0).result()
        return fn
    base = loop(lambda f, thr, slot: model.predict_async(f, thr, slot=slot), pool)
    for _ in range(2):
        base()
    print("resident loop: %.1f us per image" % timed(base))
    feeder = HostFeeder(model)
    a = loop(feeder.submit, frames)
    a()
    print("HostFeeder.submit (u8 H2D + preprocess + predict): %.1f us per image" % timed(a))
    host_f32 = [p[0].cpu().pin_memory() for p in pool]
    b = loop(feeder.submit_preprocessed, host_f32)
    b()
    print("HostFeeder.submit_preprocessed (float32 H2D + predict): %.1f us per image" % timed(b))

    # preprocess on the feeder stream but from device-resident frames (no H2D)
    def submit_dev(f, thr, slot):
        st = feeder._streams[slot % len(feeder._streams)]
        with torch.cuda.stream(st):
            img, _, _ = I.preprocess_image(f, params, 600, False)
            return model.predict_async(img.unsqueeze(0), thr, slot=slot)
    c = loop(submit_dev, dframes)
    c()
    print("preprocess (device-resident u8) + predict: %.1f us per image" % timed(c))


if __name__ == "__main__":
    main()
