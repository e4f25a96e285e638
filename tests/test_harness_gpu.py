"""
The evaluate / predict harness (SURVEY.md section 8 row a-H; reference __main__.py:62-96 evaluate(), :226-240
predict / predict_one) and the multi-GPU code path on ONE GPU:

  * BASELINE configs[2]: ResNet-50, 8 batch-1 images in flight through evaluate_stream == 8 sequential predict() calls,
    and the 600x1000 image among them reproduces the reference's golden detections;
  * VGG-16 image stream with planted ground truth through evaluate() == the sequential reference loop's mAP;
  * the mAP gather over RCCL (backend "nccl") in a one-rank group: merged_calculator(force_gather=True);
  * bench.py under torch.distributed.run with --force-dist: init_process_group("nccl"), barriers, the max-over-ranks
    all-reduce of the burst time and the record all-gather all execute on RCCL;
  * predict_one on a PNG file == oracle preprocessing + oracle predict.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from fasterrcnn_amd import evaluate as E
from fasterrcnn_amd import synthetic
from fasterrcnn_amd.datasets.training_sample import Box
from fasterrcnn_amd.statistics import PrecisionRecallCurveCalculator
from oracle import frcnn_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def iou_matrix(a, b):
    tl = np.maximum(a[:, None, 0:2], b[None, :, 0:2])
    br = np.minimum(a[:, None, 2:4], b[None, :, 2:4])
    wh = np.clip(br - tl, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    aa = np.prod(a[:, 2:4] - a[:, 0:2], axis=1)
    ab = np.prod(b[:, 2:4] - b[:, 0:2], axis=1)
    return inter / (aa[:, None] + ab[None, :] - inter + 1e-12)


def planted_gt(seed, det, h, w):
    """Seeded random boxes plus up to 3 of the image's own top detections, jittered (mAP neither 0 nor 1)."""
    rng = np.random.RandomState(104729 * int(seed) + 1)
    boxes = [Box(c, str(c), k) for c, k in synthetic.ground_truth(seed, h, w)]
    rows = [(c, r) for c, v in det.items() for r in v[:2]]
    rows.sort(key=lambda cr: -cr[1][4])
    for c, r in rows[:3]:
        boxes.append(Box(int(c), str(c), (r[:4] + rng.randn(4) * 4.0).astype(np.float32)))
    return boxes


class Sample:
    def __init__(self, image_data, gt_boxes):
        self.image_data, self.gt_boxes = image_data, gt_boxes


# ---------------------------------------------------------------------------------------------------------------
def test_resnet50_eight_in_flight_equals_sequential_and_golden(golden_dir):
    """BASELINE configs[2] ("ResNet-50, batch=8"): the reference asserts batch 1 (faster_rcnn.py:108), so the 8 images are 8
    independent batch-1 forwards in flight on 8 streams; every one must equal its sequential predict()."""
    from fasterrcnn_amd.models import resnet
    from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
    model = FasterRCNNModel(num_classes=21, backbone=resnet.ResNetBackbone(resnet.Architecture.ResNet50))
    model.load_state_dict(synthetic.resnet_state_dict(1234, "ResNet50"), strict=True)
    model = model.cuda().eval()
    model.inflight_conv_blocks_target = 0           # same split-K partition as the sequential slot -> bitwise comparable
    model.inflight_winograd_tile_rows = 0
    g = np.load(os.path.join(golden_dir, "resnet50_600x1000_s0.npz"))
    seeds = [int(g["seed"])] + list(range(31, 38))
    images = [synthetic.image_rgb(s).unsqueeze(0).cuda() for s in seeds]
    sequential = [model.predict(image_data=im, score_threshold=0.05) for im in images]
    got = {}
    samples = [(s, im, None) for s, im in zip(seeds, images)]
    E.evaluate_stream(model, samples, score_threshold=0.05, inflight=8, on_result=lambda i, d: got.__setitem__(i, d))
    assert sorted(got) == sorted(seeds)
    for s, ref in zip(seeds, sequential):
        for c in ref:
            assert np.array_equal(got[s][c], ref[c]), (s, c)
    # and a second pass that re-uses the 8 slots gives the same dicts again
    got2 = {}
    E.evaluate_stream(model, samples + samples, score_threshold=0.05, inflight=8, on_result=lambda i, d: got2.__setitem__(i, d))
    for s in seeds:
        for c in got[s]:
            assert np.array_equal(got2[s][c], got[s][c])
    # the golden image: every one of the reference's 232 detections reproduced (tests/test_resnet_gpu.py holds the same gate)
    ref = g["detections"]
    n_ok = 0
    for c in range(1, 21):
        r = ref[ref[:, 0] == c][:, 1:]
        d = got[seeds[0]][c]
        if len(r) and len(d):
            j = iou_matrix(r[:, :4], d[:, :4]).argmax(axis=1)
            n_ok += int(((np.abs(d[j, :4] - r[:, :4]).max(axis=1) <= 1e-3) & (np.abs(d[j, 4] - r[:, 4]) <= 2e-4)).sum())
    print("ResNet-50 600x1000 via evaluate_stream (8 in flight): %d/%d reference detections" % (n_ok, len(ref)))
    assert n_ok == len(ref)


def test_evaluate_stream_map_equals_sequential_reference_loop(gpu_model):
    """evaluate() (in-flight slots + record merge) == the reference's loop: predict -> add_image_results -> mAP."""
    seeds = list(range(40, 46))
    hw = [(600, 1000), (600, 1000), (224, 320), (333, 517), (600, 1000), (352, 480)]      # ragged stream: slots see several shapes
    images = [synthetic.image(s, h, w) for s, (h, w) in zip(seeds, hw)]
    calc = PrecisionRecallCurveCalculator()
    samples = []
    for s, im, (h, w) in zip(seeds, images, hw):
        det = gpu_model.predict(image_data=im.unsqueeze(0).cuda(), score_threshold=0.05)
        gt = planted_gt(s, det, h, w)
        calc.add_image_results(scored_boxes_by_class_index=det, gt_boxes=gt)
        samples.append(Sample(im.numpy(), gt))                     # numpy (3,H,W), as the reference's dataset yields it
    want = 100.0 * calc.compute_mean_average_precision()
    saved = gpu_model.inflight_conv_blocks_target, gpu_model.inflight_winograd_tile_rows, gpu_model.inflight_winograd_x3f_layers
    gpu_model.inflight_conv_blocks_target = gpu_model.inflight_winograd_tile_rows = 0
    gpu_model.inflight_winograd_x3f_layers = gpu_model.alone_winograd_x3f_layers        # the sequential loop's (slot 0's) arithmetic in the in-flight slots: the two mAPs are then EQUAL
    try:
        got = E.evaluate(gpu_model, samples, inflight=4)
        got_limited = E.evaluate(gpu_model, samples, num_samples=3, inflight=2)
    finally:
        gpu_model.inflight_conv_blocks_target, gpu_model.inflight_winograd_tile_rows, gpu_model.inflight_winograd_x3f_layers = saved
    print("mAP sequential %.6f%%, evaluate() %.6f%%" % (want, got))
    assert 0.0 < want < 100.0
    assert got == want
    calc3 = PrecisionRecallCurveCalculator()
    for smp in samples[:3]:
        calc3.add_image_results(gpu_model.predict(torch.from_numpy(smp.image_data).unsqueeze(0).cuda(), 0.05), smp.gt_boxes)
    assert got_limited == 100.0 * calc3.compute_mean_average_precision()


def test_background_uploader_keeps_order_values_and_errors():
    """evaluate()'s upload leg (E.BackgroundUploader: the reference's `t.from_numpy(image).unsqueeze(0).cuda()`, __main__.py:78-86, from a
    worker thread a few images ahead): the images arrive in order, bit for bit, entries without an image pass through, an exception of
    the sample iterator reaches the consumer, and a consumer that stops early leaves no blocked worker behind."""
    import threading
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(3)
    arrays = [rng.rand(3, 40 + i, 64).astype(np.float32) for i in range(9)]

    def items(fail_at=None):
        for i, a in enumerate(arrays):
            if i == fail_at:
                raise ValueError("sample %d" % i)
            yield i, (None if i % 4 == 3 else (torch.from_numpy(a) if i % 2 else a)), ("gt", i)
    got = list(E.BackgroundUploader(dev, depth=2).iterate(items()))
    assert [g[0] for g in got] == list(range(9)) and [g[2] for g in got] == [("gt", i) for i in range(9)]
    for i, image, _ in got:
        if i % 4 == 3:
            assert image is None
        else:
            assert image.is_cuda and tuple(image.shape) == (1,) + arrays[i].shape
            torch.cuda.current_stream().synchronize()
            assert np.array_equal(image[0].cpu().numpy(), arrays[i])
    with pytest.raises(ValueError, match="sample 5"):
        list(E.BackgroundUploader(dev, depth=2).iterate(items(fail_at=5)))
    gen = E.BackgroundUploader(dev, depth=1).iterate(items())
    next(gen)
    gen.close()                                                    # the consumer walks away: the worker must not stay blocked on its queue
    for _ in range(50):
        if not any(th.name == "frcnn-upload" and th.is_alive() for th in threading.enumerate()):
            break
        import time
        time.sleep(0.1)
    assert not any(th.name == "frcnn-upload" and th.is_alive() for th in threading.enumerate())


def test_map_gather_over_rccl_one_rank_group(gpu_model):
    """merged_calculator's exchange (sizes + padded payload all-gather per array) on backend "nccl" = RCCL, in a child process
    so that the process group does not outlive the test: merged == local, bit for bit."""
    code = r'''
import os, sys, json
sys.path.insert(0, %r)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch, torch.distributed as dist
from fasterrcnn_amd import evaluate as E
from fasterrcnn_amd.datasets.training_sample import Box
rng = np.random.RandomState(5)
rec = E.ImageRecords()
for img in range(7):
    det = {c: np.zeros((0, 5)) for c in range(1, 21)}
    gts = []
    for c in rng.choice(np.arange(1, 21), 4, replace=False):
        n = int(rng.randint(1, 6))
        y1, x1 = rng.uniform(0, 300, n), rng.uniform(0, 600, n)
        rows = np.stack([y1, x1, y1 + rng.uniform(30, 200, n), x1 + rng.uniform(30, 300, n), np.sort(rng.uniform(0.05, 1, n))[::-1]], 1)
        det[int(c)] = rows
        gts.append(Box(int(c), "x", (rows[0, :4] + rng.randn(4) * 6).astype(np.float32)))
    rec.add(img, det, gts)
local = E.merged_calculator(rec).compute_mean_average_precision()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
try:
    assert dist.get_backend() == "nccl"
    merged = E.merged_calculator(rec, force_gather=True).compute_mean_average_precision()
    x = torch.arange(6, dtype=torch.float64).reshape(3, 2)
    y = E._all_gather_rows(x, torch.device("cuda", 0))
    assert torch.equal(x, y)
    z = E._all_gather_rows(torch.zeros((0, 4), dtype=torch.float64), torch.device("cuda", 0))
    assert tuple(z.shape) == (0, 4)
finally:
    dist.destroy_process_group()
print("RESULT " + json.dumps({"local": local, "merged": merged}), flush=True)
''' % ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]        # RCCL prints its own banner lines
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0][len("RESULT "):])
    assert r["local"] == r["merged"] and 0.0 < r["local"] <= 1.0


def test_bench_multi_rank_branch_on_rccl():
    """bench.py launched exactly as the driver launches N > 1 (torch.distributed.run, one rank per GPU) with ONE rank and
    --force-dist: process group on RCCL, barriers around the timed bursts, all-reduce(MAX) of the elapsed time, record gather."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1",
           "--force-dist", "--inflight", "4", "--no-cpu-baseline", "--no-secondary", "--no-extra-legs", "--ramp-seconds", "0.2",
           "--min-timed-seconds", "0.05", "--roofline-images", "2", "--map-images", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                             # ONE JSON line, from rank 0
    r = json.loads(lines[0])
    assert r["process_group"] == "nccl x1" and r["n_gpus"] == 1 and r["steps"] == 4 and r["scaling"] == "weak"
    assert r["value"] > 0 and abs(r["value"] - 4 / (r["ms_per_step"] * 4 / 1e3)) <= 1e-2 * r["value"]
    assert r["map_at_0.5"] is not None and r["timed_bursts"]["count"] >= 1


def test_predict_one_on_png_file(gpu_model, sd_cpu, tmp_path):
    """__main__.py:237-240: load_image(min_dimension_pixels=600) -> predict(score_threshold=0.7), against the oracle's
    preprocessing (== PIL, tests/test_kernels_gpu.py) and the oracle's predict on the same pixels."""
    from PIL import Image
    rng = np.random.RandomState(3)
    low = rng.randint(0, 256, (12, 16, 3)).astype(np.uint8)
    rgb = np.array(Image.fromarray(low, mode="RGB").resize((500, 375), resample=Image.BICUBIC))    # a smooth 375x500 "photo"
    path = str(tmp_path / "image.png")
    Image.fromarray(rgb, mode="RGB").save(path)
    det, pil, scale = E.predict_one(gpu_model, path, score_threshold=0.3)
    assert scale == 600 / 375 and pil.size == (800, 600)
    data = O.preprocess_image(rgb, True, 1.0, [103.939, 116.779, 123.680], [1, 1, 1], 600, False)
    assert data.shape == (3, 600, 800)
    ref = O.predict(sd_cpu, torch.from_numpy(data).unsqueeze(0), 0.3)
    n_ref = sum(len(v) for v in ref.values())
    n_ok = 0
    for c in ref:
        if len(ref[c]) and len(det[c]):
            j = iou_matrix(ref[c][:, :4], det[c][:, :4]).argmax(axis=1)
            n_ok += int(((np.abs(det[c][j, :4] - ref[c][:, :4]).max(axis=1) <= 1e-3) & (np.abs(det[c][j, 4] - ref[c][:, 4]) <= 1e-4)).sum())
    n_ours = sum(len(v) for v in det.values())
    print("predict_one: %d detections, %d/%d of the oracle's reproduced" % (n_ours, n_ok, n_ref))
    assert n_ref > 0 and n_ok >= 0.985 * n_ref and n_ours == n_ref     # the held-out floor (one row of 70); no extra, no missing row
    # the module-level predict() (numpy (3,H,W) in, as __main__.py:226-228) at the reference's default threshold
    d7 = E.predict(gpu_model, data)
    assert sorted(d7) == list(range(1, 21)) and all((v[:, 4] > 0.7).all() for v in d7.values())
    # datasets/image.py load_image keeps the reference's 4-tuple contract
    from fasterrcnn_amd.datasets import image as I
    arr, img_obj, sf, shape = I.load_image(path, gpu_model.backbone.image_preprocessing_params, min_dimension_pixels=600)
    assert isinstance(arr, np.ndarray) and arr.dtype == np.float32 and np.array_equal(arr, data)
    assert shape == (3, 375, 500) and sf == scale and img_obj.size == (800, 600)


@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs two GPUs in ONE process")
def test_model_moves_to_second_device_in_one_process():
    """The > 64 KB dynamic-LDS limit of the MFMA kernels is a PER-DEVICE function attribute (csrc/common.h: FRCNN_MAX_LDS_ONCE):
    the same process must be able to run the model on cuda:0 and then on cuda:1 (VERDICT r2 #7d).  Skipped on one-GPU boxes."""
    from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
    from fasterrcnn_amd.models.vgg16 import VGG16Backbone
    sd = synthetic.vgg16_state_dict(1234)
    model = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
    model.load_state_dict(sd, strict=True)
    img = synthetic.image(3, 224, 320).unsqueeze(0)
    out = []
    for d in (0, 1):
        dev = torch.device("cuda", d)
        model = model.cuda(dev).eval()
        with torch.cuda.device(dev):
            out.append(model.predict(image_data=img.to(dev), score_threshold=0.05))
    for c in out[0]:
        assert np.array_equal(out[0][c], out[1][c])


def test_host_feeder_equals_upload_then_predict(gpu_model, sd_cpu):
    """The end-to-end leg of bench.py (`h2d_preprocess_images_per_sec`): pinned host frame -> async H2D -> frcnn_preprocess
    -> predict_async on per-slot streams == the synchronous `preprocess_image(...)` + `predict` of the same frame, bit for bit, for
    several frames in flight and re-used slots; the preprocessed tensor == the oracle's PIL-exact restatement (datasets/image.py:59-101);
    `submit_preprocessed` (the reference's `t.from_numpy(image).cuda()`, __main__.py:80) == predict on the device-resident tensor."""
    from fasterrcnn_amd.datasets import image as I
    params = gpu_model.backbone.image_preprocessing_params
    frames = [synthetic.image_u8(70 + i) for i in range(5)]
    feeder = E.HostFeeder(gpu_model)
    base = []
    for f in frames:
        img, scale, shape = I.preprocess_image(f.numpy(), params, 600, False)
        assert tuple(img.shape) == (3, 600, 1000) and abs(scale - 1.6) < 1e-12 and shape == (3, 375, 625)
        base.append(gpu_model.predict(image_data=img.unsqueeze(0), score_threshold=0.05))
    ref = O.preprocess_image(frames[0].numpy(), params.channel_order.value == "BGR", params.scaling, params.means, params.stds, 600, False)
    img0, _, _ = I.preprocess_image(frames[0].numpy(), params, 600, False)
    assert np.array_equal(img0.cpu().numpy(), np.asarray(ref, dtype=np.float32))
    saved = gpu_model.inflight_conv_blocks_target, gpu_model.inflight_winograd_tile_rows, gpu_model.inflight_x6_gemm_tiles
    saved_x3f = gpu_model.inflight_winograd_x3f_layers
    gpu_model.inflight_conv_blocks_target, gpu_model.inflight_winograd_tile_rows, gpu_model.inflight_x6_gemm_tiles = 0, 0, 0
    gpu_model.inflight_winograd_x3f_layers = gpu_model.alone_winograd_x3f_layers        # (slot 0's table: the other one-launch layers of the in-flight slots differ from slot 0's by float32 rounding order)
    try:
        pinned = [f.pin_memory() for f in frames]
        pend = []
        got = {}
        for i, f in enumerate(pinned):                      # 2 slots, 5 frames: slots are re-used
            if len(pend) == 2:
                j, h = pend.pop(0)
                got[j] = h.result()
            pend.append((i, feeder.submit(f, 0.05, slot=1 + i % 2)))
        for j, h in pend:
            got[j] = h.result()
        for i in range(len(frames)):
            for c in base[i]:
                assert np.array_equal(base[i][c], got[i][c]), (i, c)
        # staged ahead of their predict (what bench.py's h2d_preprocess leg does): same bits
        staged = [feeder.stage(f) for f in pinned[:2]]
        hs = [feeder.submit_staged(st, 0.05, slot=1 + k) for k, st in enumerate(staged)]
        for k, hdl in enumerate(hs):
            r = hdl.result()
            for c in base[k]:
                assert np.array_equal(base[k][c], r[c]), (k, c)
        host_f32 = img0.cpu().pin_memory()
        res = feeder.submit_preprocessed(host_f32, 0.05, slot=1).result()
        for c in base[0]:
            assert np.array_equal(base[0][c], res[c]), c
    finally:
        gpu_model.inflight_conv_blocks_target, gpu_model.inflight_winograd_tile_rows, gpu_model.inflight_x6_gemm_tiles = saved
        gpu_model.inflight_winograd_x3f_layers = saved_x3f
    with pytest.raises(ValueError):
        feeder.submit(torch.zeros((3, 10, 10), dtype=torch.uint8), 0.05, slot=1)
