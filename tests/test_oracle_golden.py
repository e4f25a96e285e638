"""Pins oracle/frcnn_oracle.py against the golden vectors captured from the imported reference
(oracle/make_golden.py).  CPU only."""
import hashlib
import os

import numpy as np
import pytest
import torch
import torch as t

from fasterrcnn_amd import synthetic
from oracle import frcnn_oracle as O


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def flat(d):
    rows = [np.hstack([np.full((v.shape[0], 1), float(c)), v]) for c, v in sorted(d.items()) if v.shape[0]]
    return np.vstack(rows) if rows else np.zeros((0, 6))


@pytest.fixture(scope="module")
def small_ops(golden_dir):
    return np.load(os.path.join(golden_dir, "small_ops.npz"))


@pytest.mark.parametrize("tag", ["vgg", "resnet", "small", "odd"])
def test_anchor_maps_bit_exact(small_ops, tag):
    s = small_ops["anchors_%s_shape" % tag]
    am, vm = O.generate_anchor_maps(tuple(s[:3]), tuple(s[3:]), 16)
    assert [sha(am), sha(vm)] == list(small_ops["anchors_%s_sha" % tag])
    assert int(vm.sum()) == int(small_ops["anchors_%s_nvalid" % tag])
    if tag in ("small", "odd"):
        assert np.array_equal(am, small_ops["anchors_%s_map" % tag])
        assert np.array_equal(vm, small_ops["anchors_%s_valid" % tag])


def test_survey_probe_hashes(small_ops):
    # hashes measured on the reference at survey time
    assert list(small_ops["anchors_vgg_sha"]) == ["bc0ded16d198f63e", "0e397daf862a5ca2"]
    assert list(small_ops["anchors_resnet_sha"]) == ["0cb76c8d833871ed", "b2cbea43c58fb086"]
    assert int(small_ops["anchors_vgg_nvalid"]) == 8044


def test_map_known_answer_and_stream(small_ops):
    gts = small_ops["map_stream_gt"]
    preds = small_ops["map_stream_pred"]
    acc = O.MeanAveragePrecision()
    for i in range(int(gts[:, 0].max()) + 1):
        g = [(int(r[1]), r[2:6].astype(np.float32)) for r in gts[gts[:, 0] == i]]
        p = {c: preds[(preds[:, 0] == i) & (preds[:, 1] == c)][:, 2:7] for c in range(1, 21)}
        acc.add_image_results(p, g)
    assert acc.mean_average_precision() == float(small_ops["map_stream_value"])
    assert float(small_ops["map_known_answer"]) == 1.0


def test_nms_edge_cases():
    assert O.nms(np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), 0.7).shape == (0,)
    b = np.array([[0, 0, 10, 10], [0, 0, 10, 10], [20, 20, 30, 30], [0, 0, 10, 9.9]], np.float32)
    s = np.array([0.5, 0.5, 0.9, 0.4], np.float32)
    # ties keep input order (stable); identical boxes suppress each other; IoU 0.99 > 0.7
    assert O.nms(b, s, 0.7).tolist() == [2, 0]
    # degenerate zero-area boxes: 0/0 -> nan, never > thr -> kept
    z = np.zeros((3, 4), np.float32)
    assert O.nms(z, np.array([3, 2, 1], np.float32), 0.5).tolist() == [0, 1, 2]


def test_roi_pool_semantics():
    fm = np.arange(1 * 2 * 6 * 8, dtype=np.float32).reshape(1, 2, 6, 8)
    # roi covering columns 0..3, rows 0..1 at scale 1: (b, x1, y1, x2, y2)
    out = O.roi_pool(fm, np.array([[0, 0, 0, 3, 1]], np.float32), 2, 1.0)
    assert out.shape == (1, 2, 2, 2)
    assert out[0, 0].tolist() == [[1.0, 3.0], [9.0, 11.0]]
    # half-away-from-zero rounding: 2.5 -> 3 (numpy's rint would give 2)
    out = O.roi_pool(fm, np.array([[0, 2.5, 0, 2.5, 0]], np.float32), 1, 1.0)
    assert out[0, 0, 0, 0] == 3.0
    # roi entirely outside the map -> empty bins -> zeros
    out = O.roi_pool(fm, np.array([[0, 100, 100, 120, 120]], np.float32), 2, 1.0)
    assert not out.any()


def iou_matrix(a, b):
    tl = np.maximum(a[:, None, 0:2], b[None, :, 0:2])
    br = np.minimum(a[:, None, 2:4], b[None, :, 2:4])
    wh = np.clip(br - tl, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    aa = np.prod(a[:, 2:4] - a[:, 0:2], axis=1)
    ab = np.prod(b[:, 2:4] - b[:, 0:2], axis=1)
    return inter / (aa[:, None] + ab[None, :] - inter + 1e-12)


def check_against_golden(g, props, classes, deltas, detail, det):
    """
    The golden vectors were produced by the imported reference on the build container's CPU.  torch's
    CPU GEMMs are bit-reproducible on the same CPU model and thread count, so there the oracle must
    equal the reference BIT FOR BIT; on another host (different ISA path / reduction order) the dense
    layers differ in the last bits and the comparison falls back to the fp32 tolerances of the GPU
    tests (feature map 2e-5 relative, boxes 1e-3 px on IoU-matched rows).
    """
    fm = detail["feature_map"].numpy()[0]
    step = fm.shape[0] // 32                    # the fixture holds 32 of the 512 (VGG) / 1024 (ResNet) channels
    same_host = np.array_equal(fm[::step], g["feature_map_sample"])
    if same_host:
        assert np.array_equal(detail["sorted_idx"].astype(np.int32), g["sorted_idx"])
        assert np.array_equal(props.numpy(), g["proposals"])
        assert np.array_equal(classes.numpy(), g["classes"])
        assert np.array_equal(deltas.numpy(), g["box_deltas"])
        assert np.array_equal(flat(det), g["detections"])
        return True
    scale = float(np.abs(g["feature_map_sample"]).max())
    assert float(np.abs(fm[::step] - g["feature_map_sample"]).max()) <= 2e-5 * scale
    assert len(set(detail["sorted_idx"].tolist()) ^ set(g["sorted_idx"].tolist())) <= 8
    m = iou_matrix(g["proposals"].astype(np.float64), props.numpy().astype(np.float64))
    j = m.argmax(axis=1)
    ok = np.abs(props.numpy()[j] - g["proposals"]).max(axis=1) <= 1e-3
    assert ok.mean() >= 0.95
    assert np.abs(classes.numpy()[j[ok]] - g["classes"][ok]).max() <= 1e-4
    assert abs(len(flat(det)) - len(g["detections"])) <= max(3, 0.05 * len(g["detections"]))
    return False


@pytest.mark.parametrize("tag,allow_edge", [("224x320_s3", True), ("333x517_s5_noedge", False)])
def test_oracle_reproduces_reference_small(golden_dir, sd_cpu, tag, allow_edge):
    g = np.load(os.path.join(golden_dir, "vgg16_%s.npz" % tag))
    img = synthetic.image(int(g["seed"]), int(g["height"]), int(g["width"])).unsqueeze(0)
    detail = {}
    props, classes, deltas = O.forward(sd_cpu, img, allow_edge_proposals=allow_edge, detail=detail)
    det = O.detections(props.numpy(), classes.numpy(), deltas.numpy(), int(g["height"]), int(g["width"]),
                       float(g["score_threshold"]))
    assert sorted(det.keys()) == list(range(1, 21))
    check_against_golden(g, props, classes, deltas, detail, det)


def test_oracle_reproduces_reference_full_size(golden_dir, sd_cpu):
    g = np.load(os.path.join(golden_dir, "vgg16_600x1000_s0.npz"))
    img = synthetic.image(0).unsqueeze(0)
    assert tuple(img.shape) == (1, 3, 600, 1000)
    detail = {}
    props, classes, deltas = O.forward(sd_cpu, img, detail=detail)
    det = O.detections(props.numpy(), classes.numpy(), deltas.numpy(), 600, 1000, 0.05)
    if check_against_golden(g, props, classes, deltas, detail, det):
        assert sha(detail["scores"].numpy()) == str(g["scores_sha"])


@pytest.mark.parametrize("name,arch", [("resnet50_250x333_s7", "ResNet50"), ("resnet101_224x320_s3", "ResNet101"),
                                       ("resnet152_250x333_s7", "ResNet152"),
                                       ("resnet50_600x1000_s0", "ResNet50")])
def test_oracle_reproduces_reference_resnet(golden_dir, name, arch):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    sd = synthetic.resnet_state_dict(1234, arch)
    img = synthetic.image_rgb(int(g["seed"]), int(g["height"]), int(g["width"])).unsqueeze(0)
    detail = {}
    props, classes, deltas = O.forward(sd, img, detail=detail)
    assert detail["feature_map"].shape[1] == 1024
    assert tuple(detail["feature_map"].shape[2:]) == (-(-int(g["height"]) // 16), -(-int(g["width"]) // 16))   # ceil
    det = O.detections(props.numpy(), classes.numpy(), deltas.numpy(), int(g["height"]), int(g["width"]),
                       float(g["score_threshold"]))
    check_against_golden(g, props, classes, deltas, detail, det)


@pytest.mark.parametrize("name", ["known", "one", "five", "many"])
def test_rpn_ground_truth_map_matches_reference(small_ops, name):
    am, vm = O.generate_anchor_maps((3, 600, 1000), (512, 37, 62), 16)
    rmap, obj, bg = O.generate_rpn_map(am, vm, small_ops["rpn_gt_%s" % name])
    assert sha(rmap) == str(small_ops["rpn_map_sha_%s" % name])
    assert np.array_equal(obj.astype(np.int32), small_ops["rpn_obj_%s" % name])
    assert len(bg) == int(small_ops["rpn_nbg_%s" % name])
    assert np.array_equal(bg[:64].astype(np.int32), small_ops["rpn_bg_head_%s" % name])
    if name == "known":        # probe values measured on the reference
        assert len(obj) == 10 and len(bg) == 6847 and obj[0].tolist() == [9, 7, 2]


@pytest.mark.parametrize("h,w,oh,ow", [(375, 500, 600, 800), (500, 375, 800, 600), (1200, 1600, 600, 800),
                                       (333, 517, 333, 517), (100, 37, 163, 60), (720, 1280, 600, 1066)])
def test_pil_bilinear_resample_restatement_is_bit_exact(h, w, oh, ow):
    """datasets/image.py:96 calls PIL's resize; PIL IS installed, so the restatement is pinned on the real thing."""
    from PIL import Image
    rng = np.random.RandomState(h + w)
    img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    ref = np.array(Image.fromarray(img, mode="RGB").resize((ow, oh), resample=Image.BILINEAR))
    assert np.array_equal(O.pil_resize_bilinear(img, oh, ow), ref)


def test_preprocess_restatement_matches_reference_sequence():
    """image.py:43-57 on PIL-resized pixels: float32 scale, mean/std, BGR flip, CHW."""
    from PIL import Image
    rng = np.random.RandomState(3)
    img = rng.randint(0, 256, (375, 500, 3)).astype(np.uint8)
    pil = Image.fromarray(img, mode="RGB").resize((800, 600), resample=Image.BILINEAR)
    x = np.array(pil).astype(np.float32)[:, :, ::-1].copy()
    for c, m in enumerate([103.939, 116.779, 123.680]):
        x[:, :, c] = (x[:, :, c] * 1.0 - m) / 1
    out = O.preprocess_image(img, True, 1.0, [103.939, 116.779, 123.680], [1, 1, 1], 600)
    assert out.dtype == np.float32 and out.shape == (3, 600, 800)
    assert np.array_equal(out, x.transpose(2, 0, 1))


# ---- training step: oracle/train_oracle.py vs the fixtures captured from the reference's train_step ----
@pytest.mark.parametrize("backbone", ["vgg16", "resnet50"])
def test_train_oracle_reproduces_reference_step(golden_dir, sd_cpu, backbone):
    import random
    from oracle import train_oracle as TO
    g = np.load(os.path.join(golden_dir, "train_%s_352x480_s4.npz" % backbone))
    seed, h, w = int(g["seed"]), int(g["height"]), int(g["width"])
    if backbone == "resnet50":
        sd_cpu = synthetic.resnet_state_dict(1234, "ResNet50")
        img = synthetic.image_rgb(seed, h, w).unsqueeze(0)
        fshape = (1024, -(-h // 16), -(-w // 16))
    else:
        img = synthetic.image(seed, h, w).unsqueeze(0)
        fshape = (512, h // 16, w // 16)
    gts = synthetic.ground_truth(seed, h, w)
    am, vm = O.generate_anchor_maps((3, h, w), fshape, 16)
    gc = np.stack([k for _, k in gts])
    gcls = np.array([c for c, _ in gts])
    rmap, obj, bg = O.generate_rpn_map(am, vm, gc)
    random.seed(int(g["rng_seed"])); t.manual_seed(int(g["rng_seed"]))
    detail = {}
    losses, grads, new_sd, bufs = TO.train_step(sd_cpu, img, am, vm, t.from_numpy(rmap).unsqueeze(0), obj, bg, gc, gcls, 21,
                                                float(g["lr"]), float(g["momentum"]), float(g["weight_decay"]), None, detail=detail)
    # host RNG use is platform independent: the sampled anchors must be the reference's
    assert np.array_equal(detail["rpn_sample_flat"], g["s0_rpn_sample_flat"])
    want = g["s0_losses"]
    got = np.array([losses[n] for n in ("rpn_class", "rpn_regression", "detector_class", "detector_regression", "total")])
    if detail["rpn_proposals"].shape[0] == int(g["s0_n_rpn_proposals"]) and \
            np.array_equal(detail["proposal_sample_indices"].astype(np.int32), g["s0_proposal_sample_indices"]):
        # same discrete selections as on the machine that wrote the fixture: float results agree to rounding
        assert np.all(np.abs(got - want) <= 1e-5 * np.abs(want)), (got, want)
        keys = [str(k) for k in g["train_keys"]]
        for k in keys:
            gk = grads[k].numpy().reshape(-1).astype(np.float64)
            pos = np.unique(np.linspace(0, gk.shape[0] - 1, min(2048, gk.shape[0])).astype(np.int64))
            ws = g["s0_gsample/" + k].astype(np.float64)
            assert np.abs(gk[pos] - ws).max() <= 1e-4 * max(np.abs(ws).max(), 1e-12), k
        for k in sd_cpu:
            if k not in keys:
                assert t.equal(new_sd[k], sd_cpu[k]), k            # frozen blocks and all biases untouched
    else:
        # a different CPU rounds the 13-layer forward differently and an RPN rank may flip: only coarse agreement
        assert np.all(np.abs(got - want) <= 0.2 * np.abs(want) + 0.05), (got, want)


def test_train_oracle_roi_pool_backward_routes_to_first_maximum():
    from oracle import train_oracle as TO
    fm = t.zeros((1, 2, 8, 8))
    fm[0, 0, 2, 3] = 5.0                    # unique maximum of the whole map for channel 0; channel 1 all ties (zeros)
    x = fm.clone().requires_grad_(True)
    out = TO.roi_pool_autograd(x, t.tensor([[0.0, 0.0, 127.0, 127.0]]))
    assert t.equal(out.detach(), t.from_numpy(O.roi_pool(fm.numpy(), np.array([[0, 0, 0, 127, 127]], dtype=np.float32), 7, 1 / 16.0)))
    out.sum().backward()
    gx = x.grad[0]
    assert float(gx[0, 2, 3]) == float((out[0, 0] == 5.0).sum())      # every bin whose window holds (2,3) routes there
    assert float(gx[1].sum()) == 49.0 and float(gx[1, 0, 0]) >= 1.0   # ties -> first cell of each window
