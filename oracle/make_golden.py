"""
oracle/make_golden.py -- TEST INFRASTRUCTURE ONLY; runs in the BUILD CONTAINER (needs /root/reference).

  python oracle/make_golden.py --calibrate     prints the layer multipliers frozen in
                                               fasterrcnn_amd/synthetic.py:CALIBRATION
  python oracle/make_golden.py --train         reference train_step vs oracle/train_oracle.py -> tests/golden/train_*.npz
  python oracle/make_golden.py                 (1) runs the imported REFERENCE (reference_shims.py)
                                               on the synthetic workload, (2) asserts that
                                               oracle/frcnn_oracle.py reproduces it, (3) writes the
                                               golden vectors to tests/golden/*.npz

The fixtures hold data only (seeds -> expected outputs); weights and images are regenerated from
the seeds by fasterrcnn_amd/synthetic.py on whichever machine runs the tests.
"""
import argparse
import hashlib
import os
import sys
import time

import numpy as np
import torch as t

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import frcnn_oracle as O          # noqa: E402
from oracle import reference_shims            # noqa: E402
from fasterrcnn_amd import synthetic          # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def build_reference_model(ref, sd, allow_edge=True, arch=None):
    if arch is None:
        backbone = ref.vgg16.VGG16Backbone(dropout_probability=0.0)
    else:
        backbone = ref.resnet.ResNetBackbone(architecture=getattr(ref.resnet.Architecture, arch))
    model = ref.faster_rcnn.FasterRCNNModel(num_classes=21, backbone=backbone, allow_edge_proposals=allow_edge)
    model.load_state_dict(sd, strict=True)
    model.eval()
    return model


def calibrate(ref):
    """Measures the five multipliers with the reference's own modules (seed 1234, image seed 0)."""
    ones = {k: 1.0 for k in synthetic.CALIBRATION}
    sd = synthetic.vgg16_state_dict(1234, calibration=ones)
    img = synthetic.image(0).unsqueeze(0)
    cal = {}
    with t.no_grad():
        model = build_reference_model(ref, sd)
        fm = model._stage1_feature_extractor(image_data=img)
        k = "_stage1_feature_extractor._block5_conv3.weight"
        cal[k] = 1.0 / float(fm.std())
        fm = fm * cal[k]
        rpn = model._stage2_region_proposal_network
        y = t.relu(rpn._rpn_conv1(fm))
        cal["_stage2_region_proposal_network._rpn_class.weight"] = 1.0 / float(rpn._rpn_class(y).std())
        cal["_stage2_region_proposal_network._rpn_boxes.weight"] = 0.3 / float(rpn._rpn_boxes(y).std())
        # detector statistics on the proposals of the calibrated RPN
        sd2 = synthetic.vgg16_state_dict(1234, calibration={**ones, **cal})
        model = build_reference_model(ref, sd2)
        detail = {}
        O.forward(sd2, img, detail=detail)
        fc2 = detail["fc2"]
        det = model._stage3_detector_network
        cal["_stage3_detector_network._classifier.weight"] = 3.0 / float(det._classifier(fc2).std())
        cal["_stage3_detector_network._regressor.weight"] = 1.0 / float(det._regressor(fc2).std())
    print("CALIBRATION = {")
    for k, v in cal.items():
        print('    "%s": %.9g,' % (k, v))
    print("}")


def calibrate_resnet(ref, arch="ResNet50"):
    """Head multipliers for the synthetic ResNet weights, measured with the reference's modules."""
    ones = {k: 1.0 for k in synthetic.RESNET_CALIBRATION}
    sd = synthetic.resnet_state_dict(1234, arch, calibration=ones)
    img = synthetic.image_rgb(0).unsqueeze(0)
    cal = {}
    with t.no_grad():
        model = build_reference_model(ref, sd, arch=arch)
        fm = model._stage1_feature_extractor(image_data=img)
        rpn = model._stage2_region_proposal_network
        y = t.relu(rpn._rpn_conv1(fm))
        k = "_stage2_region_proposal_network._rpn_conv1.weight"
        cal[k] = 1.0 / float(y.std())
        y = y * cal[k]
        cal["_stage2_region_proposal_network._rpn_class.weight"] = 1.0 / float(rpn._rpn_class(y).std())
        cal["_stage2_region_proposal_network._rpn_boxes.weight"] = 0.3 / float(rpn._rpn_boxes(y).std())
        sd2 = synthetic.resnet_state_dict(1234, arch, calibration={**ones, **cal})
        detail = {}
        O.forward(sd2, img, detail=detail)
        model = build_reference_model(ref, sd2, arch=arch)
        det = model._stage3_detector_network
        cal["_stage3_detector_network._classifier.weight"] = 3.0 / float(det._classifier(detail["fc2"]).std())
        cal["_stage3_detector_network._regressor.weight"] = 1.0 / float(det._regressor(detail["fc2"]).std())
    print("RESNET_CALIBRATION = {")
    for k, v in cal.items():
        print('    "%s": %.9g,' % (k, v))
    print("}")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def flatten_detections(d):
    rows = [np.hstack([np.full((v.shape[0], 1), float(c)), v]) for c, v in sorted(d.items()) if v.shape[0]]
    return np.vstack(rows) if rows else np.zeros((0, 6))


def assert_equal(name, a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape or not np.array_equal(a, b):
        diff = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) if a.shape == b.shape else float("nan")
        raise AssertionError("oracle != reference for %s (shapes %s %s, max|d| %g)" % (name, a.shape, b.shape, diff))
    print("  oracle == reference: %-28s %s %s" % (name, a.shape, a.dtype))


def run_case(ref, tag, sd, seed, height, width, allow_edge, score_threshold, arch=None):
    print("case %s: image seed %d, %dx%d, allow_edge=%s, backbone=%s" % (tag, seed, height, width, allow_edge, arch or "VGG16"))
    img = (synthetic.image_rgb if arch else synthetic.image)(seed, height, width).unsqueeze(0)
    model = build_reference_model(ref, sd, allow_edge, arch)
    t0 = time.time()
    with t.no_grad():
        ref_props, ref_classes, ref_deltas = model(image_data=img)
    t_fwd = time.time() - t0
    ref_det = model.predict(image_data=img, score_threshold=score_threshold)
    print("  reference forward %.2f s; %d proposals; %d detections over %d classes" % (
        t_fwd, ref_props.shape[0], sum(v.shape[0] for v in ref_det.values()),
        sum(1 for v in ref_det.values() if v.shape[0])))

    detail = {}
    o_props, o_classes, o_deltas = O.forward(sd, img, allow_edge_proposals=allow_edge, detail=detail)
    o_det = O.detections(o_props.numpy(), o_classes.numpy(), o_deltas.numpy(), height, width, score_threshold)
    assert_equal("proposals", o_props.numpy(), ref_props.numpy())
    assert_equal("classes", o_classes.numpy(), ref_classes.numpy())
    assert_equal("box_deltas", o_deltas.numpy(), ref_deltas.numpy())
    assert sorted(o_det.keys()) == sorted(ref_det.keys()) == list(range(1, 21))
    for c in ref_det:
        assert ref_det[c].dtype == np.float64 and ref_det[c].shape[1:] == (5,)
    assert_equal("detections", flatten_detections(o_det), flatten_detections(ref_det))

    # the reference's own intermediate tensors, for stage-level parity
    with t.no_grad():
        fm = model._stage1_feature_extractor(image_data=img)
        am, vm = ref.anchors.generate_anchor_maps(tuple(img.shape[1:]), model.backbone.compute_feature_map_shape(tuple(img.shape[1:])), 16)
        smap, dmap, _ = model._stage2_region_proposal_network(
            feature_map=fm, image_shape=tuple(img.shape[1:]), anchor_map=am, anchor_valid_map=vm,
            max_proposals_pre_nms=6000, max_proposals_post_nms=300)
    assert_equal("feature_map", detail["feature_map"].numpy(), fm.numpy())
    scores = smap.reshape(-1).numpy()
    if allow_edge:
        assert_equal("objectness", detail["scores"].numpy(), scores)
    sorted_idx = detail["sorted_idx"]
    top_scores = (scores if allow_edge else scores[vm.reshape(-1) > 0])
    n_unique_top = len(np.unique(np.sort(detail["scores"].numpy())[::-1][: len(sorted_idx)]))
    # margin between consecutive sorted scores (how robust the order is to fp32 noise)
    ss = np.sort(detail["scores"].numpy().astype(np.float64))[::-1][: len(sorted_idx)]
    print("  top-%d: %d unique scores, min gap %.3g, median gap %.3g" % (
        len(sorted_idx), n_unique_top, float(np.min(-np.diff(ss))) if len(ss) > 1 else 0.0,
        float(np.median(-np.diff(ss))) if len(ss) > 1 else 0.0))

    fm_np = fm.numpy()[0]
    out = {
        "seed": np.int64(seed), "height": np.int64(height), "width": np.int64(width),
        "allow_edge": np.int64(1 if allow_edge else 0), "score_threshold": np.float64(score_threshold),
        "weights_seed": np.int64(1234),
        "proposals": ref_props.numpy(), "classes": ref_classes.numpy(), "box_deltas": ref_deltas.numpy(),
        "detections": flatten_detections(ref_det),
        "sorted_idx": sorted_idx.astype(np.int32),
        "n_after_filter": np.int64(detail["n_after_filter"]),
        "scores_sample": scores[::7].copy(), "scores_sha": np.array(sha(scores)),
        "feature_map_sample": fm_np[::(32 if arch else 16), :, :].copy(),   # 32 of the 512 / 1024 channels
        "feature_map_absmean": np.float64(np.abs(fm_np).mean()),
        "rpn_deltas_sample": dmap.reshape(-1, 4).numpy()[::11].copy(),
        "fc2_sample": detail["fc2"].numpy()[:, ::64].copy(),
        "class_logits": detail["class_logits"].numpy(),
    }
    name = "%s_%s.npz" % (arch.lower() if arch else "vgg16", tag)
    np.savez_compressed(os.path.join(GOLDEN, name), **out)
    print("  wrote tests/golden/%s" % name)


def golden_small_ops(ref):
    """Reference-side known answers for the pieces that are cheap to pin exactly."""
    out = {}
    for tag, ishape, fshape in (("vgg", (3, 600, 1000), (512, 37, 62)), ("resnet", (3, 600, 1000), (1024, 38, 63)),
                                ("small", (3, 224, 320), (512, 14, 20)), ("odd", (3, 333, 517), (512, 20, 32))):
        am, vm = ref.anchors.generate_anchor_maps(ishape, fshape, 16)
        am2, vm2 = O.generate_anchor_maps(ishape, fshape, 16)
        assert_equal("anchors[%s]" % tag, am2, am)
        assert_equal("valid[%s]" % tag, vm2, vm)
        out["anchors_%s_shape" % tag] = np.array(list(ishape) + list(fshape), dtype=np.int64)
        out["anchors_%s_sha" % tag] = np.array([sha(am), sha(vm)])
        out["anchors_%s_nvalid" % tag] = np.int64(vm.sum())
        if tag in ("small", "odd"):
            out["anchors_%s_map" % tag] = am
            out["anchors_%s_valid" % tag] = vm

    # RPN ground-truth labelling (anchors.py:137-262): SURVEY's known answer + seeded box sets
    Box_ = ref.training_sample.Box
    gt_sets = {"known": np.array([[100, 200, 400, 700], [50, 50, 300, 180]], dtype=np.float32)}
    rng = np.random.RandomState(21)
    for name, m in (("one", 1), ("five", 5), ("many", 17)):
        y1 = rng.uniform(0, 500, m); x1 = rng.uniform(0, 850, m)
        gt_sets[name] = np.stack([y1, x1, y1 + rng.uniform(20, 99, m) * rng.choice([1, 4], m),
                                  x1 + rng.uniform(20, 149, m) * rng.choice([1, 4], m)], axis=1).astype(np.float32)
    am, vm = ref.anchors.generate_anchor_maps((3, 600, 1000), (512, 37, 62), 16)
    for name, gtc in gt_sets.items():
        rmap, obj, bg = ref.anchors.generate_rpn_map(am, vm, [Box_(1, "x", c) for c in gtc])
        omap, oobj, obg = O.generate_rpn_map(am, vm, gtc)
        assert_equal("rpn_map[%s]" % name, omap, rmap)
        assert_equal("rpn_obj[%s]" % name, oobj, obj)
        assert_equal("rpn_bg[%s]" % name, obg, bg)
        out["rpn_gt_%s" % name] = gtc
        out["rpn_obj_%s" % name] = obj.astype(np.int32)
        out["rpn_nbg_%s" % name] = np.int64(len(bg))
        out["rpn_bg_head_%s" % name] = bg[:64].astype(np.int32)
        out["rpn_map_sha_%s" % name] = np.array(sha(rmap))
        trainable = rmap[..., 0] > 0
        out["rpn_targets_obj_%s" % name] = rmap[(rmap[..., 1] > 0) & trainable][:, 2:6]
    assert len(out["rpn_obj_known"]) == 10 and int(out["rpn_nbg_known"]) == 6847          # SURVEY 8(f2) probe
    assert out["rpn_obj_known"][0].tolist() == [9, 7, 2]

    # numpy float64 decode (math_utils.py:65-97) on seeded inputs
    rng = np.random.RandomState(5)
    anchors = np.abs(rng.randn(64, 4)) * 100 + 20
    deltas = (rng.randn(64, 4) * 0.5).astype(np.float32)
    r = ref.math_utils.convert_deltas_to_boxes(deltas, anchors, [0, 0, 0, 0], [0.1, 0.1, 0.2, 0.2])
    o = O.convert_deltas_to_boxes(deltas, anchors, [0, 0, 0, 0], [0.1, 0.1, 0.2, 0.2])
    assert_equal("convert_deltas_to_boxes", o, r)

    # mAP: the SURVEY's known answer + a seeded random accumulation through the reference's class
    Box = ref.training_sample.Box
    calc = ref.statistics.PrecisionRecallCurveCalculator()
    gts = [(7, np.array([100, 200, 400, 700], dtype=np.float32)), (15, np.array([50, 50, 300, 180], dtype=np.float32))]
    preds = {7: np.array([[100, 200, 400, 700, .9], [110, 210, 390, 690, .8], [0, 0, 50, 50, .7]], dtype=np.float64),
             15: np.array([[50, 50, 300, 180, .6]], dtype=np.float64)}
    calc.add_image_results(preds, [Box(c, "x", k) for c, k in gts])
    known = float(calc.compute_mean_average_precision())
    assert known == 1.0, known
    rng = np.random.RandomState(11)
    calc = ref.statistics.PrecisionRecallCurveCalculator()
    mine = O.MeanAveragePrecision()
    stream = []
    for img_i in range(12):
        g = []
        for _ in range(rng.randint(1, 5)):
            y1, x1 = rng.uniform(0, 400), rng.uniform(0, 700)
            g.append((int(rng.randint(1, 6)), np.array([y1, x1, y1 + rng.uniform(40, 200), x1 + rng.uniform(40, 300)], dtype=np.float32)))
        p = {}
        for c in range(1, 21):
            rows = []
            for cls, k in g:
                if cls == c and rng.rand() < 0.8:
                    rows.append(np.concatenate([k + rng.randn(4) * rng.choice([3, 40]), [rng.uniform(0.05, 1)]]))
            for _ in range(rng.randint(0, 3)):
                y1, x1 = rng.uniform(0, 400), rng.uniform(0, 700)
                rows.append(np.array([y1, x1, y1 + 80, x1 + 120, rng.uniform(0.05, 1)]))
            rows.sort(key=lambda r_: -r_[4])
            p[c] = np.array(rows, dtype=np.float64).reshape(-1, 5)
        calc.add_image_results(p, [Box(c, "x", k) for c, k in g])
        mine.add_image_results(p, g)
        stream.append((g, p))
    ref_map = float(calc.compute_mean_average_precision())
    assert mine.mean_average_precision() == ref_map, (mine.mean_average_precision(), ref_map)
    print("  oracle == reference: mAP stream  %.12f" % ref_map)
    out["map_known_answer"] = np.float64(known)
    out["map_stream_value"] = np.float64(ref_map)
    out["map_stream_gt"] = np.array([[i, c] + k.tolist() for i, (g, _) in enumerate(stream) for c, k in g], dtype=np.float64)
    out["map_stream_pred"] = np.array([[i, c] + row.tolist() for i, (_, p) in enumerate(stream) for c in p for row in p[c]], dtype=np.float64)
    np.savez_compressed(os.path.join(GOLDEN, "small_ops.npz"), **out)
    print("  wrote tests/golden/small_ops.npz")



def sample_positions(n, count=2048):
    return np.unique(np.linspace(0, n - 1, min(count, n)).astype(np.int64))


def golden_train(ref, tag, seed, height, width, steps=2, lr=1e-6, momentum=0.9, weight_decay=5e-4, arch=None, sample_count=2048):
    """
    Runs the REFERENCE's train_step (faster_rcnn.py:228-362) with torch.optim.SGD built as __main__.py:98-105 does,
    asserts oracle/train_oracle.py reproduces losses / gradients / updated weights bit for bit under the same RNG
    seeds, and writes the fixture the GPU parity test compares against.
    """
    import random
    from oracle import train_oracle as TO
    print("train case %s: image seed %d, %dx%d, %d steps" % (tag, seed, height, width, steps))
    sd0 = synthetic.resnet_state_dict(1234, arch) if arch else synthetic.vgg16_state_dict(1234)
    img = (synthetic.image_rgb if arch else synthetic.image)(seed, height, width).unsqueeze(0)
    gts = synthetic.ground_truth(seed, height, width)
    Box = ref.training_sample.Box
    boxes = [Box(c, "x", k) for c, k in gts]
    if arch:
        backbone = ref.resnet.ResNetBackbone(architecture=getattr(ref.resnet.Architecture, arch))
    else:
        backbone = ref.vgg16.VGG16Backbone(dropout_probability=0.0)
    model = ref.faster_rcnn.FasterRCNNModel(num_classes=21, backbone=backbone, allow_edge_proposals=True)
    model.load_state_dict(sd0, strict=True)
    ishape = tuple(img.shape[1:])
    am, vm = ref.anchors.generate_anchor_maps(ishape, backbone.compute_feature_map_shape(ishape), 16)
    rmap, obj, bg = ref.anchors.generate_rpn_map(am, vm, boxes)
    print("  %d gt boxes, %d object / %d background anchors" % (len(gts), len(obj), len(bg)))
    params = []
    for key, value in dict(model.named_parameters()).items():          # __main__.py:98-105
        if not value.requires_grad:
            continue
        if "weight" in key:
            params += [{"params": [value], "weight_decay": weight_decay}]
    optimizer = t.optim.SGD(params, lr=lr, momentum=momentum)
    gt_corners = np.stack([k for _, k in gts]).astype(np.float32)
    gt_cls = np.array([c for c, _ in gts], dtype=np.int64)

    out = {"seed": np.int64(seed), "height": np.int64(height), "width": np.int64(width), "weights_seed": np.int64(1234),
           "steps": np.int64(steps), "lr": np.float64(lr), "momentum": np.float64(momentum),
           "weight_decay": np.float64(weight_decay), "rng_seed": np.int64(100 + seed), "sample_count": np.int64(sample_count)}
    sd = {k: v.clone() for k, v in sd0.items()}
    bufs = None
    keys = TO.trainable_weight_keys(sd0)
    out["train_keys"] = np.array(keys)
    random.seed(100 + seed); t.manual_seed(100 + seed)
    rng_py, rng_t = random.getstate(), t.get_rng_state()
    for step in range(steps):
        # reference
        random.setstate(rng_py); t.set_rng_state(rng_t)
        t0 = time.time()
        loss = model.train_step(optimizer=optimizer, image_data=img, anchor_map=am, anchor_valid_map=vm,
                                gt_rpn_map=t.from_numpy(rmap).unsqueeze(dim=0), gt_rpn_object_indices=[obj],
                                gt_rpn_background_indices=[bg], gt_boxes=[boxes])
        ref_after_py, ref_after_t = random.getstate(), t.get_rng_state()
        print("  step %d reference %.1f s: %s" % (step, time.time() - t0, loss))
        ref_grads = {k: v.grad.detach().clone() for k, v in model.named_parameters() if k in keys}
        ref_sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        # oracle, same RNG state
        random.setstate(rng_py); t.set_rng_state(rng_t)
        detail = {}
        losses, grads, new_sd, bufs = TO.train_step(sd, img, am, vm, t.from_numpy(rmap).unsqueeze(dim=0), obj, bg,
                                                    gt_corners, gt_cls, 21, lr, momentum, weight_decay, bufs, detail=detail)
        for name in ("rpn_class", "rpn_regression", "detector_class", "detector_regression", "total"):
            assert_equal("step%d loss.%s" % (step, name), np.float64(losses[name]), np.float64(getattr(loss, name)))
        # step 0 is bit-identical.  From step 1 on the detector branch's softmax gradient involves cancellation of
        # 1/(p+eps)-sized terms and the CPU BLAS the linear backward runs on does not fix its summation order
        # across buffer alignments: the two runs agree to ~1e-7 of each tensor's largest entry, not bitwise.
        for k in keys:
            if step == 0:
                assert_equal("step%d grad %s" % (step, k[-40:]), grads[k].numpy(), ref_grads[k].numpy())
                assert_equal("step%d new  %s" % (step, k[-40:]), new_sd[k].numpy(), ref_sd[k].numpy())
            else:
                gscale = max(float(v.abs().max()) for v in ref_grads.values())
                for what, a, b in (("grad", grads[k], ref_grads[k]), ("new", new_sd[k], ref_sd[k])):
                    floor = 1e-6 * gscale if what == "grad" else 1e-30      # a tensor of pure rounding noise is not compared
                    d = float((a - b).abs().max()) / max(float(b.abs().max()), floor)
                    assert d < 1e-5, (step, what, k, d)
                    print("  oracle ~= reference: step%d %s %-44s rel max|d| %.2g" % (step, what, k[-44:], d))
        rng_same = random.getstate() == ref_after_py and bool((t.get_rng_state() == ref_after_t).all())
        print("  RNG state after step %d identical: %s" % (step, rng_same))
        for k in sd0:
            if k not in keys:
                assert_equal("step%d frozen %s" % (step, k[-40:]), new_sd[k].numpy(), sd0[k].numpy())
        pre = "s%d_" % step
        out[pre + "losses"] = np.array([losses[n] for n in ("rpn_class", "rpn_regression", "detector_class",
                                                             "detector_regression", "total")], dtype=np.float64)
        out[pre + "rpn_sample_flat"] = detail["rpn_sample_flat"]
        out[pre + "proposal_sample_indices"] = detail["proposal_sample_indices"].astype(np.int32)
        out[pre + "n_rpn_proposals"] = np.int64(detail["rpn_proposals"].shape[0])
        out[pre + "n_labelled"] = np.int64(detail["labelled"][0].shape[0])
        out[pre + "sampled_props"] = detail["sampled"][0].numpy()
        out[pre + "sampled_class_idx"] = detail["sampled"][1].numpy().argmax(axis=1).astype(np.int32)
        for k in keys:
            g = grads[k].numpy().reshape(-1).astype(np.float64)
            pos = sample_positions(g.shape[0], sample_count)
            out[pre + "gnorm/" + k] = np.float64(np.sqrt((g * g).sum()))
            out[pre + "gsample/" + k] = g[pos].astype(np.float32)
            dw = (new_sd[k].numpy().reshape(-1).astype(np.float64) - sd[k].numpy().reshape(-1).astype(np.float64))
            out[pre + "dwnorm/" + k] = np.float64(np.sqrt((dw * dw).sum()))
            out[pre + "dwsample/" + k] = dw[pos].astype(np.float32)
        sd = new_sd
        rng_py, rng_t = ref_after_py, ref_after_t
    name = "train_%s_%s.npz" % (arch.lower() if arch else "vgg16", tag)
    np.savez_compressed(os.path.join(GOLDEN, name), **out)
    print("  wrote tests/golden/%s" % name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calibrate", action="store_true")
    ap.add_argument("--calibrate-resnet", action="store_true")
    ap.add_argument("--only-resnet", action="store_true")
    ap.add_argument("--train", action="store_true", help="only the train-step fixtures (tests/golden/train_*.npz)")
    ap.add_argument("--only-resnet101", action="store_true", help="with --train: only the ResNet-101 fixture")
    ap.add_argument("--only-resnet152", action="store_true", help="only the ResNet-152 inference fixture")
    ap.add_argument("--only-resnet101-600", action="store_true", help="only the ResNet-101 600x1000 inference fixture (round 3)")
    args = ap.parse_args()
    t.manual_seed(0)
    ref = reference_shims.install(O)
    if args.calibrate:
        calibrate(ref)
        return
    if args.calibrate_resnet:
        calibrate_resnet(ref)
        return
    os.makedirs(GOLDEN, exist_ok=True)
    if args.only_resnet101_600:
        sd = synthetic.resnet_state_dict(1234, "ResNet101")
        run_case(ref, "600x1000_s2", sd, 2, 600, 1000, True, 0.05, arch="ResNet101")
        return
    if args.only_resnet152:
        sd = synthetic.resnet_state_dict(1234, "ResNet152")
        run_case(ref, "250x333_s7", sd, 7, 250, 333, True, 0.05, arch="ResNet152")
        return
    if args.train and args.only_resnet101:
        golden_train(ref, "320x416_s6", 6, 320, 416, arch="ResNet101", sample_count=512)
        return
    if args.train:
        if not args.only_resnet:
            golden_train(ref, "352x480_s4", 4, 352, 480)
            golden_train(ref, "416x544_s6", 6, 416, 544)
        golden_train(ref, "352x480_s4", 4, 352, 480, arch="ResNet50")
        golden_train(ref, "320x416_s6", 6, 320, 416, arch="ResNet101", sample_count=512)
        return
    if not args.only_resnet:
        golden_small_ops(ref)
        sd = synthetic.vgg16_state_dict(1234)
        run_case(ref, "600x1000_s0", sd, 0, 600, 1000, True, 0.05)
        run_case(ref, "224x320_s3", sd, 3, 224, 320, True, 0.05)           # A = 2520 < 6000 anchors
        run_case(ref, "333x517_s5_noedge", sd, 5, 333, 517, False, 0.05)   # ragged size, valid-anchor filter
    sd = synthetic.resnet_state_dict(1234, "ResNet50")
    run_case(ref, "600x1000_s0", sd, 0, 600, 1000, True, 0.05, arch="ResNet50")
    run_case(ref, "250x333_s7", sd, 7, 250, 333, True, 0.05, arch="ResNet50")   # ceil() feature map 16x21
    sd = synthetic.resnet_state_dict(1234, "ResNet101")
    run_case(ref, "224x320_s3", sd, 3, 224, 320, True, 0.05, arch="ResNet101")
    run_case(ref, "600x1000_s2", sd, 2, 600, 1000, True, 0.05, arch="ResNet101")
    sd = synthetic.resnet_state_dict(1234, "ResNet152")
    run_case(ref, "250x333_s7", sd, 7, 250, 333, True, 0.05, arch="ResNet152")


if __name__ == "__main__":
    main()
