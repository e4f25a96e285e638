"""
oracle/make_stress.py -- TEST INFRASTRUCTURE ONLY; runs in the BUILD CONTAINER (needs /root/reference).

  python oracle/make_stress.py --calibrate      prints STRESS_CALIBRATION (paste into fasterrcnn_amd/synthetic.py)
  python oracle/make_stress.py [--only TAG]     writes tests/golden/stress/*.npz

The STRESS parity set ("what's missing" 3 / "do this" 2): the held-out sweep of oracle/make_holdout.py -- imported reference,
oracle asserted bit-identical, float64 truth, the reference's own distance from the truth -- on inputs built against the data-dependent
operand scaling of the HIP path's f32x3 arithmetic (fasterrcnn_amd/synthetic.py, "stress recipes"):

  heavy        Student-t(3) weights, log-normal per-channel gains, a few output channels x64; smooth image
  edges        standard weights; image of hard-edged rectangles, saturated blocks, flat black regions, one-pixel lines
  heavy_edges  both
  outlier      one activation channel 2^12 above the tensor's median (VGG-16: conv3_2's output; ResNet: layer1's residual stream); edges image

tests/test_stress_gpu.py runs the HIP path on the same seeded inputs and asserts the two criteria of tests/test_holdout_gpu.py.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch as t

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import frcnn_oracle as O          # noqa: E402
from oracle import f64_truth as T             # noqa: E402
from oracle import reference_shims            # noqa: E402
from oracle import make_holdout as MH         # noqa: E402
from oracle.make_golden import build_reference_model   # noqa: E402
from fasterrcnn_amd import synthetic          # noqa: E402

STRESS = os.path.join(ROOT, "tests", "golden", "stress")

# (architecture, kind, image seed, weights seed); none of the seeds appears anywhere else in the repository
CASES = ([("VGG16", "heavy", s, 7001) for s in (501, 502)] + [("VGG16", "heavy", 503, 7002)] +
         [("VGG16", "edges", s, 1234) for s in (504, 505)] +
         [("VGG16", "heavy_edges", 506, 7001), ("VGG16", "heavy_edges", 507, 7002)] +
         [("VGG16", "outlier", 508, 7003), ("VGG16", "outlier", 509, 7004)] +
         [("ResNet50", "heavy", 601, 7101), ("ResNet50", "edges", 602, 1234), ("ResNet50", "heavy_edges", 603, 7101),
          ("ResNet50", "outlier", 604, 7103), ("ResNet50", "outlier", 605, 7104)])


def case_tag(arch, kind, seed, wseed):
    return "%s_%s_s%d_w%d" % (arch.lower(), kind, seed, wseed)


def state_dict(arch, kind, wseed, calibration=None):
    if arch == "VGG16":
        return synthetic.stress_vgg16_state_dict(wseed, kind, calibration=calibration)
    return synthetic.stress_resnet_state_dict(wseed, kind, arch, calibration=calibration)


def image(arch, kind, seed):
    return (synthetic.stress_image if arch == "VGG16" else synthetic.stress_image_rgb)(seed, kind, MH.HEIGHT, MH.WIDTH).unsqueeze(0)


def calibrate(ref):
    """The five multipliers per (architecture, weight recipe, weights seed), measured with the reference's own modules on the FIRST image the
    recipe is used with (make_golden.calibrate / calibrate_resnet: same targets)."""
    done = {}
    for arch, kind, seed, wseed in CASES:
        wk = synthetic.stress_weights_kind(kind)
        if wk == "he" or (arch, wk, wseed) in done:
            continue
        vgg = arch == "VGG16"
        sd = state_dict(arch, kind, wseed, calibration={})
        img = image(arch, kind, seed)
        cal = {}
        with t.no_grad():
            model = build_reference_model(ref, sd, True, None if vgg else arch)
            fm = model._stage1_feature_extractor(image_data=img)
            rpn = model._stage2_region_proposal_network
            if vgg:
                k = "_stage1_feature_extractor._block5_conv3.weight"
                cal[k] = 1.0 / float(fm.std())
                fm = fm * cal[k]
                y = t.relu(rpn._rpn_conv1(fm))
            else:
                y = t.relu(rpn._rpn_conv1(fm))
                k = "_stage2_region_proposal_network._rpn_conv1.weight"
                cal[k] = 1.0 / float(y.std())
                y = y * cal[k]
            cal["_stage2_region_proposal_network._rpn_class.weight"] = 1.0 / float(rpn._rpn_class(y).std())
            cal["_stage2_region_proposal_network._rpn_boxes.weight"] = 0.3 / float(rpn._rpn_boxes(y).std())
            sd2 = state_dict(arch, kind, wseed, calibration=cal)
            detail = {}
            O.forward(sd2, img, detail=detail)
            det = build_reference_model(ref, sd2, True, None if vgg else arch)._stage3_detector_network
            cal["_stage3_detector_network._classifier.weight"] = 3.0 / float(det._classifier(detail["fc2"]).std())
            cal["_stage3_detector_network._regressor.weight"] = 1.0 / float(det._regressor(detail["fc2"]).std())
        done[(arch, wk, wseed)] = cal
        print("calibrated", arch, wk, wseed, file=sys.stderr, flush=True)
    print("STRESS_CALIBRATION = {")
    for (arch, wk, wseed), cal in done.items():
        print('    ("%s", "%s", %d): {' % (arch, wk, wseed))
        for k, v in cal.items():
            print('        "%s": %.9g,' % (k, v))
        print("    },")
    print("}")


def activation_spread(arch, sd, img):
    """max / median-of-positives of the tensor the 'outlier' recipe plants its channel in (printed: the recipe's own evidence)"""
    with t.no_grad():
        if arch == "VGG16":
            x = img
            for name, _, _ in synthetic._VGG_CONVS:
                x = t.relu(t.nn.functional.conv2d(x, sd["_stage1_feature_extractor." + name + ".weight"], sd["_stage1_feature_extractor." + name + ".bias"], padding=1))
                if name == "_block3_conv2":
                    break
                if name.endswith(("_block1_conv2", "_block2_conv2")):
                    x = t.nn.functional.max_pool2d(x, 2)
        else:
            fe = "_stage1_feature_extractor._feature_extractor."
            x = t.relu(O._conv_bn(img, sd, fe + "0.weight", fe + "1.", stride=2, padding=3))
            x = t.nn.functional.max_pool2d(x, 3, 2, 1)
            x = O._layer(x, sd, fe + "4.", 1)
    pos = x[x > 0]
    return float(x.max()), float(pos.median())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calibrate", action="store_true")
    ap.add_argument("--only", default=None)
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    if args.threads:
        t.set_num_threads(args.threads)
    t.manual_seed(0)
    ref = reference_shims.install(O)
    if args.calibrate:
        calibrate(ref)
        return
    os.makedirs(STRESS, exist_ok=True)
    summary = []
    for arch, kind, seed, wseed in CASES:
        tag = case_tag(arch, kind, seed, wseed)
        if args.only and tag != args.only:
            continue
        sd = state_dict(arch, kind, wseed)
        img = image(arch, kind, seed)
        mx, med = activation_spread(arch, sd, img)
        print("%s: planted-tensor max / median of positives = %.3g / %.3g = 2^%.1f" % (tag, mx, med, np.log2(mx / med)), flush=True)
        extra = {"kind": np.array(kind), "planted_max": np.float64(mx), "planted_median": np.float64(med)}
        r = MH.run_case(ref, arch, seed, wseed, sd, T.to_f64(sd), img=img, tag=tag, out_dir=STRESS, extra=extra)
        r["planted_max_over_median_log2"] = float(np.log2(mx / med))
        summary.append(r)
    if not args.only:
        with open(os.path.join(STRESS, "reference_vs_truth.json"), "w") as f:
            json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
