"""tools/holdout_report.py -- the held-out parity sweep (tests/golden/holdout/, oracle/make_holdout.py) for a list of arithmetic tables.

  python tools/holdout_report.py [--arch VGG16|ResNet50|ResNet101] [--tables default,f32,x6,x3_all,...] [--out gpurun_out/holdout.json]

For every table: per-case lines (tests/holdout_lib.py: format_line) and the pooled numbers the default table is CHOSEN by
(DESIGN.md section 4): the HIP path's distance from the float64 truth next to the reference's own, and the fraction of the reference's
rows reproduced within 1e-3 px.  Development aid; the asserted form of the same measurement is tests/test_holdout_gpu.py."""
import argparse
import json
import os
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np

import holdout_lib as H
from fasterrcnn_amd import _native as nv

X6 = ("conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3", "rpn_trunk")

# name -> attributes set on the model (VGG-16).  "default" leaves the model as constructed.
X3F = ("conv2_2", "conv3_1", "conv3_2", "conv3_3")
VGG_TABLES = {
    "default": {},
    "f32": {"winograd_x6_layers": (), "winograd_x3_layers": (), "winograd_x3f_layers": (), "fc_math_mode": "f32"},   # every GEMM on the exact-f32 pipe
    "f32_direct": {"math_mode": "f32", "fc_math_mode": "f32"},                                         # no Winograd at all
    "x6": {"winograd_x6_layers": X6, "winograd_x3_layers": (), "winograd_x3f_layers": (), "fc_math_mode": "f32x6"},
    "x3_all": {"winograd_x6_layers": X6, "winograd_x3_layers": X6, "winograd_x3f_layers": (), "fc_math_mode": "f32x3"},   # round 4's table without the one-launch layers
    "x3_conv_only": {"winograd_x6_layers": X6, "winograd_x3_layers": X6, "winograd_x3f_layers": (), "fc_math_mode": "f32"},
    "x3_fc_only": {"winograd_x6_layers": (), "winograd_x3_layers": (), "winograd_x3f_layers": (), "fc_math_mode": "f32x3"},
    "x3f_only": {"winograd_x6_layers": (), "winograd_x3_layers": (), "winograd_x3f_layers": X3F, "fc_math_mode": "f32"},
    "x3f_conv1_2": {"winograd_x3f_layers": ("conv1_2", "conv2_1") + X3F},
    "x3_everything": {"winograd_x6_layers": X6, "winograd_x3_layers": X6, "winograd_x3f_layers": ("conv2_1",) + X3F, "fc_math_mode": "f32x3"},
}
RESNET_TABLES = {
    "default": {},
    "r3_default": {"bottleneck_g3": "off"},            # round 3's default: float32 backbone, layer4 head + RPN trunk in f32x3
    "f32": {"bottleneck_g3": "off", "x6_conv1x1": "off", "winograd_x6_layers": (), "winograd_x3_layers": ()},
    "f32_direct": {"bottleneck_g3": "off", "math_mode": "f32", "x6_conv1x1": "off"},
    "x6_head": {"bottleneck_g3": "off", "x6_conv1x1": "head", "x6_conv1x1_arith": "f32x6", "winograd_x6_layers": (), "winograd_x3_layers": ()},
    "x3_all": {"bottleneck_g3": "off", "x6_conv1x1": "all", "x6_conv1x1_arith": "f32x3", "winograd_x6_layers": ("rpn_trunk",), "winograd_x3_layers": ("rpn_trunk",)},
    "g3_backbone": {"bottleneck_g3": "backbone"},     # layer1..3 in the f32x3 arithmetic under one scale per tensor (conv_gather_x3_kernel)
    "g3_all": {"bottleneck_g3": "all"},               # ... and the per-RoI layer4
}


def apply(model, attrs):
    for k, v in attrs.items():
        setattr(model, k, v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="VGG16")
    ap.add_argument("--tables", default="default,f32")
    ap.add_argument("--out", default=None)
    ap.add_argument("--max-cases", type=int, default=0)
    ap.add_argument("--slot", type=int, default=0, help="0 = forward / predict; 1 = an in-flight slot of predict_async (VGG-16: its one-launch table)")
    ap.add_argument("--stress", action="store_true", help="the stress set (tests/golden/stress/, oracle/make_stress.py) instead of the held-out set")
    ap.add_argument("--snapshot", default=None, metavar="KEY", help="merge the pooled counts of the FIRST table into gpurun_out/observed_counts.json under KEY "
                    "(copy to tests/golden/holdout/observed_counts.json: the committed counts tests/test_holdout_gpu.py holds a run to)")
    args = ap.parse_args()
    tables = VGG_TABLES if args.arch == "VGG16" else RESNET_TABLES
    files = H.stress_cases(args.arch) if args.stress else H.cases(args.arch)
    if args.max_cases:
        files = files[: args.max_cases]
    report = {}
    for name in args.tables.split(","):
        models = {}
        results = []
        for f in files:
            g = np.load(f)
            ws = (int(g["weights_seed"]), str(g["kind"]) if "kind" in g else None)
            if ws not in models:
                models.clear()
                models[ws] = H.build_model(args.arch, ws[0], ws[1])
                apply(models[ws], tables[name])
            r = H.measure(models[ws], g, args.slot)
            results.append(r)
            print("%-12s %s" % (name, H.format_line(r)), flush=True)
        rows = sum(r["prop_rows"] for r in results)
        ok = sum(r["prop_rows_within_gate"] for r in results)
        oks = sum(r["prop_rows_matched_within_gate"] for r in results)
        doks = sum(r["det_rows_matched_within_gate"] for r in results)
        drows = sum(r["det_rows"] for r in results)
        dok = sum(r["det_rows_within_gate"] for r in results)
        p, rp = H.pooled(results, "prop_vs_truth"), H.pooled(results, "ref_prop_vs_truth")
        d, rd = H.pooled(results, "det_vs_truth"), H.pooled(results, "ref_det_vs_truth")
        print("== %s / %s: %d cases | reference rows within 1e-3 px: proposals %d/%d = %.4f (set %.4f), detections %d/%d = %.4f (set %.4f) | proposals vs truth: "
              "median %.3g (ref %.3g, x%.2f) p95 %.3g (ref %.3g, x%.2f) max %.3g (ref %.3g) far %d | detections vs truth: median %.3g (ref %.3g, x%.2f) "
              "p95 %.3g (ref %.3g, x%.2f) | fm %.3g (ref %.3g)" % (
                  args.arch, name, len(results), ok, rows, ok / max(rows, 1), oks / max(rows, 1), dok, drows, dok / max(drows, 1), doks / max(drows, 1),
                  p["median"], rp["median"], p["median"] / rp["median"], p["p95"], rp["p95"], p["p95"] / rp["p95"], p["max"], rp["max"], p["n_far"],
                  d["median"], rd["median"], d["median"] / rd["median"], d["p95"], rd["p95"], d["p95"] / rd["p95"],
                  float(np.median([r["fm_err"] for r in results])), float(np.median([r["ref_fm_err"] for r in results]))), flush=True)
        report[name] = {"cases": results, "pooled": {"prop_vs_truth": p, "ref_prop_vs_truth": rp, "det_vs_truth": d, "ref_det_vs_truth": rd,
                                                     "prop_rows": rows, "prop_rows_within_gate": ok, "prop_rows_matched_within_gate": oks, "det_rows": drows,
                                                     "det_rows_within_gate": dok, "det_rows_matched_within_gate": doks}}
    if args.snapshot:
        path = os.path.join("gpurun_out", "observed_counts.json")
        snap = json.load(open(path)) if os.path.exists(path) else {}
        q = report[args.tables.split(",")[0]]["pooled"]
        snap[args.snapshot] = {"prop_rows_ok": q["prop_rows_within_gate"], "prop_rows_ok_set": q["prop_rows_matched_within_gate"],
                               "det_rows_ok": q["det_rows_within_gate"], "det_rows_ok_set": q["det_rows_matched_within_gate"],
                               "prop_rows": q["prop_rows"], "det_rows": q["det_rows"], "source": "tools/holdout_report.py --arch %s --tables %s --slot %d" % (
                                   args.arch, args.tables.split(",")[0], args.slot)}
        os.makedirs("gpurun_out", exist_ok=True)
        with open(path, "w") as f:
            json.dump(snap, f, indent=1)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
