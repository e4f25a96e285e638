"""
The HELD-OUT parity sweep: 16 VGG-16, 8 ResNet-50 and 8 ResNet-101 600x1000 cases that no table, tolerance or expectation
of this repository was tuned on (tests/golden/holdout/, written by oracle/make_holdout.py from the IMPORTED REFERENCE and from
oracle/f64_truth.py, the float64 evaluation of the same network on the same inputs).

Two questions, two kinds of assert.

(1) Is the HIP path's arithmetic as good as the reference's?  The yardstick is the float64 truth: every proposal / detection row is the
    decode of one anchor, and its error is its distance from the float64 decode of that anchor.  The reference's own float32 run
    (torch-CPU) sits at a measured distance from the truth -- stored per case in the fixture; VGG-16: median 1.0e-4 px, p95 2.9e-4, worst
    row 6e-4; ResNet-101: median 2.0e-4, p95 6.2e-4, worst row 1.4e-3 px, i.e. the REFERENCE ITSELF is beyond north_star's 1e-3 px of the
    exact answer on 1-3 rows of every ResNet-101 image.  The criterion the default arithmetic table is chosen by:

        pooled over the held-out set, ours-vs-truth  <=  K_TRUTH x reference-vs-truth     (median of medians and median of p95s)

    K_TRUTH is set PER TABLE at its measured level plus a stated margin (below).  For orientation: the table with NO split-operand layer
    (every GEMM on the exact-f32 MFMA pipe, Winograd layers in float32) measures 1.50 / 1.46 on the VGG-16 set, the all-direct exact-f32
    table 1.84 / 1.78 -- a float32 FMA chain over K = 4608 rounds more than oneDNN's blocked accumulation does; the default tables sit at
    VGG-16 1.30 / 1.17, ResNet-50 1.14 / 1.18, ResNet-101 0.88 / 0.96 (closer to the truth than the reference's own run).  The measured
    value of every table is printed by tools/holdout_report.py and recorded in DESIGN.md section 4.

(2) north_star's bar, "boxes within 1e-3 px of the PyTorch reference": the fraction of the reference's rows the HIP path reproduces within
    1e-3 px, pooled over the held-out set -- a printed and asserted number, counted two ways: AT THE SAME ROW INDEX (same proposals in the
    same order; one near-tied NMS decision that falls the other way shifts every later row of that image by one index), and as a SET
    (nearest row).  What two float32 runs that are each e_ref from the truth can agree to is ~sqrt(2) e_ref per row, so the floors are
    are the measured fractions minus a margin of ~0.002 (VGG-16, ResNet-50) -- ResNet-101: its reference sits 1e-3 px from the truth by
    itself, and the saturated class scores of its synthetic head (exactly 1.0f for ~150 rows of one class) make the per-class NMS order a
    matter of last-bit ties: floors 0.95 / 0.90 / 0.94 against a measured 0.973 / 0.938 / 0.967.

(3) The kernels are deterministic, so the pooled COUNTS of a run are reproducible to the row: tests/golden/holdout/observed_counts.json is the
    committed snapshot of the last measured run and a run may lose at most COUNT_SLACK rows against it (a regression that stays
    inside the floors is still caught).  test_the_gates_catch_one_layer_in_bf16 shows the gates measure something: ONE convolution's filter
    rounded to bf16 fails them on every architecture.
"""
import json
import os

import numpy as np
import pytest

import holdout_lib as H

pytestmark = pytest.mark.gpu

# (1) K_TRUTH per table = measured (profiles/r0*/holdout_*.json: proposals median, p95, detections median, p95) + a margin of ~0.1.
# Measured: VGG-16 1.30 / 1.17 / 1.30 / 1.18 (in flight 1.28 / 1.21 / 1.26 / 1.30); ResNet-50 1.14 / 1.18 / 1.18 / 1.28; ResNet-101 0.88 /
# 0.96 / 0.85 / 1.02; ResNet-101 with the f32x3 backbone (not its default) 1.30 / 1.25 / 1.27 / 1.17.  Round 4's single K = 1.5 let a
# regression up to the level of the all-exact-f32 table pass.
K_TRUTH = {"VGG16": 1.40, "ResNet50": 1.38, "ResNet101": 1.10, "ResNet101_g3": 1.40}
# (2) pooled fraction of the reference's rows reproduced within 1e-3 px: (proposals as a set, proposals at the same row index, detections
# as a set).  Measured: VGG-16 0.9985 / 0.9981 / 0.9991 (in flight 0.9994 / 0.9990 / 0.9996); ResNet-50 0.9988 / 0.9988 / 1.0;
# ResNet-101 0.9733 / 0.9383 / 0.9671; with the f32x3 backbone 0.9604 / 0.9250 / 0.9526.
ROW_FRACTION_FLOOR = {"VGG16": (0.997, 0.996, 0.997), "ResNet50": (0.997, 0.997, 0.998), "ResNet101": (0.95, 0.90, 0.94),
                      "ResNet101_g3": (0.94, 0.90, 0.93)}
MIN_CASES = {"VGG16": 16, "ResNet50": 8, "ResNet101": 8}
# (3) the committed counts of the last measured run (tools/holdout_report.py --snapshot)
COUNT_SLACK = 2
SNAPSHOT = os.path.join(H.HOLDOUT, "observed_counts.json")
COUNT_KEYS = ("prop_rows_ok", "prop_rows_ok_set", "det_rows_ok", "det_rows_ok_set")


def violations(key, s):
    """the gates of table `key` that summary `s` (report()) misses: a list of strings, empty = admitted"""
    k = K_TRUTH[key]
    bad = []
    for name, ours, ref in (("proposals", s["prop_vs_truth"], s["ref_prop_vs_truth"]), ("detections", s["det_vs_truth"], s["ref_det_vs_truth"])):
        for q in ("median", "p95"):
            if not ours[q] <= k * ref[q]:
                bad.append("%s %s vs truth x%.3f > K %.2f" % (name, q, ours[q] / ref[q], k))
    fp, fpi, fd = ROW_FRACTION_FLOOR[key]
    for name, v, floor in (("prop_set_fraction", s["prop_set_fraction"], fp), ("prop_row_fraction", s["prop_row_fraction"], fpi),
                           ("det_set_fraction", s["det_set_fraction"], fd)):
        if not v >= floor:
            bad.append("%s %.4f < %.4f" % (name, v, floor))
    return bad


def snapshot_violations(key, s):
    if not os.path.exists(SNAPSHOT):
        return ["%s missing" % SNAPSHOT]
    with open(SNAPSHOT) as f:
        snap = json.load(f).get(key)
    if snap is None:
        return ["no snapshot for %s" % key]
    return ["%s %d < snapshot %d - %d" % (n, s[n], snap[n], COUNT_SLACK) for n in COUNT_KEYS if s[n] < snap[n] - COUNT_SLACK]


def sweep(arch, slot=0, attrs=None):
    files = H.cases(arch)
    assert len(files) >= MIN_CASES[arch], "held-out fixtures missing: run oracle/make_holdout.py in the build container"
    results, models = [], {}
    for f in files:
        g = np.load(f)
        ws = int(g["weights_seed"])
        if ws not in models:
            models.clear()
            models[ws] = H.build_model(arch, ws)
            for k, v in (attrs or {}).items():
                setattr(models[ws], k, v)
        r = H.measure(models[ws], g, slot)
        print(H.format_line(r))
        results.append(r)
    return results


def report(arch, results):
    rows = sum(r["prop_rows"] for r in results)
    ok = sum(r["prop_rows_within_gate"] for r in results)
    oks = sum(r["prop_rows_matched_within_gate"] for r in results)
    drows = sum(r["det_rows"] for r in results)
    dok = sum(r["det_rows_within_gate"] for r in results)
    doks = sum(r["det_rows_matched_within_gate"] for r in results)
    same_order = sum(1 for r in results if r["order_identical"])
    p, rp = H.pooled(results, "prop_vs_truth"), H.pooled(results, "ref_prop_vs_truth")
    d, rd = H.pooled(results, "det_vs_truth"), H.pooled(results, "ref_det_vs_truth")
    out = {"arch": arch, "cases": len(results), "prop_row_fraction": ok / max(rows, 1), "det_row_fraction": dok / max(drows, 1),
           "prop_set_fraction": oks / max(rows, 1), "det_set_fraction": doks / max(drows, 1), "images_with_identical_order": same_order,
           "prop_rows": rows, "det_rows": drows, "prop_rows_ok": ok, "prop_rows_ok_set": oks, "det_rows_ok": dok, "det_rows_ok_set": doks, "prop_vs_truth": p, "ref_prop_vs_truth": rp, "det_vs_truth": d, "ref_det_vs_truth": rd,
           "fm_err_median": float(np.median([r["fm_err"] for r in results])),
           "ref_fm_err_median": float(np.median([r["ref_fm_err"] for r in results]))}
    print("HELD-OUT %s (%d cases): reference rows reproduced within 1e-3 px: proposals %d/%d = %.4f as a set, %d = %.4f at the same row index "
          "(%d of %d images: all rows in the reference's order); detections %d/%d = %.4f as a set, %d = %.4f at the same row"
          % (arch, len(results), oks, rows, out["prop_set_fraction"], ok, out["prop_row_fraction"], same_order, len(results),
             doks, drows, out["det_set_fraction"], dok, out["det_row_fraction"]))
    print("   proposals vs float64 truth: median %.3g px (reference %.3g: x%.2f), p95 %.3g (reference %.3g: x%.2f), worst row %.3g (reference %.3g)"
          % (p["median"], rp["median"], p["median"] / rp["median"], p["p95"], rp["p95"], p["p95"] / rp["p95"], p["max"], rp["max"]))
    print("   detections vs float64 truth: median %.3g px (reference %.3g: x%.2f), p95 %.3g (reference %.3g: x%.2f); feature map %.3g of max (reference %.3g)"
          % (d["median"], rd["median"], d["median"] / rd["median"], d["p95"], rd["p95"], d["p95"] / rd["p95"], out["fm_err_median"],
             out["ref_fm_err_median"]))
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", "holdout_%s.json" % arch.lower()), "w") as f:
            json.dump({"summary": out, "cases": results}, f, indent=1)
    except OSError:
        pass
    return out


# ("VGG16", 1): the arithmetic table of the THROUGHPUT configuration -- predict_async's in-flight slots run the 512-channel f32x3 layers in
# the one-launch form (FasterRCNNModel.layer_tables; the headline of bench.py is measured on it), slot 0 in the three-launch form
@pytest.mark.parametrize("arch,slot", [("VGG16", 0), ("VGG16", 1), ("ResNet50", 0), ("ResNet101", 0)])
def test_holdout_sweep(arch, slot):
    results = sweep(arch, slot)
    s = report(arch if slot == 0 else "%s_inflight" % arch, results)
    if arch == "VGG16":
        from fasterrcnn_amd import _native as nv
        m = H.build_model(arch, 1234)
        arch_tables = m.layer_tables(slot)
        assert set(nv.DEFAULT_INFLIGHT_X3F_LAYERS_VGG16) <= (set(arch_tables[2]) if slot else set(arch_tables[1]) | set(arch_tables[2])), arch_tables
        assert set(nv.DEFAULT_ALONE_X3F_LAYERS_VGG16) <= set(arch_tables[2])
        del m
    # every case: the same number of proposals as the reference, every row the decode of a candidate anchor (no gross misses)
    for r in results:
        assert r["n_proposals"] == r["prop_rows"], r
        assert r["prop_vs_truth"]["n_far"] == 0, r
    # (1) the arithmetic criterion (ours-vs-truth <= K_TRUTH x reference-vs-truth, pooled), (2) north_star's bar against the reference as
    # pooled fractions, (3) the committed counts of the last measured run
    bad = violations(arch, s) + snapshot_violations(arch if slot == 0 else "%s_inflight" % arch, s)
    assert not bad, bad


def test_resnet101_with_the_f32x3_backbone_meets_the_truth_criterion():
    """ResNet-101 with bottleneck_g3 = "backbone" (NOT its default: FasterRCNNModel.__init__) is admitted by the arithmetic criterion -- the
    table a user who wants the 32 % may switch on -- at a measured cost in the reference's rows (the set fraction is printed and held to the
    same floor as the default table's)."""
    results = sweep("ResNet101", 0, {"bottleneck_g3": "backbone"})
    s = report("ResNet101_g3", results)
    bad = violations("ResNet101_g3", s) + snapshot_violations("ResNet101_g3", s)
    assert not bad, bad


# one layer's filter rounded to bf16 (8 significant bits instead of the 22 of the f32x3 split): the regression the gates exist for
DOWNGRADED_LAYER = {"VGG16": "_stage1_feature_extractor._block3_conv2.weight",
                    "ResNet50": "_stage1_feature_extractor._feature_extractor.5.1.conv2.weight",
                    "ResNet101": "_stage1_feature_extractor._feature_extractor.6.11.conv2.weight"}


@pytest.mark.parametrize("arch", ["VGG16", "ResNet50", "ResNet101"])
def test_the_gates_catch_one_layer_in_bf16(arch):
    """VERDICT r4 'do this' 3: a deliberate one-layer downgrade to bf16 operands must fail at least one gate per architecture -- otherwise
    the gates above measure nothing.  The FIXTURES stay those of the float32 weights; the HIP path runs with ONE convolution's filter
    rounded to bf16 (every other layer, and the arithmetic of that layer, untouched)."""
    import torch
    files = H.cases(arch)[:4]
    results, models = [], {}
    for f in files:
        g = np.load(f)
        ws = int(g["weights_seed"])
        if ws not in models:
            models.clear()
            m = H.build_model(arch, ws)
            sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
            key = DOWNGRADED_LAYER[arch]
            sd[key] = sd[key].to(torch.bfloat16).to(torch.float32)
            m.load_state_dict(sd, strict=True)
            models[ws] = m.cuda().eval()
        results.append(H.measure(models[ws], g, 0))
    s = report("%s_one_layer_bf16" % arch, results)
    bad = violations(arch, s)
    print("gates missed with %s in bf16: %s" % (DOWNGRADED_LAYER[arch], bad))
    assert bad, "no gate noticed a bf16 layer: %s" % s
