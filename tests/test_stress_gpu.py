"""
The STRESS parity sweep (VERDICT r4 "what's missing" 3 / "do this" 2): the two questions of tests/test_holdout_gpu.py asked again on inputs
built against the data-dependent operand scaling of the HIP path's f32x3 arithmetic -- per-tile scales in the Winograd layers, per-row
scales in the fc layers, ONE power-of-two scale per tensor in the ResNet-50 backbone (csrc/conv_gather.hip):

  heavy        Student-t(3) weights, log-normal per-output-channel gains, two output channels x64 in three layers; smooth image
  edges        standard weights; hard-edged rectangles, saturated (255) blocks, flat black regions, one-pixel lines
  heavy_edges  both
  outlier      one activation channel 2^12 .. 2^15 above its tensor's median (VGG-16: conv3_2's output; ResNet-50: the residual stream
               from layer1 on), hard-edged image

tests/golden/stress/*.npz (oracle/make_stress.py; recipes in fasterrcnn_amd/synthetic.py): per case the IMPORTED reference's outputs (the
oracle asserted bit-identical), the float64 truth's candidates / detections and the reference's own distance from the truth.  9 VGG-16 + 5
ResNet-50 cases; nothing in the repository was tuned on them -- the default tables were frozen (round 4) before these inputs existed.

Asserted:

(1) per architecture, pooled over its cases: ours-vs-truth <= K_STRESS x reference-vs-truth (proposals and detections, median and p95).
    K_STRESS = 1.5 is the ADMISSION criterion itself: the level of a table with every GEMM on the exact-f32 matrix
    pipe on the held-out set.  (The held-out sweep's per-table K_TRUTH are regression gates: measured + margin on ITS inputs.)  Measured
    here (round 5, profiles/r05/stress_*.json): VGG-16 1.20 / 1.03 / 1.33 / 1.28 (every slot runs one table), ResNet-50 1.16 / 1.26 / 1.16 /
    1.35; the all-exact-f32 table on the same cases: VGG-16 1.52 / 1.37 / 1.57 / 1.35, ResNet-50 1.12 / 1.27 / 1.21 / 0.91.
(2) per case, north_star's bar against the reference -- where the reference itself allows it: on these inputs the REFERENCE's float32 run
    sits up to 1.1e-3 px (p95; worst row 2.0e-3) from the float64 truth, so two equally good float32 runs cannot agree to 1e-3 px on every
    row.  A row on which the reference is within HALF the gate of the exact answer is a row an equally good run reproduces within the gate:
    rows of ours within 1e-3 px of the reference's row  >=  rows of the reference within 0.5e-3 px of the truth  - max(3, 2 % of the rows),
    proposals (at the same row index) and detections alike.  Measured: 231 >= 193 on the worst-conditioned case, 283 >= 278 and (detections)
    174 >= 174 and 185 >= 188 - 3 on the tightest (one near-tied per-class NMS decision moves three rows).
(3) per case: as many proposals as the reference, no row that is not the decode of a candidate anchor, and ZERO saturated operands in the
    per-tensor-scaled ResNet-50 backbone (frcnn_x3_saturation_events).
"""
import numpy as np
import pytest

import holdout_lib as H
import test_holdout_gpu as HG

pytestmark = pytest.mark.gpu

MIN_CASES = {"VGG16": 8, "ResNet50": 4}
# the admission criterion's level on this set: MEASURED (profiles/r05/stress_*.json, unchanged by round 6's bit-identical kernel changes) + 0.1, per
# architecture -- VGG-16: proposals 1.20 / 1.03, detections 1.33 / 1.28 (median / p95 of our distance from the float64 truth over the reference's
# own); ResNet-50: 1.16 / 1.26 and 1.16 / 1.35.  (Round 5 gated at the admission level 1.5 itself: a regression from 1.20 to 1.49 passed.)
K_STRESS = {"VGG16": 1.43, "ResNet50": 1.46}


def sweep(arch, slot):
    from fasterrcnn_amd import _native as nv
    files = H.stress_cases(arch)
    assert len(files) >= MIN_CASES[arch], "stress fixtures missing: run oracle/make_stress.py in the build container"
    results, models = [], {}
    for f in files:
        g = np.load(f)
        key = (str(g["kind"]), int(g["weights_seed"]))
        if key not in models:
            models.clear()
            models[key] = H.build_model(arch, key[1], key[0])
        sat0 = nv.x3_saturation_count()
        r = H.measure(models[key], g, slot)
        r["saturated_operands"] = nv.x3_saturation_count() - sat0
        r["ref_prop_rows_within_half_gate"] = int((g["ref_prop_err"] <= 0.5 * H.GATE).sum())
        r["ref_det_rows_within_half_gate"] = int((g["ref_det_err"] <= 0.5 * H.GATE).sum())
        print("%-11s %s | saturated operands %d" % (key[0], H.format_line(r), r["saturated_operands"]))
        results.append(r)
    return results


@pytest.mark.parametrize("arch,slot", [("VGG16", 0), ("VGG16", 1), ("ResNet50", 0)])
def test_stress_sweep(arch, slot):
    results = sweep(arch, slot)
    s = HG.report("stress_%s%s" % (arch, "_inflight" if slot else ""), results)
    bad = []
    for r in results:
        assert r["n_proposals"] == r["prop_rows"], r
        assert r["prop_vs_truth"]["n_far"] == 0, r
        assert r["saturated_operands"] == 0, r
        # (2) the reference's rows, where the reference itself is within half the gate of the exact answer
        if r["prop_rows_within_gate"] < r["ref_prop_rows_within_half_gate"] - max(3, r["prop_rows"] // 50):
            bad.append("%s s%d: proposals %d < %d" % (r["kind"], r["seed"], r["prop_rows_within_gate"], r["ref_prop_rows_within_half_gate"]))
        if r["det_rows_within_gate"] < r["ref_det_rows_within_half_gate"] - max(3, r["det_rows"] // 50):
            bad.append("%s s%d: detections %d < %d" % (r["kind"], r["seed"], r["det_rows_within_gate"], r["ref_det_rows_within_half_gate"]))
    # (1) the admission criterion
    for name, ours, ref in (("proposals", s["prop_vs_truth"], s["ref_prop_vs_truth"]), ("detections", s["det_vs_truth"], s["ref_det_vs_truth"])):
        for q in ("median", "p95"):
            if not ours[q] <= K_STRESS[arch] * ref[q]:
                bad.append("%s %s vs truth x%.3f > K %.2f" % (name, q, ours[q] / ref[q], K_STRESS[arch]))
    assert not bad, bad
