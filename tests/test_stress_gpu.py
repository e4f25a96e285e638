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

Asserted per architecture, pooled over its cases, with the constants of the held-out sweep: ours-vs-truth <= K_TRUTH x reference-vs-truth
and the fractions of the reference's rows reproduced within 1e-3 px; per case: as many proposals as the reference, no row that is not the
decode of a candidate anchor, and ZERO saturated operands in the per-tensor-scaled ResNet-50 backbone (frcnn_x3_saturation_count).
"""
import numpy as np
import pytest

import holdout_lib as H
import test_holdout_gpu as HG

pytestmark = pytest.mark.gpu

MIN_CASES = {"VGG16": 8, "ResNet50": 4}


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
        print("%-11s %s | saturated operands %d" % (key[0], H.format_line(r), r["saturated_operands"]))
        results.append(r)
    return results


@pytest.mark.parametrize("arch,slot", [("VGG16", 0), ("VGG16", 1), ("ResNet50", 0)])
def test_stress_sweep(arch, slot):
    results = sweep(arch, slot)
    s = HG.report("stress_%s%s" % (arch, "_inflight" if slot else ""), results)
    for r in results:
        assert r["n_proposals"] == r["prop_rows"], r
        assert r["prop_vs_truth"]["n_far"] == 0, r
        assert r["saturated_operands"] == 0, r
    bad = HG.violations(arch, s)
    assert not bad, bad
