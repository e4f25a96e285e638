"""
tests/holdout_lib.py -- the measurements of the held-out parity sweep, shared by tests/test_holdout_gpu.py and tools/holdout_report.py.

For one held-out fixture (tests/golden/holdout/*.npz, written by oracle/make_holdout.py from the imported reference and the float64
truth) and one run of the HIP path on the same seeded inputs:

  vs the REFERENCE (north_star's bar): row i of ours against row i of the reference -- same proposals in the same ORDER -- as the
      fraction of rows within 1e-3 px; detections per class, in the reference's (score) order;
  vs the float64 TRUTH (the yardstick): the distance of every row of ours from the float64 decode of the anchor it came from
      (oracle/f64_truth.py: proposal_errors / detection_errors), summarised as median / p95 / max, next to the same numbers of the
      reference's own float32 run stored in the fixture.
"""
import glob
import os

import numpy as np
import torch

from fasterrcnn_amd import synthetic
from oracle import f64_truth as T

HOLDOUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "holdout")
STRESS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stress")     # oracle/make_stress.py
GATE = 1e-3              # north_star: boxes within 1e-3 px of the PyTorch reference


def cases(arch):
    return sorted(glob.glob(os.path.join(HOLDOUT, "%s_*.npz" % arch.lower())))


def stress_cases(arch):
    return sorted(glob.glob(os.path.join(STRESS, "%s_*.npz" % arch.lower())))


def build_model(arch, weights_seed, kind=None):
    """kind: a stress kind (synthetic.STRESS_KINDS) selects that recipe's weights; None = the standard recipe"""
    from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
    if arch == "VGG16":
        from fasterrcnn_amd.models.vgg16 import VGG16Backbone
        m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
        sd = synthetic.vgg16_state_dict(weights_seed) if kind is None else synthetic.stress_vgg16_state_dict(weights_seed, kind)
    else:
        from fasterrcnn_amd.models import resnet
        m = FasterRCNNModel(num_classes=21, backbone=resnet.ResNetBackbone(getattr(resnet.Architecture, arch)))
        sd = synthetic.resnet_state_dict(weights_seed, arch) if kind is None else synthetic.stress_resnet_state_dict(weights_seed, kind, arch)
    m.load_state_dict(sd, strict=True)
    return m.cuda().eval()


def flatten_detections(d):
    rows = [np.hstack([np.full((v.shape[0], 1), float(c)), v]) for c, v in sorted(d.items()) if v.shape[0]]
    return np.vstack(rows) if rows else np.zeros((0, 6))


def rowwise(ours, ref):
    """max |coordinate difference| of row i of ours against row i of ref (inf for rows one side lacks)."""
    n = max(len(ours), len(ref))
    err = np.full(n, np.inf)
    m = min(len(ours), len(ref))
    if m:
        err[:m] = np.abs(np.asarray(ours, dtype=np.float64)[:m, :4] - np.asarray(ref, dtype=np.float64)[:m, :4]).max(axis=1)
    return err


def measure(model, g, slot=0):
    """Runs the HIP path on fixture `g`'s inputs; returns a dict of plain numbers / small arrays (see the module docstring).
    slot: 0 = forward / predict (one image at a time); > 0 = that in-flight slot of predict_async, i.e. the arithmetic table of the
    throughput configuration (VGG-16: the 512-channel f32x3 layers in the one-launch form, FasterRCNNModel.layer_tables)."""
    arch = str(g["arch"])
    seed, h, w = int(g["seed"]), int(g["height"]), int(g["width"])
    if "kind" in g:                                       # a stress fixture (oracle/make_stress.py)
        img = (synthetic.stress_image if arch == "VGG16" else synthetic.stress_image_rgb)(seed, str(g["kind"]), h, w).unsqueeze(0).cuda()
    else:
        img = (synthetic.image if arch == "VGG16" else synthetic.image_rgb)(seed, h, w).unsqueeze(0).cuda()
    with torch.no_grad():
        props, classes, deltas = model._enqueue(img, None, None, None, slot).result()
    ctx = model.context(slot)
    fh, fw = (h // 16, w // 16) if arch == "VGG16" else (-(-h // 16), -(-w // 16))
    fm = ctx.tensor(0).cpu().reshape(fh, fw, -1).permute(2, 0, 1).numpy()
    scores = ctx.tensor(2).cpu().numpy()
    det = flatten_detections(model.predict_async(img, float(g["score_threshold"]), slot).result())
    ours = props.cpu().numpy()
    out = {"arch": arch, "seed": seed, "weights_seed": int(g["weights_seed"]), "kind": str(g["kind"]) if "kind" in g else "", "n_proposals": int(ours.shape[0]),
           "n_detections": int(det.shape[0]), "n_ref_detections": int(g["ref_detections"].shape[0])}
    # --- continuous tensors against the truth
    out["fm_err"] = float(np.abs(fm[::128, ::2, ::2].astype(np.float64) - g["truth_fm_sample"]).max() / float(g["truth_fm_scale"]))
    out["ref_fm_err"] = float(g["ref_fm_sample_err"])
    out["obj_err"] = float(np.abs(scores[::16].astype(np.float64) - g["truth_scores_sample"]).max())
    out["ref_obj_err"] = float(g["ref_scores_sample_err"])
    # --- against the reference: same rows in the same order
    r_err = rowwise(ours, g["ref_proposals"])
    out["prop_rows"] = int(len(r_err))
    out["prop_rows_within_gate"] = int((r_err <= GATE).sum())
    out["prop_row_err_max"] = float(r_err.max()) if len(r_err) else 0.0
    out["order_identical"] = bool(len(r_err) and np.isfinite(r_err).all() and r_err.max() <= 0.05)   # every row the reference's row
    # the same as a SET (nearest row of ours for every reference row): one flipped NMS decision shifts every later row index by one
    s_err, _ = T.proposal_errors(g["ref_proposals"], ours) if len(ours) else (np.full(len(g["ref_proposals"]), np.inf), None)
    out["prop_rows_matched_within_gate"] = int((s_err <= GATE).sum())
    ref_det = g["ref_detections"]
    d_ok = d_set = 0
    d_rows = len(ref_det)
    for c in np.unique(ref_det[:, 0]) if d_rows else []:
        r = ref_det[ref_det[:, 0] == c]
        o = det[det[:, 0] == c]
        e = rowwise(o[:, 1:5], r[:, 1:5])[: len(r)]
        s = np.full(len(r), np.inf)
        m = min(len(o), len(r))
        s[:m] = np.abs(o[:m, 5] - r[:m, 5])
        d_ok += int(((e <= GATE) & (s <= 1e-4)).sum())
        if len(o):
            dd = np.abs(o[None, :, 1:5] - r[:, None, 1:5]).max(axis=2)
            jn = dd.argmin(axis=1)
            d_set += int(((dd[np.arange(len(r)), jn] <= GATE) & (np.abs(o[jn, 5] - r[:, 5]) <= 1e-4)).sum())
    out["det_rows"] = int(d_rows)
    out["det_rows_within_gate"] = int(d_ok)
    out["det_rows_matched_within_gate"] = int(d_set)
    out["det_extra_rows"] = int(max(0, det.shape[0] - d_rows))
    # --- against the truth
    p_err, p_near = T.proposal_errors(ours, g["truth_cand_boxes"])
    out["prop_vs_truth"] = T.summarize(p_err)
    out["ref_prop_vs_truth"] = T.summarize(g["ref_prop_err"])
    out["same_candidate_rows"] = int((p_near[: len(g["ref_prop_candidate"])] == g["ref_prop_candidate"][: len(p_near)]).sum())
    d_err, s_err = T.detection_errors(det, g["truth_detections"])
    out["det_vs_truth"] = T.summarize(d_err)
    out["ref_det_vs_truth"] = T.summarize(g["ref_det_err"])
    return out


def pooled(results, key):
    """Pooled statistics over the cases: medians of the per-case medians / p95s, the worst max, summed counts."""
    s = [r[key] for r in results]
    return {"median": float(np.median([x["median"] for x in s])), "p95": float(np.median([x["p95"] for x in s])),
            "max": float(np.max([x["max"] for x in s])), "n": int(np.sum([x["n"] for x in s])),
            "n_far": int(np.sum([x["n_far"] for x in s])), "beyond_gate": int(np.sum([x["beyond_gate"] for x in s]))}


def format_line(r):
    return ("%-9s s%-3d w%-4d | vs REF rows<=1e-3: prop %3d/%3d (as a set %3d; order %s; max %.2e) det %3d/%3d (set %3d, +%d) | vs TRUTH prop med %.2e p95 %.2e max %.2e far %d "
            "[ref %.2e %.2e %.2e] det med %.2e p95 %.2e [ref %.2e %.2e] | fm %.2e [ref %.2e] obj %.2e [ref %.2e]" % (
                r["arch"], r["seed"], r["weights_seed"], r["prop_rows_within_gate"], r["prop_rows"], r["prop_rows_matched_within_gate"],
                "same" if r["order_identical"] else "DIFFERS", r["prop_row_err_max"],
                r["det_rows_within_gate"], r["det_rows"], r["det_rows_matched_within_gate"], r["det_extra_rows"],
                r["prop_vs_truth"]["median"], r["prop_vs_truth"]["p95"], r["prop_vs_truth"]["max"], r["prop_vs_truth"]["n_far"],
                r["ref_prop_vs_truth"]["median"], r["ref_prop_vs_truth"]["p95"], r["ref_prop_vs_truth"]["max"],
                r["det_vs_truth"]["median"], r["det_vs_truth"]["p95"], r["ref_det_vs_truth"]["median"], r["ref_det_vs_truth"]["p95"],
                r["fm_err"], r["ref_fm_err"], r["obj_err"], r["ref_obj_err"]))
