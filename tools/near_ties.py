"""
tools/near_ties.py -- how close to a coin flip are the discrete decisions of a golden fixture?  (development aid / evidence for DESIGN section 4)

  python tools/near_ties.py [--fixture vgg16_600x1000_s0] [--eps 1e-5] [--ours gpurun_out/props_all_x3.npy]

Runs the oracle (CPU restatement of the reference) on the fixture's image and lists, for the RPN stage that produces the 300 proposals,
every decision whose margin is below --eps:
  * NMS: a candidate's largest IoU with an earlier KEPT candidate against the 0.7 threshold (kept: 0.7 - IoU; suppressed: IoU - 0.7);
  * order: two candidates adjacent in score order whose scores differ by less than eps AND whose boxes overlap by more than 0.7 (their
    order decides which one survives);
  * filter: height or width within eps px of the 16 px minimum.
With --ours (an (N, 4) .npy of proposals produced on the GPU, tools/dump_props.py) it also reports whether the rows are the same
proposals in the same order and how far their coordinates are from the golden ones (the 1e-3 px gate of the tests is 1.7e-6 of a
600 px box side: float32 noise of a 14-layer network sits right at it).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from fasterrcnn_amd import synthetic  # noqa: E402
from oracle import frcnn_oracle as O  # noqa: E402


def iou_row(b, r):
    d0 = np.maximum(np.minimum(b[2], r[:, 2]) - np.maximum(b[0], r[:, 0]), 0)
    d1 = np.maximum(np.minimum(b[3], r[:, 3]) - np.maximum(b[1], r[:, 1]), 0)
    inter = d0 * d1
    a = (b[2] - b[0]) * (b[3] - b[1])
    ar = (r[:, 2] - r[:, 0]) * (r[:, 3] - r[:, 1])
    return inter / (a + ar - inter)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fixture", default="vgg16_600x1000_s0")
    ap.add_argument("--eps", type=float, default=1e-5)
    ap.add_argument("--ours", default="")
    args = ap.parse_args()
    g = np.load(os.path.join(ROOT, "tests", "golden", args.fixture + ".npz"))
    h, w = int(g["height"]), int(g["width"])
    allow_edge = "noedge" not in args.fixture
    img = synthetic.image(int(g["seed"]), h, w).unsqueeze(0)
    sd = synthetic.vgg16_state_dict(1234)
    detail = {}
    props, classes, deltas = O.forward(sd, img, allow_edge_proposals=allow_edge, detail=detail)
    assert np.array_equal(props.numpy(), g["proposals"]), "the oracle on this host does not reproduce the committed fixture bit for bit"
    cand = detail["candidates"].numpy().astype(np.float64)
    sc = detail["candidate_scores"].numpy().astype(np.float64)
    order = np.argsort(-sc, kind="stable")
    b, s = cand[order], sc[order]
    kept, fragile = [], []
    for j in range(len(b)):
        if len(kept) >= 300:
            break
        m = 0.0
        if kept:
            m = float(iou_row(b[j], b[np.asarray(kept)]).max())
        is_kept = not (m > np.float64(np.float32(0.7)))
        margin = abs(m - float(np.float32(0.7)))
        if margin < args.eps:
            fragile.append(("nms", j, "kept" if is_kept else "suppressed", margin))
        if is_kept:
            kept.append(j)
    last = j
    for j in range(1, last):
        if s[j - 1] - s[j] < args.eps and float(iou_row(b[j], b[j - 1:j])[0]) > 0.7:
            fragile.append(("order", j, "score gap %.3g, IoU %.4f" % (s[j - 1] - s[j], float(iou_row(b[j], b[j - 1:j])[0])), s[j - 1] - s[j]))
    hh, ww = b[:last, 2] - b[:last, 0], b[:last, 3] - b[:last, 1]
    for j in np.nonzero((np.abs(hh - 16) < 1e-3) | (np.abs(ww - 16) < 1e-3))[0]:
        fragile.append(("filter", int(j), "h %.5f w %.5f" % (hh[j], ww[j]), float(min(abs(hh[j] - 16), abs(ww[j] - 16)))))
    print("%s: %d candidates examined for the 300 proposals; decisions with a margin below %.0e: %d" % (args.fixture, last, args.eps, len(fragile)))
    for kind, j, what, margin in fragile:
        print("  %-6s candidate #%d (score %.9f, box %s): %s, margin %.3g" % (kind, j, s[j], np.round(b[j], 3).tolist(), what, margin))
    # margins of ALL NMS decisions: how rare is a near-tie?
    margins = []
    kept2 = []
    for j in range(last):
        m = float(iou_row(b[j], b[np.asarray(kept2)]).max()) if kept2 else 0.0
        margins.append(abs(m - 0.7))
        if not m > 0.7:
            kept2.append(j)
    margins = np.sort(np.asarray(margins))
    print("  smallest NMS margins of the %d decisions: %s" % (last, ", ".join("%.2e" % v for v in margins[:6])))
    if args.ours:
        ours = np.load(args.ours).astype(np.float64)
        ref = g["proposals"].astype(np.float64)
        d = np.abs(ours[:, None, :] - ref[None, :, :]).max(axis=2)
        jn = d.argmin(axis=0)
        e = d[jn, np.arange(len(ref))]
        same_rows = ours.shape == ref.shape and bool((jn == np.arange(len(ref))).all())
        print("  ours: %d of %d golden proposals within 1e-3 px; same rows in the same order: %s; coordinate error max %.3g px, median %.3g px" % (
            int((e <= 1e-3).sum()), len(ref), same_rows, e.max(), np.median(e)))
        for k in np.argsort(-e)[:4]:
            size = max(ref[k][2] - ref[k][0], ref[k][3] - ref[k][1])
            print("    golden proposal %3d %s: error %.3g px = %.2g of its %d px side" % (k, np.round(ref[k], 3).tolist(), e[k], e[k] / size, size))


if __name__ == "__main__":
    main()
