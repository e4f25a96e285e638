"""
oracle/f64_truth.py -- TEST INFRASTRUCTURE ONLY.

A float64 evaluation of the reference's inference path: the yardstick every float32 implementation of it -- the reference's own
torch-CPU run, oracle/frcnn_oracle.py, and the HIP path -- is a perturbation of.  The parity bar of BASELINE.json's north_star is
"boxes within 1e-3 px of the PyTorch reference"; whether a given float32 arithmetic can hold that on a given image depends on how
far float32 noise moves a 600 px box at all, and that is what this file measures: it is what makes the gate "ours-vs-truth <=
k x reference-vs-truth" expressible (tests/test_holdout_gpu.py, tools/holdout_report.py).

What is restated in float64 (paths relative to /root/reference/pytorch/FasterRCNN/):
  models/faster_rcnn.py:80-132   forward (stage 1 -> 2 -> 3)
  models/vgg16.py:60-98, models/resnet.py:38-46,79-81,109-118   the feature extractors (through oracle.frcnn_oracle's functional
                                 forms, which are dtype-generic: they are handed float64 weights and a float64 image)
  models/rpn.py:88-153           3x3 trunk + two 1x1 heads, sigmoid, decode, top-N, clip, >= 16 px, NMS 0.7
  models/math_utils.py:99-128    delta -> box decode (means 0, stds 1)
  models/detector.py:65-80       RoIPool 7x7 @ 1/16, fc1 / fc2 (or ResNet layer4 + mean), classifier softmax, regressor
  models/faster_rcnn.py:175-224  predict(): decode with stds [.1,.1,.2,.2], clip, score > thr, per-class NMS 0.3

Conventions: weights and image are the float32 values of the workload converted to float64 (exact), so the truth is the exact
network function of the SAME inputs up to float64 rounding (~1e-13 of the largest activation).  Anchors are the float32 anchor map
(anchors.py:118-135 rounds them once to float32; every implementation consumes those values).  The discrete steps (top-N order,
>= 16 px filter, NMS, RoIPool's bin rounding, score threshold) are taken on the float64 values; RoIPool's bin edges are computed
from the float32 rounding of the float64 proposal, in float32, exactly as torchvision would be handed them (the bins are integers;
a proposal that sits within 1e-5 px of a rounding boundary may land in a different bin in a float32 run -- such rows show up as
outliers of the detector comparison and are counted, not hidden).
"""
import math

import numpy as np
import torch as t
from torch.nn import functional as F

from oracle import frcnn_oracle as O

_S2 = "_stage2_region_proposal_network."
_S3 = "_stage3_detector_network."


def to_f64(sd):
    return {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}


def decode_boxes(deltas, anchors):
    """math_utils.py:99-128 with means 0 / stds 1, in the dtype of `deltas` (anchors are promoted)."""
    a = anchors.to(deltas.dtype)
    center = a[:, 2:4] * deltas[:, 0:2] + a[:, 0:2]
    size = a[:, 2:4] * t.exp(deltas[:, 2:4])
    return t.cat([center - 0.5 * size, center + 0.5 * size], dim=1)


def clip_boxes(boxes, image_shape):
    """rpn.py:135-137 (clip to [0, H] x [0, W], not H-1 / W-1)."""
    out = boxes.clone()
    out[:, 0:2] = t.clamp(out[:, 0:2], min=0)
    out[:, 2] = t.clamp(out[:, 2], max=image_shape[1])
    out[:, 3] = t.clamp(out[:, 3], max=image_shape[2])
    return out


def roi_pool(fm, rois_xyxy, output_size=7, spatial_scale=1.0 / 16.0):
    """torchvision RoIPool on a float64 map: O.roi_pool's bin arithmetic (float32, from float32 RoIs), the maximum in fm's dtype."""
    fm = np.asarray(fm)
    assert fm.shape[0] == 1
    fm = fm[0]
    c, h, w = fm.shape
    rois = np.asarray(rois_xyxy, dtype=np.float32)
    out = np.zeros((rois.shape[0], c, output_size, output_size), dtype=fm.dtype)
    scale = np.float32(spatial_scale)
    for r in range(rois.shape[0]):
        rs_w = int(O._c_round(rois[r, 1] * scale)); rs_h = int(O._c_round(rois[r, 2] * scale))
        re_w = int(O._c_round(rois[r, 3] * scale)); re_h = int(O._c_round(rois[r, 4] * scale))
        roi_w = max(re_w - rs_w + 1, 1); roi_h = max(re_h - rs_h + 1, 1)
        bin_h = np.float32(roi_h) / np.float32(output_size)
        bin_w = np.float32(roi_w) / np.float32(output_size)
        for ph in range(output_size):
            hs = min(max(int(np.floor(np.float32(ph) * bin_h)) + rs_h, 0), h)
            he = min(max(int(np.ceil(np.float32(ph + 1) * bin_h)) + rs_h, 0), h)
            for pw in range(output_size):
                ws = min(max(int(np.floor(np.float32(pw) * bin_w)) + rs_w, 0), w)
                we = min(max(int(np.ceil(np.float32(pw + 1) * bin_w)) + rs_w, 0), w)
                if he > hs and we > ws:
                    out[r, :, ph, pw] = fm[:, hs:he, ws:we].max(axis=(1, 2))
    return out


def detections(proposals, classes, box_deltas, image_height, image_width, score_threshold):
    """faster_rcnn.py:175-224 on float64 inputs (O.detections is the same with the reference's float32 inputs)."""
    proposals = np.asarray(proposals, dtype=np.float64)
    classes = np.asarray(classes, dtype=np.float64)
    box_deltas = np.asarray(box_deltas, dtype=np.float64)
    anchors = np.empty(proposals.shape)
    anchors[:, 0] = 0.5 * (proposals[:, 0] + proposals[:, 2])
    anchors[:, 1] = 0.5 * (proposals[:, 1] + proposals[:, 3])
    anchors[:, 2:4] = proposals[:, 2:4] - proposals[:, 0:2]
    result = {}
    for c in range(1, classes.shape[1]):
        j = (c - 1) * 4
        boxes = O.convert_deltas_to_boxes(box_deltas[:, j:j + 4], anchors, [0, 0, 0, 0], [0.1, 0.1, 0.2, 0.2])
        boxes[:, 0::2] = np.clip(boxes[:, 0::2], 0, image_height - 1)
        boxes[:, 1::2] = np.clip(boxes[:, 1::2], 0, image_width - 1)
        s = classes[:, c]
        sel = np.where(s > score_threshold)[0]
        boxes, s = boxes[sel], s[sel]
        keep = O.nms(boxes, s, 0.3)
        result[c] = np.hstack([boxes[keep], s[keep][:, None]]) if keep.size else np.zeros((0, 5))
    return result


def forward(sd, image, allow_edge_proposals=True, pre_nms=6000, post_nms=300, score_threshold=0.05, sd64=None):
    """
    The whole path in float64.  `sd` / `image`: the float32 workload (converted here).  Returns a dict:
      feature_map (1,C,H,W) f64, scores (A,) f64, decoded (A,4) f64 (unclipped), clipped (A,4) f64,
      sorted_idx (<= pre_nms,) int64 anchor indices in score order, prop_idx (<= post_nms,) the anchors of the final proposals,
      proposals (n,4) f64, classes (n,K) f64, box_deltas (n,4(K-1)) f64, detections {class: (m,5) f64}
    """
    assert image.shape[0] == 1
    sd64 = sd64 if sd64 is not None else to_f64(sd)
    x = image.double()
    image_shape = tuple(image.shape[1:])
    resnet = O.is_resnet(sd)
    with t.no_grad():
        if resnet:
            fshape = (1024, math.ceil(image_shape[1] / 16), math.ceil(image_shape[2] / 16))
        else:
            fshape = (512, image_shape[1] // 16, image_shape[2] // 16)
        anchor_map, anchor_valid_map = O.generate_anchor_maps(image_shape, fshape, 16)
        fm = O.resnet_features(sd64, x) if resnet else O.vgg16_features(sd64, x)
        y = F.relu(F.conv2d(fm, sd64[_S2 + "_rpn_conv1.weight"], sd64[_S2 + "_rpn_conv1.bias"], padding=1))
        score_map = t.sigmoid(F.conv2d(y, sd64[_S2 + "_rpn_class.weight"], sd64[_S2 + "_rpn_class.bias"]))
        delta_map = F.conv2d(y, sd64[_S2 + "_rpn_boxes.weight"], sd64[_S2 + "_rpn_boxes.bias"])
        scores_all = score_map.permute(0, 2, 3, 1).reshape(-1)
        deltas_all = delta_map.permute(0, 2, 3, 1).reshape(-1, 4)
        anchors_all = t.from_numpy(np.ascontiguousarray(anchor_map)).reshape(-1, 4)
        decoded = decode_boxes(deltas_all, anchors_all)
        clipped = clip_boxes(decoded, image_shape)
        flat = t.arange(scores_all.shape[0])
        if not allow_edge_proposals:
            flat = t.nonzero(t.from_numpy(np.ascontiguousarray(anchor_valid_map)).reshape(-1) > 0).reshape(-1)
        order = t.argsort(scores_all[flat], stable=True).flip(dims=(0,))[0:pre_nms]
        top = flat[order]
        cand = clipped[top]
        big = t.where(((cand[:, 2] - cand[:, 0]) >= 16) & ((cand[:, 3] - cand[:, 1]) >= 16))[0]
        keep = O.nms(cand[big].numpy(), scores_all[top][big].numpy(), 0.7)[0:post_nms]
        prop_idx = top[big][t.from_numpy(keep)]
        proposals = clipped[prop_idx]
        # stage 3 on the float64 proposals
        props32 = proposals.numpy().astype(np.float32)
        rois = np.zeros((props32.shape[0], 5), dtype=np.float32)
        rois[:, 1:] = props32[:, [1, 0, 3, 2]]
        pooled = t.from_numpy(roi_pool(fm.numpy(), rois, 7, 1.0 / 16.0))
        v = O.resnet_pool_to_feature_vector(sd64, pooled) if resnet else O.pool_to_feature_vector(sd64, pooled)
        logits = F.linear(v, sd64[_S3 + "_classifier.weight"], sd64[_S3 + "_classifier.bias"])
        classes = F.softmax(logits, dim=1)
        deltas = F.linear(v, sd64[_S3 + "_regressor.weight"], sd64[_S3 + "_regressor.bias"])
    det = detections(proposals.numpy(), classes.numpy(), deltas.numpy(), image.shape[2], image.shape[3], score_threshold)
    return {"feature_map": fm, "scores": scores_all, "decoded": decoded, "clipped": clipped, "sorted_idx": top.numpy(),
            "prop_idx": prop_idx.numpy(), "proposals": proposals.numpy(), "classes": classes.numpy(),
            "box_deltas": deltas.numpy(), "detections": det, "rpn_deltas": deltas_all}


# ------------------------------------------------------------------------------------------------
# distance of a float32 run's outputs from the truth
# ------------------------------------------------------------------------------------------------
def proposal_errors(proposals, truth_boxes, chunk=64):
    """
    For every proposal row of a float32 run: the L-infinity distance (px) to the NEAREST truth box among `truth_boxes` (the float64
    decoded + clipped boxes of the candidate anchors), and that box's index.  No ordering or matching assumption: a proposal is the
    decode of ONE anchor, and its error is its distance from the float64 decode of that anchor; a row that is not the decode of any
    candidate (a bug) is far from all of them.
    """
    p = np.asarray(proposals, dtype=np.float64)
    tb = np.asarray(truth_boxes, dtype=np.float64)
    err = np.empty(p.shape[0])
    idx = np.empty(p.shape[0], dtype=np.int64)
    for s in range(0, p.shape[0], chunk):
        d = np.abs(p[s:s + chunk, None, :] - tb[None, :, :]).max(axis=2)
        idx[s:s + chunk] = d.argmin(axis=1)
        err[s:s + chunk] = d[np.arange(d.shape[0]), idx[s:s + chunk]]
    return err, idx


def detection_errors(det_rows, truth_rows):
    """
    det_rows / truth_rows: (n,6) [class, y1, x1, y2, x2, score].  For every row of the run: L-infinity box distance (px) and score
    distance to the nearest truth detection OF THE SAME CLASS (inf when the truth has none of that class).
    """
    d = np.asarray(det_rows, dtype=np.float64).reshape(-1, 6)
    tr = np.asarray(truth_rows, dtype=np.float64).reshape(-1, 6)
    box_err = np.full(d.shape[0], np.inf)
    score_err = np.full(d.shape[0], np.inf)
    for i in range(d.shape[0]):
        same = tr[tr[:, 0] == d[i, 0]]
        if same.shape[0]:
            e = np.abs(same[:, 1:5] - d[i, 1:5]).max(axis=1)
            j = int(e.argmin())
            box_err[i] = e[j]
            score_err[i] = abs(same[j, 5] - d[i, 5])
    return box_err, score_err


def summarize(err, gate=1e-3):
    """median / p95 / max of the finite errors + how many rows lie beyond `gate` px (decision flips and gross misses)."""
    e = np.asarray(err, dtype=np.float64)
    fin = e[np.isfinite(e) & (e <= 0.5)]
    return {"n": int(e.size), "n_far": int(e.size - fin.size),
            "median": float(np.median(fin)) if fin.size else float("nan"),
            "p95": float(np.percentile(fin, 95)) if fin.size else float("nan"),
            "max": float(fin.max()) if fin.size else float("nan"),
            "beyond_gate": int((fin > gate).sum())}
