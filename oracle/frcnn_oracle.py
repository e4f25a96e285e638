"""
oracle/frcnn_oracle.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement (numpy + torch-CPU) of the reference's inference hot path, used ONLY by tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg as the checker / CPU baseline.  Nothing
under fasterrcnn_amd/ imports it; the product path has no CPU fallback.

Every function cites the reference lines it follows (paths relative to
/root/reference/pytorch/FasterRCNN/).  Dense layers use the very torch-CPU ops the reference calls
(F.conv2d, F.linear, F.max_pool2d, softmax, sigmoid, argsort): torch core is importable here.
torchvision is NOT vendored in the reference nor installed (pytorch/requirements.txt pins
torchvision==0.15.0+cu117), so `nms` and `roi_pool` below restate torchvision's published
semantics; the reference has no tests that pin them -> PARITY UNPINNED at that boundary
(DESIGN.md section "Oracle").  Everything else is pinned by oracle/make_golden.py, which imports
the reference itself in the build container and checks this file against it; the captured vectors
live in tests/golden/.
"""
import itertools
import math
from collections import defaultdict

import numpy as np
import torch as t
from torch.nn import functional as F


# ------------------------------------------------------------------------------------------------
# preprocessing  (datasets/image.py:34-101; PIL's 8-bit BILINEAR resampler restated)
# ------------------------------------------------------------------------------------------------
def _pil_coeffs(in_size, out_size):
    """libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle) filter."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds, kk = [], []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)]
        ww = sum(w[:0], 0.0)
        for v in w:
            ww += v
        w = [(v / ww if ww != 0.0 else v) for v in w] + [0.0] * (ksize - xmax)
        kk.append([int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22)) for v in w])
        bounds.append((xmin, xmax))
    return bounds, np.asarray(kk, dtype=np.int64)


def pil_resize_bilinear(rgb, out_h, out_w):
    """Image.resize((out_w, out_h), Image.BILINEAR) of a uint8 (H,W,3) image: horizontal pass, then vertical."""
    src = np.asarray(rgb, dtype=np.int64)
    h, w, _ = src.shape
    bx, kx = _pil_coeffs(w, out_w)
    tmp = np.empty((h, out_w, 3), dtype=np.int64)
    for xx, (x0, n) in enumerate(bx):
        acc = (src[:, x0:x0 + n, :] * kx[xx, :n][None, :, None]).sum(axis=1) + (1 << 21)
        tmp[:, xx, :] = np.clip(acc >> 22, 0, 255)
    by, ky = _pil_coeffs(h, out_h)
    out = np.empty((out_h, out_w, 3), dtype=np.int64)
    for yy, (y0, n) in enumerate(by):
        acc = (tmp[y0:y0 + n, :, :] * ky[yy, :n][:, None, None]).sum(axis=0) + (1 << 21)
        out[yy] = np.clip(acc >> 22, 0, 255)
    return out.astype(np.uint8)


def preprocess_image(rgb, bgr, scaling, means, stds, min_dimension_pixels=None, horizontal_flip=False):
    """image.py:89-100 + :43-57 from the decoded uint8 RGB array to the float32 (3,h,w) tensor."""
    rgb = np.asarray(rgb, dtype=np.uint8)
    h0, w0 = rgb.shape[0], rgb.shape[1]
    if horizontal_flip:
        rgb = rgb[:, ::-1, :]
    if min_dimension_pixels is not None:
        sf = min_dimension_pixels / h0 if w0 > h0 else min_dimension_pixels / w0          # :34-41
        rgb = pil_resize_bilinear(rgb, int(h0 * sf), int(w0 * sf))                       # :94-96
    data = rgb.astype(np.float32)
    if bgr:
        data = data[:, :, ::-1]
    data = data.copy()
    for c in range(3):
        data[:, :, c] *= scaling
    for c in range(3):
        data[:, :, c] = (data[:, :, c] - means[c]) / stds[c]
    return data.transpose([2, 0, 1]).copy()


# ------------------------------------------------------------------------------------------------
# anchors  (models/anchors.py:25-135)
# ------------------------------------------------------------------------------------------------
def anchor_sizes():
    """(9,2) (height,width); k = area-major, aspect-minor (anchors.py:33-41)."""
    out = []
    for area, aspect in itertools.product([128 * 128, 256 * 256, 512 * 512], [0.5, 1.0, 2.0]):
        root = math.sqrt(area / aspect)
        out.append((aspect * root, root))
    return np.array(out, dtype=np.float64)


def generate_anchor_maps(image_shape, feature_map_shape, feature_pixels):
    """
    anchors.py:43-135.  Returns anchor_map (H,W,36) float32 of (cy,cx,h,w) and valid map (H,W,9)
    float32.  The arithmetic order matters for bit-exactness: cell centres are rounded to float32
    (:118) and then combined with the float64 template; a single cast to float32 ends it (:135).
    """
    sizes = anchor_sizes()
    height, width = int(feature_map_shape[-2]), int(feature_map_shape[-1])
    ys = (np.arange(height) * feature_pixels + 0.5 * feature_pixels).astype(np.float32)     # :105,:118
    xs = (np.arange(width) * feature_pixels + 0.5 * feature_pixels).astype(np.float32)
    shape = (height, width, 9)
    cy = np.broadcast_to(ys[:, None, None].astype(np.float64), shape)
    cx = np.broadcast_to(xs[None, :, None].astype(np.float64), shape)
    half_h = 0.5 * sizes[None, None, :, 0]
    half_w = 0.5 * sizes[None, None, :, 1]
    y1, x1 = cy + (-half_h), cx + (-half_w)                                                  # :92,:118
    y2, x2 = cy + half_h, cx + half_w                                                        # :93,:118
    image_height, image_width = image_shape[1], image_shape[2]
    valid = (y1 >= 0) & (x1 >= 0) & (y2 <= image_height) & (x2 <= image_width)              # :124-125
    amap = np.stack([0.5 * (y1 + y2), 0.5 * (x1 + x2), y2 - y1, x2 - x1], axis=-1)           # :128-130
    return amap.reshape(height, width, 36).astype(np.float32), valid.astype(np.float32)


def generate_rpn_map(anchor_map, anchor_valid_map, gt_corners, object_iou_threshold=0.7, background_iou_threshold=0.3):
    """
    anchors.py:137-262.  gt_corners: (M,4) float32 (y1,x1,y2,x2).  Returns (rpn_map float32
    (H,W,9,6), object indices (N,3), background indices (M,3)) with the reference's dtype flow:
    anchor corners float32 -> float64 (:186-190), IoU float64 (:199), targets float32 (:244-246).
    """
    height, width, k = anchor_valid_map.shape
    gt = np.asarray(gt_corners)
    am = anchor_map.reshape(-1, 4)
    corners = np.empty(am.shape)                                             # float64
    corners[:, 0:2] = am[:, 0:2] - 0.5 * am[:, 2:4]
    corners[:, 2:4] = am[:, 0:2] + 0.5 * am[:, 2:4]
    n = corners.shape[0]
    tl = np.maximum(corners[:, None, 0:2], gt[None, :, 0:2])                 # math_utils.py:29-37
    br = np.minimum(corners[:, None, 2:4], gt[None, :, 2:4])
    ok = np.all(tl < br, axis=2)
    inter = ok * np.prod(br - tl, axis=2)
    a1 = np.prod(corners[:, 2:4] - corners[:, 0:2], axis=1)
    a2 = np.prod(gt[:, 2:4] - gt[:, 0:2], axis=1)
    ious = inter / (a1[:, None] + a2[None, :] - inter + 1e-7)
    ious[anchor_valid_map.reshape(-1) == 0, :] = -1.0                        # :204
    best = ious.max(axis=1)
    best_m = ious.argmax(axis=1)
    gt_best = ious.max(axis=0)
    top_anchor = np.where(ious == gt_best)[0]                                # :218
    label = np.full(n, -1)
    label[best < background_iou_threshold] = 0
    label[best >= object_iou_threshold] = 1
    label[top_anchor] = 1
    enable = (label >= 0).astype(np.float32)
    label[label < 0] = 0
    centers = 0.5 * (gt[:, 0:2] + gt[:, 2:4])
    sides = gt[:, 2:4] - gt[:, 0:2]
    targets = np.empty((n, 4))
    targets[:, 0:2] = (centers[best_m] - am[:, 0:2]) / am[:, 2:4]
    targets[:, 2:4] = np.log(sides[best_m] / am[:, 2:4])
    rpn_map = np.zeros((height, width, k, 6))
    rpn_map[..., 0] = anchor_valid_map * enable.reshape(height, width, k)
    rpn_map[..., 1] = label.reshape(height, width, k)
    rpn_map[..., 2:6] = targets.reshape(height, width, k, 4)
    coords = np.stack(np.meshgrid(np.arange(height), np.arange(width), np.arange(k), indexing="ij"), axis=-1)
    obj = coords[(rpn_map[..., 1] > 0) & (rpn_map[..., 0] > 0)]
    bg = coords[(rpn_map[..., 1] == 0) & (rpn_map[..., 0] > 0)]
    return rpn_map.astype(np.float32), obj, bg


# ------------------------------------------------------------------------------------------------
# torchvision.ops restated (third-party; see module docstring)
# ------------------------------------------------------------------------------------------------
def nms(boxes, scores, iou_threshold):
    """
    torchvision.ops.nms(boxes[N,4], scores[N], thr) as called at rpn.py:147-151 and
    faster_rcnn.py:216-220: stable score-descending order; a later box j is dropped by an earlier
    kept box i iff inter/(area_i+area_j-inter) > thr, computed in the dtype of `boxes`; no +1, no
    epsilon.  The threshold is a C float inside torchvision's CUDA devIoU, hence float32(thr).
    Returns int64 indices in kept (score-descending) order.
    """
    boxes = np.asarray(boxes)
    scores = np.asarray(scores)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    dt = boxes.dtype
    thr = dt.type(np.float32(iou_threshold))
    order = np.argsort(-scores.astype(np.float64), kind="stable")
    b = boxes[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    zero = dt.type(0)
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if i + 1 >= n:
            break
        r = b[i + 1:]
        d0 = np.maximum(np.minimum(b[i, 2], r[:, 2]) - np.maximum(b[i, 0], r[:, 0]), zero)
        d1 = np.maximum(np.minimum(b[i, 3], r[:, 3]) - np.maximum(b[i, 1], r[:, 1]), zero)
        inter = d0 * d1
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / (area[i] + area[i + 1:] - inter)
        suppressed[i + 1:] |= iou > thr
    return order[np.asarray(keep, dtype=np.int64)]


def _c_round(x):
    """C round(): half away from zero (numpy's round is half-to-even)."""
    return np.where(x >= 0, np.floor(x + np.float32(0.5)), np.ceil(x - np.float32(0.5)))


def roi_pool(feature_map, rois_xyxy, output_size, spatial_scale):
    """
    torchvision.ops.RoIPool(output_size, spatial_scale)(input NCHW (1,C,H,W), rois (K,5) =
    (batch, x1, y1, x2, y2)) as used at detector.py:27,72.  float32 arithmetic throughout:
      start/end = round(coord*scale); size = max(end-start+1, 1); bin = size/out;
      window = [floor(p*bin)+start, ceil((p+1)*bin)+start) clipped to the map; empty -> 0 else max.
    Returns (K, C, out, out) float32.
    """
    fm = np.asarray(feature_map, dtype=np.float32)
    assert fm.shape[0] == 1
    fm = fm[0]
    c, h, w = fm.shape
    rois = np.asarray(rois_xyxy, dtype=np.float32)
    k = rois.shape[0]
    out = np.zeros((k, c, output_size, output_size), dtype=np.float32)
    scale = np.float32(spatial_scale)
    for r in range(k):
        rs_w = int(_c_round(rois[r, 1] * scale)); rs_h = int(_c_round(rois[r, 2] * scale))
        re_w = int(_c_round(rois[r, 3] * scale)); re_h = int(_c_round(rois[r, 4] * scale))
        roi_w = max(re_w - rs_w + 1, 1); roi_h = max(re_h - rs_h + 1, 1)
        bin_h = np.float32(roi_h) / np.float32(output_size)
        bin_w = np.float32(roi_w) / np.float32(output_size)
        for ph in range(output_size):
            hs = int(np.floor(np.float32(ph) * bin_h)) + rs_h
            he = int(np.ceil(np.float32(ph + 1) * bin_h)) + rs_h
            hs = min(max(hs, 0), h); he = min(max(he, 0), h)
            for pw in range(output_size):
                ws = int(np.floor(np.float32(pw) * bin_w)) + rs_w
                we = int(np.ceil(np.float32(pw + 1) * bin_w)) + rs_w
                ws = min(max(ws, 0), w); we = min(max(we, 0), w)
                if he > hs and we > ws:
                    out[r, :, ph, pw] = fm[:, hs:he, ws:we].max(axis=(1, 2))
    return out


def roi_align_weights(h, w, roi_xyxy, output_size, spatial_scale, sampling_ratio=2, aligned=False):
    """
    The sampling plan of torchvision.ops.roi_align for ONE roi (x1, y1, x2, y2) on an h x w map, float32 arithmetic in the
    operation order of torchvision 0.15's roi_align_kernel.cpp (third party, absent here: restated from its published
    algorithm -- parity unpinned, like nms / RoIPool):
      offset = 0.5 if aligned else 0;  start = coord * scale - offset;  size = end - start (clamped to >= 1 unless aligned);
      bin = size / out;  grid = sampling_ratio if > 0 else ceil(size / out);  count = max(grid_h * grid_w, 1);
      sample (ph, pw, iy, ix) at y = start_h + ph * bin_h + (iy + .5) * bin_h / grid_h (x alike);
      bilinear_interpolate: outside [-1, H] x [-1, W] -> 0; clamp to >= 0; low = int(y); at the last row / column low = high = H - 1
      and y = low; weights hy hx, hy lx, ly hx, ly lx.
    Returns (grid_h, grid_w, count, list over (ph, pw) of lists of (y_low, x_low, y_high, x_high, w1, w2, w3, w4)).
    """
    f = np.float32
    scale = f(spatial_scale)
    offset = f(0.5) if aligned else f(0.0)
    x1, y1, x2, y2 = (f(v) for v in roi_xyxy)
    start_w, start_h = x1 * scale - offset, y1 * scale - offset
    end_w, end_h = x2 * scale - offset, y2 * scale - offset
    roi_w, roi_h = end_w - start_w, end_h - start_h
    if not aligned:
        roi_w, roi_h = max(roi_w, f(1.0)), max(roi_h, f(1.0))
    bin_h, bin_w = roi_h / f(output_size), roi_w / f(output_size)
    grid_h = int(sampling_ratio) if sampling_ratio > 0 else int(np.ceil(roi_h / f(output_size)))
    grid_w = int(sampling_ratio) if sampling_ratio > 0 else int(np.ceil(roi_w / f(output_size)))
    count = max(grid_h * grid_w, 1)
    plan = []
    for ph in range(output_size):
        for pw in range(output_size):
            samples = []
            for iy in range(grid_h):
                y = start_h + f(ph) * bin_h + (f(iy) + f(0.5)) * bin_h / f(grid_h)
                for ix in range(grid_w):
                    x = start_w + f(pw) * bin_w + (f(ix) + f(0.5)) * bin_w / f(grid_w)
                    yy, xx = y, x
                    if yy < f(-1.0) or yy > f(h) or xx < f(-1.0) or xx > f(w):
                        continue                                  # contributes 0
                    yy, xx = max(yy, f(0.0)), max(xx, f(0.0))
                    y_low, x_low = int(yy), int(xx)
                    if y_low >= h - 1:
                        y_high = y_low = h - 1
                        yy = f(y_low)
                    else:
                        y_high = y_low + 1
                    if x_low >= w - 1:
                        x_high = x_low = w - 1
                        xx = f(x_low)
                    else:
                        x_high = x_low + 1
                    ly, lx = yy - f(y_low), xx - f(x_low)
                    hy, hx = f(1.0) - ly, f(1.0) - lx
                    samples.append((y_low, x_low, y_high, x_high, hy * hx, hy * lx, ly * hx, ly * lx))
            plan.append(samples)
    return grid_h, grid_w, count, plan


def roi_align(feature_map, rois_xyxy, output_size, spatial_scale, sampling_ratio=2, aligned=False):
    """
    torchvision.ops.roi_align(input NCHW (1,C,H,W), rois (K,5) = (batch, x1, y1, x2, y2), output_size, spatial_scale,
    sampling_ratio, aligned) restated (see roi_align_weights).  float32: per bin the samples' values
    w1 v1 + w2 v2 + w3 v3 + w4 v4 are summed in (iy, ix) order and divided by count.  Returns (K, C, out, out) float32.
    Beyond the reference, which uses RoIPool (models/detector.py:16,27): BASELINE.json's north_star names RoIAlign.
    """
    fm = np.asarray(feature_map, dtype=np.float32)
    assert fm.shape[0] == 1
    fm = fm[0]
    c, h, w = fm.shape
    rois = np.asarray(rois_xyxy, dtype=np.float32)
    k = rois.shape[0]
    out = np.zeros((k, c, output_size, output_size), dtype=np.float32)
    for r in range(k):
        _, _, count, plan = roi_align_weights(h, w, rois[r, 1:5], output_size, spatial_scale, sampling_ratio, aligned)
        for b, samples in enumerate(plan):
            acc = np.zeros((c,), dtype=np.float32)
            for (yl, xl, yh, xh, w1, w2, w3, w4) in samples:
                acc = acc + (w1 * fm[:, yl, xl] + w2 * fm[:, yl, xh] + w3 * fm[:, yh, xl] + w4 * fm[:, yh, xh])
            out[r, :, b // output_size, b % output_size] = acc / np.float32(count)
    return out


def roi_align_backward(grad_out, fm_shape, rois_xyxy, output_size, spatial_scale, sampling_ratio=2, aligned=False):
    """Gradient of roi_align with respect to the feature map (float64 accumulation: the truth the kernel's float32 sum is held to).
    grad_out (K, C, out, out) -> (1, C, H, W)."""
    _, c, h, w = fm_shape
    g = np.asarray(grad_out, dtype=np.float64)
    rois = np.asarray(rois_xyxy, dtype=np.float32)
    d = np.zeros((c, h, w), dtype=np.float64)
    for r in range(rois.shape[0]):
        _, _, count, plan = roi_align_weights(h, w, rois[r, 1:5], output_size, spatial_scale, sampling_ratio, aligned)
        for b, samples in enumerate(plan):
            gb = g[r, :, b // output_size, b % output_size] / count
            for (yl, xl, yh, xh, w1, w2, w3, w4) in samples:
                d[:, yl, xl] += float(w1) * gb
                d[:, yl, xh] += float(w2) * gb
                d[:, yh, xl] += float(w3) * gb
                d[:, yh, xh] += float(w4) * gb
    return d[None]


# ------------------------------------------------------------------------------------------------
# network stages on torch-CPU (the same ATen ops the reference calls)
# ------------------------------------------------------------------------------------------------
_S1 = "_stage1_feature_extractor."
_S2 = "_stage2_region_proposal_network."
_S3 = "_stage3_detector_network."


def vgg16_features(sd, image):
    """vgg16.py:60-98: 13 x (conv3x3 same + ReLU), max-pool 2x2/2 after blocks 1-4.  image (1,3,H,W)."""
    y = image
    for block, convs in ((1, 2), (2, 2), (3, 3), (4, 3), (5, 3)):
        for i in range(1, convs + 1):
            key = "%s_block%d_conv%d" % (_S1, block, i)
            y = F.relu(F.conv2d(y, sd[key + ".weight"], sd[key + ".bias"], stride=1, padding=1))
        if block < 5:
            y = F.max_pool2d(y, kernel_size=2, stride=2)
    return y


# ---- ResNet (models/resnet.py:33-118 over torchvision's v1.5 Bottleneck, restated) ----------------
_RFE = _S1 + "_feature_extractor."
_RL4 = _S3 + "_pool_to_feature_vector._layer4."


def is_resnet(sd):
    return (_RFE + "0.weight") in sd


def _bn(x, sd, prefix):
    """BatchNorm2d in eval mode (resnet.py:58-77,100-107 force it), eps 1e-5."""
    return F.batch_norm(x, sd[prefix + "running_mean"], sd[prefix + "running_var"], sd[prefix + "weight"],
                        sd[prefix + "bias"], False, 0.0, 1e-5)


# conv + frozen BatchNorm of a Bottleneck.  oracle/train_oracle.py swaps this for a function with a bf16 weight gradient while it
# restates the reduced-precision train step (grad_math="bf16"); the forward value is the same.
CONV_BN = None


def _conv_bn(x, sd, wkey, bn_prefix, stride=1, padding=0):
    if CONV_BN is not None:
        return CONV_BN(x, sd, wkey, bn_prefix, stride, padding)
    return _bn(F.conv2d(x, sd[wkey], stride=stride, padding=padding), sd, bn_prefix)


def _bottleneck(x, sd, prefix, stride):
    out = F.relu(_conv_bn(x, sd, prefix + "conv1.weight", prefix + "bn1."))
    out = F.relu(_conv_bn(out, sd, prefix + "conv2.weight", prefix + "bn2.", stride=stride, padding=1))
    out = _conv_bn(out, sd, prefix + "conv3.weight", prefix + "bn3.")
    identity = x
    if (prefix + "downsample.0.weight") in sd:
        identity = _conv_bn(x, sd, prefix + "downsample.0.weight", prefix + "downsample.1.", stride=stride)
    return F.relu(out + identity)


def _layer(x, sd, prefix, first_stride):
    b = 0
    while (prefix + "%d.conv1.weight" % b) in sd:
        x = _bottleneck(x, sd, prefix + "%d." % b, first_stride if b == 0 else 1)
        b += 1
    return x


def resnet_features(sd, image):
    """resnet.py:38-46,79-81: conv1, bn1, relu, maxpool(3,2,1), layer1..layer3 -> (1,1024,ceil(H/16),ceil(W/16))."""
    y = F.relu(_bn(F.conv2d(image, sd[_RFE + "0.weight"], stride=2, padding=3), sd, _RFE + "1."))
    y = F.max_pool2d(y, kernel_size=3, stride=2, padding=1)
    y = _layer(y, sd, _RFE + "4.", 1)
    y = _layer(y, sd, _RFE + "5.", 2)
    return _layer(y, sd, _RFE + "6.", 2)


def resnet_pool_to_feature_vector(sd, rois):
    """resnet.py:109-118: layer4 (stride 2) then `.mean(-1).mean(-1)`."""
    y = _layer(rois, sd, _RL4, 2)
    return y.mean(-1).mean(-1)


def decode_boxes_f32(deltas, anchors):
    """math_utils.py:99-128 with means 0 / stds 1 (rpn.py:118-123), float32 torch ops."""
    d = deltas * t.ones(4) + t.zeros(4)
    center = anchors[:, 2:4] * d[:, 0:2] + anchors[:, 0:2]
    size = anchors[:, 2:4] * t.exp(d[:, 2:4])
    boxes = t.empty(d.shape, dtype=t.float32)
    boxes[:, 0:2] = center - 0.5 * size
    boxes[:, 2:4] = center + 0.5 * size
    return boxes


def proposals_from_maps(score_map, delta_map, image_shape, anchor_map, anchor_valid_map, pre_nms, post_nms,
                        allow_edge_proposals=True, detail=None):
    """
    rpn.py:98-153 from the permuted head outputs: score_map (1,H,W,9) (already sigmoid-ed),
    delta_map (1,H,W,36).  Returns proposals (N,4).
    """
    anchors = t.from_numpy(np.ascontiguousarray(anchor_map)).reshape(-1, 4)
    scores = score_map.reshape(-1)
    deltas = delta_map.reshape(-1, 4)
    if not allow_edge_proposals:                                     # :167-173
        idx = t.from_numpy(np.ascontiguousarray(anchor_valid_map)).reshape(-1) > 0
        flat_index = t.nonzero(idx).reshape(-1)
        anchors, scores, deltas = anchors[idx], scores[idx], deltas[idx]
    else:
        flat_index = t.arange(scores.shape[0])
    proposals = decode_boxes_f32(deltas, anchors)                    # :118-123
    order = t.argsort(scores, stable=True).flip(dims=(0,))           # :129-130 (ties -> higher index first)
    top = order[0:pre_nms]
    proposals = proposals[top]                                       # :131-132
    top_scores = scores[top]
    proposals[:, 0:2] = t.clamp(proposals[:, 0:2], min=0)            # :135-137
    proposals[:, 2] = t.clamp(proposals[:, 2], max=image_shape[1])
    proposals[:, 3] = t.clamp(proposals[:, 3], max=image_shape[2])
    hh = proposals[:, 2] - proposals[:, 0]                           # :140-144
    ww = proposals[:, 3] - proposals[:, 1]
    big = t.where((hh >= 16) & (ww >= 16))[0]
    cand = proposals[big]
    cand_scores = top_scores[big]
    keep = nms(cand.numpy(), cand_scores.numpy(), 0.7)[0:post_nms]   # :147-153
    out = cand[t.from_numpy(keep)]
    if detail is not None:
        detail["scores"] = scores
        detail["decoded"] = decode_boxes_f32(deltas, anchors)
        detail["sorted_idx"] = flat_index[top].numpy().astype(np.int64)
        detail["n_after_filter"] = int(big.shape[0])
        detail["candidates"] = cand
        detail["candidate_scores"] = cand_scores
    return out


def rpn_forward(sd, feature_map, image_shape, anchor_map, anchor_valid_map, pre_nms, post_nms,
                allow_edge_proposals=True, detail=None):
    """
    rpn.py:88-156.  Returns (objectness_score_map (1,H,W,9), box_deltas_map (1,H,W,36), proposals (N,4)).
    `detail`, if a dict, receives the intermediates the parity tests compare.
    """
    y = F.relu(F.conv2d(feature_map, sd[_S2 + "_rpn_conv1.weight"], sd[_S2 + "_rpn_conv1.bias"], padding=1))
    score_map = t.sigmoid(F.conv2d(y, sd[_S2 + "_rpn_class.weight"], sd[_S2 + "_rpn_class.bias"]))
    delta_map = F.conv2d(y, sd[_S2 + "_rpn_boxes.weight"], sd[_S2 + "_rpn_boxes.bias"])
    score_map = score_map.permute(0, 2, 3, 1).contiguous()          # :95-96
    delta_map = delta_map.permute(0, 2, 3, 1).contiguous()
    out = proposals_from_maps(score_map, delta_map, image_shape, anchor_map, anchor_valid_map, pre_nms, post_nms,
                              allow_edge_proposals, detail)
    if detail is not None:
        detail["rpn_trunk"] = y
        detail["delta_map"] = delta_map
    return score_map, delta_map, out


def pool_to_feature_vector(sd, rois):
    """vgg16.py:129-133 (dropout is identity at inference)."""
    x = rois.reshape(rois.shape[0], 512 * 7 * 7)
    p = _S3 + "_pool_to_feature_vector."
    y = F.relu(F.linear(x, sd[p + "_fc1.weight"], sd[p + "_fc1.bias"]))
    return F.relu(F.linear(y, sd[p + "_fc2.weight"], sd[p + "_fc2.bias"]))


def detector_forward(sd, feature_map, proposals, detail=None, roi_pooling="pool", sampling_ratio=2):
    """detector.py:65-80: RoIPool 7x7 @ 1/16 -> fc1, fc2 -> softmax classes, box deltas.
    roi_pooling="align" (beyond the reference): torchvision roi_align semantics instead of RoIPool."""
    props = proposals.numpy() if isinstance(proposals, t.Tensor) else np.asarray(proposals)
    rois = np.zeros((props.shape[0], 5), dtype=np.float32)
    rois[:, 1:] = props[:, [1, 0, 3, 2]]                             # (y1,x1,y2,x2) -> (x1,y1,x2,y2), :69
    if roi_pooling == "align":
        pooled = t.from_numpy(roi_align(feature_map.numpy(), rois, 7, 1.0 / 16.0, sampling_ratio, False))
    else:
        pooled = t.from_numpy(roi_pool(feature_map.numpy(), rois, 7, 1.0 / 16.0))
    y = resnet_pool_to_feature_vector(sd, pooled) if is_resnet(sd) else pool_to_feature_vector(sd, pooled)
    logits = F.linear(y, sd[_S3 + "_classifier.weight"], sd[_S3 + "_classifier.bias"])
    classes = F.softmax(logits, dim=1)
    deltas = F.linear(y, sd[_S3 + "_regressor.weight"], sd[_S3 + "_regressor.bias"])
    if detail is not None:
        detail["pooled"] = pooled
        detail["fc2"] = y
        detail["class_logits"] = logits
    return classes, deltas


def forward(sd, image, anchor_map=None, anchor_valid_map=None, allow_edge_proposals=True,
            pre_nms=6000, post_nms=300, detail=None, roi_pooling="pool", sampling_ratio=2):
    """faster_rcnn.py:80-132.  image: torch float32 (1,3,H,W) on CPU."""
    assert image.shape[0] == 1
    image_shape = tuple(image.shape[1:])
    with t.no_grad():
        if anchor_map is None or anchor_valid_map is None:
            if is_resnet(sd):                                                 # resnet.py:183-185 (ceil)
                fshape = (1024, math.ceil(image_shape[1] / 16), math.ceil(image_shape[2] / 16))
            else:
                fshape = (512, image_shape[1] // 16, image_shape[2] // 16)    # vgg16.py:155-158
            anchor_map, anchor_valid_map = generate_anchor_maps(image_shape, fshape, 16)
        fm = resnet_features(sd, image) if is_resnet(sd) else vgg16_features(sd, image)
        _, _, proposals = rpn_forward(sd, fm, image_shape, anchor_map, anchor_valid_map, pre_nms, post_nms,
                                      allow_edge_proposals, detail)
        classes, deltas = detector_forward(sd, fm, proposals, detail, roi_pooling, sampling_ratio)
    if detail is not None:
        detail["feature_map"] = fm
    return proposals, classes, deltas


def convert_deltas_to_boxes(box_deltas, anchors, box_delta_means, box_delta_stds):
    """math_utils.py:65-97 (numpy; promotes to float64 with list-valued means/stds)."""
    d = box_deltas * np.asarray(box_delta_stds) + np.asarray(box_delta_means)
    center = anchors[:, 2:4] * d[:, 0:2] + anchors[:, 0:2]
    size = anchors[:, 2:4] * np.exp(d[:, 2:4])
    boxes = np.empty(d.shape)
    boxes[:, 0:2] = center - 0.5 * size
    boxes[:, 2:4] = center + 0.5 * size
    return boxes


def detections(proposals, classes, box_deltas, image_height, image_width, score_threshold):
    """
    faster_rcnn.py:175-224 on numpy inputs (float32): float32 anchor conversion stored into float64
    (:180-183), float64 decode with stds [.1,.1,.2,.2] (:190-197), clip (:200-201), score filter
    (:208), per-class nms 0.3 on float64 boxes (:216-220), rows (y1,x1,y2,x2,score) float64 (:221-224).
    """
    proposals = np.asarray(proposals, dtype=np.float32)
    classes = np.asarray(classes, dtype=np.float32)
    box_deltas = np.asarray(box_deltas, dtype=np.float32)
    anchors = np.empty(proposals.shape)
    anchors[:, 0] = 0.5 * (proposals[:, 0] + proposals[:, 2])
    anchors[:, 1] = 0.5 * (proposals[:, 1] + proposals[:, 3])
    anchors[:, 2:4] = proposals[:, 2:4] - proposals[:, 0:2]
    result = {}
    for c in range(1, classes.shape[1]):
        j = (c - 1) * 4
        boxes = convert_deltas_to_boxes(box_deltas[:, j:j + 4], anchors, [0, 0, 0, 0], [0.1, 0.1, 0.2, 0.2])
        boxes[:, 0::2] = np.clip(boxes[:, 0::2], 0, image_height - 1)
        boxes[:, 1::2] = np.clip(boxes[:, 1::2], 0, image_width - 1)
        s = classes[:, c]
        sel = np.where(s > score_threshold)[0]
        boxes, s = boxes[sel], s[sel]
        keep = nms(boxes, s, 0.3)
        result[c] = np.hstack([boxes[keep], s[keep][:, None].astype(np.float64)]) if keep.size else np.zeros((0, 5))
    return result


def predict(sd, image, score_threshold, **kw):
    """faster_rcnn.py:134-226."""
    proposals, classes, deltas = forward(sd, image, **kw)
    return detections(proposals.numpy(), classes.numpy(), deltas.numpy(), image.shape[2], image.shape[3],
                      score_threshold)


# ------------------------------------------------------------------------------------------------
# mAP  (statistics.py:65-214), written as the reference's plain loops
# ------------------------------------------------------------------------------------------------
def iou_pair(a, b):
    """math_utils.py:13-37 for one pair of (y1,x1,y2,x2) boxes."""
    tl = np.maximum(a[0:2], b[0:2])
    br = np.minimum(a[2:4], b[2:4])
    ok = bool(np.all(tl < br))
    inter = (np.prod(br - tl) if ok else 0.0) * 1.0
    ua = np.prod(a[2:4] - a[0:2]) + np.prod(b[2:4] - b[0:2]) - inter
    return inter / (ua + 1e-7)


class MeanAveragePrecision:
    def __init__(self):
        self.preds = defaultdict(list)     # class -> [(score, is_tp)]
        self.gt_count = defaultdict(int)

    def add_image_results(self, scored_boxes_by_class_index, gt_boxes):
        """gt_boxes: list of (class_index, corners).  statistics.py:77-156."""
        for cls, _ in gt_boxes:
            self.gt_count[cls] += 1
        for cls, rows in scored_boxes_by_class_index.items():
            gts = [corners for (c, corners) in gt_boxes if c == cls]
            tp = [False] * len(rows)
            found = [False] * len(gts)
            # statistics.py:99's sort is a no-op (constant key): order stays gt-major, box-minor
            for g in range(len(gts)):
                for bidx in range(len(rows)):
                    if iou_pair(np.asarray(rows[bidx][0:4]), np.asarray(gts[g])) <= 0.5:
                        continue
                    if tp[bidx] or found[g]:
                        continue
                    tp[bidx] = True
                    found[g] = True
            self.preds[cls] += [(rows[i][4], tp[i]) for i in range(len(rows))]

    def average_precision(self, cls):
        """statistics.py:158-197."""
        ranked = sorted(self.preds[cls], key=lambda p: p[0], reverse=True)
        n_pos = self.gt_count[cls]
        recall, precision = [0.0], [0.0]
        tps = fps = 0
        for _, ok in ranked:
            if ok:
                tps += 1
            else:
                fps += 1
            recall.append(tps / n_pos)
            precision.append(tps / (tps + fps))
        recall.append(1.0)
        precision.append(0.0)
        for i in range(len(precision)):
            precision[i] = max(precision[i:])
        ap = 0
        for i in range(len(recall) - 1):
            ap += (recall[i + 1] - recall[i]) * precision[i + 1]
        return ap

    def mean_average_precision(self):
        """statistics.py:199-214: mean over classes that occur in the ground truth."""
        return float(np.mean([self.average_precision(c) for c in self.gt_count]))
