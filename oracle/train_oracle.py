"""
oracle/train_oracle.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU restatement of the reference's training step (SURVEY.md section 8 rows f2 + f3) on torch-CPU autograd:
  label_proposals        models/faster_rcnn.py:421-510
  sample_proposals       models/faster_rcnn.py:512-561   (torch.randperm on the CPU generator, as the reference)
  sample_rpn_minibatch   models/faster_rcnn.py:364-419   (python random.sample, as the reference)
  rpn_class_loss / rpn_regression_loss            models/rpn.py:176-272
  detector_class_loss / detector_regression_loss  models/detector.py:83-155
  RoIPoolFunction        torchvision.ops.RoIPool forward + backward (third party, restated: gradient to the
                         first maximum of each bin in (h, w) scan order; PARITY UNPINNED)
  train_step             models/faster_rcnn.py:228-362 + torch.optim.SGD as built at __main__.py:98-105

Pinned against the imported reference by oracle/make_golden.py --train (same seeds -> identical losses,
gradients and updated weights); the reference's own RoIPool import is served by RoIPoolFunction there.
"""
import random

import numpy as np
import torch as t
from torch.nn import functional as F

from oracle import frcnn_oracle as orc

_S1 = "_stage1_feature_extractor."
_S2 = "_stage2_region_proposal_network."
_S3 = "_stage3_detector_network."

VGG_LAYERS = [("_block1_conv1", False), ("_block1_conv2", True), ("_block2_conv1", False), ("_block2_conv2", True),
              ("_block3_conv1", False), ("_block3_conv2", False), ("_block3_conv3", True),
              ("_block4_conv1", False), ("_block4_conv2", False), ("_block4_conv3", True),
              ("_block5_conv1", False), ("_block5_conv2", False), ("_block5_conv3", False)]
FROZEN = ("_block1_conv1", "_block1_conv2", "_block2_conv1", "_block2_conv2")        # vgg16.py:49-58


def trainable_weight_keys(sd):
    """
    The parameters __main__.py:98-105 hands to SGD: requires_grad and "weight" in the key (biases are NOT trained).
    VGG-16: blocks 1-2 frozen (vgg16.py:49-58).  ResNet: conv1, bn1, layer1 and EVERY BatchNorm frozen
    (resnet.py:48-55,86,123), so the conv weights of layer2, layer3 and layer4 train.
    """
    keys = []
    resnet = orc.is_resnet(sd)
    rfe = _S1 + "_feature_extractor."
    for k in sd:
        if "weight" not in k:
            continue
        if resnet:
            if k.startswith(rfe + "0.") or k.startswith(rfe + "1.") or k.startswith(rfe + "4."):
                continue
            if ".bn" in k or ".downsample.1." in k:            # BatchNorm affine weights are frozen
                continue
        elif any((_S1 + f + ".") in k for f in FROZEN):
            continue
        keys.append(k)
    return keys


# ---- f2: labelling / sampling -----------------------------------------------------------------------
def t_iou(boxes1, boxes2):
    """math_utils.py:39-63, float32 torch ops."""
    tl = t.maximum(boxes1[:, None, 0:2], boxes2[:, 0:2])
    br = t.minimum(boxes1[:, None, 2:4], boxes2[:, 2:4])
    ok = t.all(tl < br, axis=2)
    inter = ok * t.prod(br - tl, dim=2)
    a1 = t.prod(boxes1[:, 2:4] - boxes1[:, 0:2], dim=1)
    a2 = t.prod(boxes2[:, 2:4] - boxes2[:, 0:2], dim=1)
    union = a1[:, None] + a2 - inter
    return inter / (union + 1e-7)


def label_proposals(proposals, gt_corners, gt_class_idx, num_classes, min_background_iou_threshold=0.0,
                    min_object_iou_threshold=0.5, means=(0, 0, 0, 0), stds=(0.1, 0.1, 0.2, 0.2)):
    """faster_rcnn.py:421-510.  proposals (N,4) f32 tensor, gt_corners (M,4) f32, gt_class_idx (M,) long."""
    gt_corners = t.as_tensor(gt_corners, dtype=t.float32)
    gt_class_idx = t.as_tensor(gt_class_idx, dtype=t.long)
    proposals = t.vstack([proposals, gt_corners])
    ious = t_iou(proposals, gt_corners)
    best = t.max(ious, dim=1).values
    box_idx = t.argmax(ious, dim=1)
    cls = gt_class_idx[box_idx]
    gtc = gt_corners[box_idx]
    idxs = t.where(best >= min_background_iou_threshold)[0]
    proposals, best, cls, gtc = proposals[idxs], best[idxs], cls[idxs].clone(), gtc[idxs]
    cls[best < min_object_iou_threshold] = 0
    n = proposals.shape[0]
    gt_classes = t.zeros((n, num_classes), dtype=t.float32)
    gt_classes[t.arange(n), cls] = 1.0
    pc = 0.5 * (proposals[:, 0:2] + proposals[:, 2:4])
    ps = proposals[:, 2:4] - proposals[:, 0:2]
    gc = 0.5 * (gtc[:, 0:2] + gtc[:, 2:4])
    gs = gtc[:, 2:4] - gtc[:, 0:2]
    tg = t.empty((n, 4), dtype=t.float32)
    tg[:, 0:2] = (gc - pc) / ps
    tg[:, 2:4] = t.log(gs / ps)
    tg[:, :] -= t.tensor(means, dtype=t.float32)
    tg[:, :] /= t.tensor(stds, dtype=t.float32)
    gt_box_deltas = t.zeros((n, 2, 4 * (num_classes - 1)), dtype=t.float32)
    gt_box_deltas[:, 0, :] = t.repeat_interleave(gt_classes, repeats=4, dim=1)[:, 4:]
    gt_box_deltas[:, 1, :] = t.tile(tg, dims=(1, num_classes - 1))
    return proposals, gt_classes, gt_box_deltas


def sample_proposals(proposals, gt_classes, gt_box_deltas, max_proposals, positive_fraction, detail=None):
    """faster_rcnn.py:512-561 (consumes the global torch CPU generator exactly as the reference does)."""
    if max_proposals <= 0:
        return proposals, gt_classes, gt_box_deltas
    class_indices = t.argmax(gt_classes, axis=1)
    pos = t.where(class_indices > 0)[0]
    neg = t.where(class_indices <= 0)[0]
    num_samples = min(max_proposals, len(class_indices))
    n_pos = min(round(num_samples * positive_fraction), len(pos))
    n_neg = min(num_samples - n_pos, len(neg))
    if n_pos <= 0 or n_neg <= 0:
        return proposals[[]], gt_classes[[]], gt_box_deltas[[]]
    ps = pos[t.randperm(len(pos))[0:n_pos]]
    ns = neg[t.randperm(len(neg))[0:n_neg]]
    indices = t.cat([ps, ns])
    if detail is not None:
        detail["proposal_sample_indices"] = indices.numpy().copy()
    return proposals[indices], gt_classes[indices], gt_box_deltas[indices]


def sample_rpn_minibatch(rpn_map, object_indices, background_indices, rpn_minibatch_size=256, detail=None):
    """faster_rcnn.py:364-419; rpn_map (1,H,W,9,6) tensor, index arrays (N,3) of (y,x,k)."""
    pos, neg = object_indices, background_indices
    assert len(pos) + len(neg) >= rpn_minibatch_size
    assert len(pos) > 0
    n_pos = min(rpn_minibatch_size // 2, len(pos))
    n_neg = rpn_minibatch_size - n_pos
    pi = random.sample(range(len(pos)), n_pos)
    ni = random.sample(range(len(neg)), n_neg)
    train = np.concatenate([pos[pi], neg[ni]])
    out = rpn_map.clone()
    out[:, :, :, :, 0] = 0
    out[(np.zeros(len(train)), train[:, 0], train[:, 1], train[:, 2], 0)] = 1
    if detail is not None:
        w = rpn_map.shape[2]
        detail["rpn_sample_flat"] = ((train[:, 0] * w + train[:, 1]) * 9 + train[:, 2]).astype(np.int32)
    return out


# ---- losses -----------------------------------------------------------------------------------------
def rpn_class_loss(predicted_scores, y_true):
    """rpn.py:176-216"""
    y_true_class = y_true[:, :, :, :, 1].reshape(predicted_scores.shape)
    y_mask = y_true[:, :, :, :, 0].reshape(predicted_scores.shape)
    n_cls = t.count_nonzero(y_mask) + 1e-7
    loss_all = F.binary_cross_entropy(input=predicted_scores, target=y_true_class, reduction="none")
    return t.sum(y_mask * loss_all) / n_cls


def rpn_regression_loss(predicted_box_deltas, y_true):
    """rpn.py:218-272"""
    sigma_squared = 9.0
    y_true_regression = y_true[:, :, :, :, 2:6].reshape(predicted_box_deltas.shape)
    y_included = y_true[:, :, :, :, 0].reshape(y_true.shape[0:4])
    y_positive = y_true[:, :, :, :, 1].reshape(y_true.shape[0:4])
    y_mask = (y_included * y_positive).repeat_interleave(repeats=4, dim=3)
    n_cls = t.count_nonzero(y_included) + 1e-7
    x = y_true_regression - predicted_box_deltas
    x_abs = t.abs(x)
    is_neg = (x_abs < (1.0 / sigma_squared)).float()
    r_neg = 0.5 * x * x * sigma_squared
    r_pos = x_abs - 0.5 / sigma_squared
    loss_all = is_neg * r_neg + (1.0 - is_neg) * r_pos
    return 1.0 * t.sum(y_mask * loss_all) / n_cls


def detector_class_loss(predicted_classes, y_true):
    """detector.py:83-104"""
    per_row = -(y_true * t.log(predicted_classes + 1e-7)).sum(dim=1)
    n = per_row.shape[0] + 1e-7
    return 1.0 * (t.sum(per_row) / n)


def detector_regression_loss(predicted_box_deltas, y_true):
    """detector.py:106-155"""
    y_mask = y_true[:, 0, :]
    y_t = y_true[:, 1, :]
    x = y_t - predicted_box_deltas
    x_abs = t.abs(x)
    is_neg = (x_abs < 1.0).float()
    r_neg = 0.5 * x * x * 1.0
    r_pos = x_abs - 0.5 / 1.0
    losses = is_neg * r_neg + (1.0 - is_neg) * r_pos
    n = y_true.shape[0] + 1e-7
    return 1.0 * t.sum(y_mask * losses) / n


# ---- RoIPool with backward ----------------------------------------------------------------------------
class RoIPoolFunction(t.autograd.Function):
    """torchvision.ops.RoIPool((7,7), scale): input (1,C,H,W), rois (K,5) = (b, x1, y1, x2, y2)."""
    @staticmethod
    def forward(ctx, inp, rois, output_size, spatial_scale):
        fm = inp.detach().numpy()[0]
        c, h, w = fm.shape
        r = rois.detach().numpy().astype(np.float32)
        k = r.shape[0]
        out = np.zeros((k, c, output_size, output_size), dtype=np.float32)
        arg = np.full((k, c, output_size, output_size), -1, dtype=np.int64)
        scale = np.float32(spatial_scale)
        for i in range(k):
            rs_w = int(orc._c_round(r[i, 1] * scale)); rs_h = int(orc._c_round(r[i, 2] * scale))
            re_w = int(orc._c_round(r[i, 3] * scale)); re_h = int(orc._c_round(r[i, 4] * scale))
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
                        win = fm[:, hs:he, ws:we].reshape(c, -1)
                        am = win.argmax(axis=1)                       # first maximum, (h, w) scan order
                        out[i, :, ph, pw] = win[np.arange(c), am]
                        arg[i, :, ph, pw] = (hs + am // (we - ws)) * w + (ws + am % (we - ws))
        ctx.save_for_backward(t.from_numpy(arg))
        ctx.shape = tuple(inp.shape)
        return t.from_numpy(out)

    @staticmethod
    def backward(ctx, grad_out):
        (arg,) = ctx.saved_tensors
        _, c, h, w = ctx.shape
        g = t.zeros((c, h * w), dtype=grad_out.dtype)
        k = arg.shape[0]
        ch = t.arange(c)
        for i in range(k):                                    # RoIs ascending, bins (ph, pw) ascending
            for ph in range(arg.shape[2]):
                for pw in range(arg.shape[3]):
                    a = arg[i, :, ph, pw]
                    m = a >= 0
                    if bool(m.any()):
                        g[ch[m], a[m]] += grad_out[i, :, ph, pw][m]
        return g.reshape(1, c, h, w), None, None, None


def roi_pool_autograd(feature_map, proposals):
    """detector.py:65-72: (y1,x1,y2,x2) proposals -> (b,x1,y1,x2,y2) rois -> RoIPool 7x7 @ 1/16."""
    rois = t.zeros((proposals.shape[0], 5), dtype=t.float32)
    rois[:, 1:] = proposals[:, [1, 0, 3, 2]]
    return RoIPoolFunction.apply(feature_map, rois, 7, 1.0 / 16.0)


class RoIAlignFunction(t.autograd.Function):
    """torchvision.ops.roi_align(input (1,C,H,W), rois (K,5), 7, scale, sampling_ratio, aligned=False): forward and the gradient
    with respect to the input through the oracle's restatement (orc.roi_align / orc.roi_align_backward; float64 accumulation in
    the backward).  Beyond the reference, which trains with RoIPool."""
    @staticmethod
    def forward(ctx, inp, rois, output_size, spatial_scale, sampling_ratio):
        ctx.meta = (tuple(inp.shape), rois.numpy().copy(), output_size, spatial_scale, sampling_ratio)
        return t.from_numpy(orc.roi_align(inp.detach().numpy(), rois.numpy(), output_size, spatial_scale, sampling_ratio, False))

    @staticmethod
    def backward(ctx, grad_out):
        shape, rois, output_size, spatial_scale, sampling_ratio = ctx.meta
        d = orc.roi_align_backward(grad_out.numpy(), shape, rois, output_size, spatial_scale, sampling_ratio, False)
        return t.from_numpy(d.astype(np.float32)), None, None, None, None


def roi_align_autograd(feature_map, proposals, sampling_ratio=2):
    rois = t.zeros((proposals.shape[0], 5), dtype=t.float32)
    rois[:, 1:] = proposals[:, [1, 0, 3, 2]]
    return RoIAlignFunction.apply(feature_map, rois, 7, 1.0 / 16.0, sampling_ratio)


# ---- reduced-precision gradient GEMMs (grad_math="bf16"; beyond the reference, which trains in float32) ------------------
# What csrc/gemm_tn.hip's bf16 kernel computes, restated: every gradient GEMM of the step -- the weight gradient of every
# convolution and dense layer, the data gradient of the dense layers and of the RPN's 1x1 heads -- multiplies operands rounded to
# bfloat16 (round to nearest even: torch's .bfloat16()) and accumulates in float32.  Round 4: so do the forward and data-gradient
# convolutions of the trainable ResNet bottlenecks (_ConvBnGradBf16 below).  The VGG-16 / RPN 3x3 forward and data-gradient
# convolutions, bias gradients, losses and the optimizer are float32 as before.
def _bf16r(x):
    return x.to(t.bfloat16).to(t.float32)


class _ConvGradBf16(t.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, padding, dense):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding, dense, b is not None)
        return F.conv2d(x, w, b, stride=stride, padding=padding)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        stride, padding, dense, has_b = ctx.cfg
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = (t.nn.grad.conv2d_input(x.shape, _bf16r(w), _bf16r(g), stride=stride, padding=padding) if dense
                  else t.nn.grad.conv2d_input(x.shape, w, g, stride=stride, padding=padding))
        if ctx.needs_input_grad[1]:
            gw = t.nn.grad.conv2d_weight(_bf16r(x), w.shape, _bf16r(g), stride=stride, padding=padding)
        if has_b and ctx.needs_input_grad[2]:
            gb = g.sum(dim=(0, 2, 3))
        return gx, gw, gb, None, None, None


class _LinearGradBf16(t.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gr = _bf16r(g)
        gx = gr @ _bf16r(w) if ctx.needs_input_grad[0] else None
        gw = gr.t() @ _bf16r(x) if ctx.needs_input_grad[1] else None
        gb = g.sum(dim=0) if ctx.needs_input_grad[2] else None
        return gx, gw, gb


class _ConvBnGradBf16(t.autograd.Function):
    """conv + frozen BatchNorm (fasterrcnn_amd/training.py _TrainConv: one convolution with the folded weight w * scale[co]; the raw
    weight's gradient is the folded weight's bf16 gradient GEMM times scale[co]).  `trainable` (round 4: BASELINE configs[4] as written;
    the convolutions of layer2 / layer3 / layer4, csrc/conv_gather.hip conv_gather_bf16_kernel): the FORWARD convolution and the data
    gradient multiply operands rounded to bfloat16 too -- the activations (or output gradients) and the FOLDED weight w * scale[co] --
    with float32 accumulation; bias, residual and ReLU stay float32.  The frozen layers below (conv1, layer1) run the inference kernels."""
    @staticmethod
    def forward(ctx, x, w, gamma, beta, mean, var, stride, padding, trainable):
        scale = gamma / t.sqrt(var + 1e-5)
        ctx.save_for_backward(x, w, scale)
        ctx.cfg = (stride, padding, trainable)
        if not trainable:
            return F.batch_norm(F.conv2d(x, w, stride=stride, padding=padding), mean, var, gamma, beta, False, 0.0, 1e-5)
        wf = w * scale.reshape(-1, 1, 1, 1)
        return F.conv2d(_bf16r(x), _bf16r(wf), stride=stride, padding=padding) + (beta - mean * scale).reshape(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, g):
        x, w, scale = ctx.saved_tensors
        stride, padding, trainable = ctx.cfg
        sc = scale.reshape(-1, 1, 1, 1)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = (t.nn.grad.conv2d_input(x.shape, _bf16r(w * sc), _bf16r(g), stride=stride, padding=padding) if trainable
                  else t.nn.grad.conv2d_input(x.shape, w * sc, g, stride=stride, padding=padding))
        gw = sc * t.nn.grad.conv2d_weight(_bf16r(x), w.shape, _bf16r(g), stride=stride, padding=padding) if ctx.needs_input_grad[1] else None
        return gx, gw, None, None, None, None, None, None, None


def _conv_bn_bf16(x, sd, wkey, bn_prefix, stride, padding):
    trainable = wkey.startswith((orc._RFE + "5.", orc._RFE + "6.", orc._RL4))           # layer2, layer3, layer4 (resnet.py:48-55,86,123)
    return _ConvBnGradBf16.apply(x, sd[wkey], sd[bn_prefix + "weight"], sd[bn_prefix + "bias"], sd[bn_prefix + "running_mean"],
                                 sd[bn_prefix + "running_var"], stride, padding, trainable)


def _conv2d(x, w, b, padding, grad_math, dense=False):
    if grad_math == "bf16":
        return _ConvGradBf16.apply(x, w, b, 1, padding, dense)
    return F.conv2d(x, w, b, padding=padding)


def _linear(x, w, b, grad_math):
    return _LinearGradBf16.apply(x, w, b) if grad_math == "bf16" else F.linear(x, w, b)


# ---- the step ---------------------------------------------------------------------------------------
def vgg16_features_train(p, image, detail=None, grad_math="f32"):
    y = image
    for name, pool in VGG_LAYERS:
        y = F.relu(_conv2d(y, p[_S1 + name + ".weight"], p[_S1 + name + ".bias"], 1, grad_math))
        if detail is not None:
            detail[name] = y
        if pool:
            y = F.max_pool2d(y, kernel_size=2, stride=2)
    return y


def train_step(sd, image, anchor_map, anchor_valid_map, gt_rpn_map, gt_rpn_object_indices, gt_rpn_background_indices,
               gt_corners, gt_class_idx, num_classes, lr, momentum, weight_decay, momentum_buffers=None,
               rpn_minibatch_size=256, proposal_batch_size=128, allow_edge_proposals=True, detail=None, roi_pooling="pool",
               sampling_ratio=2, grad_math="f32"):
    """
    faster_rcnn.py:228-362 for the VGG-16 backbone (dropout 0) followed by SGD.step().
    sd: state_dict of float32 CPU tensors (not modified).  Returns (losses dict, grads dict, new_sd, buffers).
    RNG use (python `random`, then torch's CPU generator) is in the reference's order.
    grad_math="bf16": the gradient GEMMs on bfloat16-rounded operands (see above); "f32" is the reference.
    """
    if grad_math not in ("f32", "bf16"):
        raise ValueError("grad_math")
    train_keys = trainable_weight_keys(sd)
    p = {k: v.clone().requires_grad_(k in train_keys) for k, v in sd.items()}
    image_shape = tuple(image.shape[1:])
    resnet = orc.is_resnet(sd)
    orc.CONV_BN = _conv_bn_bf16 if grad_math == "bf16" else None
    try:
        return _train_step(sd, p, train_keys, image, image_shape, resnet, anchor_map, anchor_valid_map, gt_rpn_map,
                           gt_rpn_object_indices, gt_rpn_background_indices, gt_corners, gt_class_idx, num_classes, lr, momentum,
                           weight_decay, momentum_buffers, rpn_minibatch_size, proposal_batch_size, allow_edge_proposals, detail,
                           roi_pooling, sampling_ratio, grad_math)
    finally:
        orc.CONV_BN = None


def _train_step(sd, p, train_keys, image, image_shape, resnet, anchor_map, anchor_valid_map, gt_rpn_map, gt_rpn_object_indices,
                gt_rpn_background_indices, gt_corners, gt_class_idx, num_classes, lr, momentum, weight_decay, momentum_buffers,
                rpn_minibatch_size, proposal_batch_size, allow_edge_proposals, detail, roi_pooling, sampling_ratio, grad_math):
    fm = orc.resnet_features(p, image) if resnet else vgg16_features_train(p, image, detail, grad_math)
    # stage 2 (rpn.py:88-156) with 12000 / 2000
    y = F.relu(_conv2d(fm, p[_S2 + "_rpn_conv1.weight"], p[_S2 + "_rpn_conv1.bias"], 1, grad_math))
    score_map = t.sigmoid(_conv2d(y, p[_S2 + "_rpn_class.weight"], p[_S2 + "_rpn_class.bias"], 0, grad_math, dense=True))
    delta_map = _conv2d(y, p[_S2 + "_rpn_boxes.weight"], p[_S2 + "_rpn_boxes.bias"], 0, grad_math, dense=True)
    score_map = score_map.permute(0, 2, 3, 1).contiguous()
    delta_map = delta_map.permute(0, 2, 3, 1).contiguous()
    with t.no_grad():
        proposals = orc.proposals_from_maps(score_map.detach(), delta_map.detach(), image_shape, anchor_map,
                                            anchor_valid_map, 12000, 2000, allow_edge_proposals)
    if detail is not None:
        detail["rpn_proposals"] = proposals.clone()
    minibatch = sample_rpn_minibatch(gt_rpn_map, gt_rpn_object_indices, gt_rpn_background_indices, rpn_minibatch_size, detail)
    props, gt_classes, gt_box_deltas = label_proposals(proposals, gt_corners, gt_class_idx, num_classes, 0.0, 0.5)
    if detail is not None:
        detail["labelled"] = (props.clone(), gt_classes.clone(), gt_box_deltas.clone())
    props, gt_classes, gt_box_deltas = sample_proposals(props, gt_classes, gt_box_deltas, proposal_batch_size, 0.25, detail)
    if detail is not None:
        detail["sampled"] = (props.clone(), gt_classes.clone(), gt_box_deltas.clone())
    # stage 3 (detector.py:65-80)
    pooled = roi_align_autograd(fm, props, sampling_ratio) if roi_pooling == "align" else roi_pool_autograd(fm, props)
    if resnet:
        h1 = None
        h2 = orc.resnet_pool_to_feature_vector(p, pooled)                     # layer4 + mean (resnet.py:109-118)
    else:
        x = pooled.reshape(pooled.shape[0], 512 * 7 * 7)
        pv = _S3 + "_pool_to_feature_vector."
        h1 = F.relu(_linear(x, p[pv + "_fc1.weight"], p[pv + "_fc1.bias"], grad_math))
        h2 = F.relu(_linear(h1, p[pv + "_fc2.weight"], p[pv + "_fc2.bias"], grad_math))
    classes = F.softmax(_linear(h2, p[_S3 + "_classifier.weight"], p[_S3 + "_classifier.bias"], grad_math), dim=1)
    deltas = _linear(h2, p[_S3 + "_regressor.weight"], p[_S3 + "_regressor.bias"], grad_math)
    l_rc = rpn_class_loss(score_map, minibatch)
    l_rr = rpn_regression_loss(delta_map, minibatch)
    l_dc = detector_class_loss(classes, gt_classes)
    l_dr = detector_regression_loss(deltas, gt_box_deltas)
    total = l_rc + l_rr + l_dc + l_dr
    watched = {}
    if detail is not None:          # gradients of intermediates, for stage-by-stage parity reports
        watched = dict(feature_map=fm, rpn_trunk=y, pooled=pooled, fc2=h2, score_map=score_map, delta_map=delta_map,
                       classes=classes, deltas=deltas)
        if not resnet:
            watched["fc1"] = h1
            watched.update({name: detail[name] for name, _ in VGG_LAYERS[4:]})
        for v in watched.values():
            v.retain_grad()
    total.backward()
    if detail is not None:
        detail["grad_of"] = {k: v.grad.detach().clone() for k, v in watched.items() if v.grad is not None}
    losses = dict(rpn_class=l_rc.item(), rpn_regression=l_rr.item(), detector_class=l_dc.item(),
                  detector_regression=l_dr.item(), total=total.item())
    grads = {k: p[k].grad.detach().clone() for k in train_keys}
    # torch.optim.SGD (momentum, weight decay, dampening 0, no nesterov)
    new_sd = {k: v.clone() for k, v in sd.items()}
    bufs = {} if momentum_buffers is None else {k: v.clone() for k, v in momentum_buffers.items()}
    for k in train_keys:
        g = grads[k]
        if weight_decay != 0:
            g = g.add(sd[k], alpha=weight_decay)
        if momentum != 0:
            if k not in bufs:
                bufs[k] = g.clone()
            else:
                bufs[k].mul_(momentum).add_(g)
            g = bufs[k]
        new_sd[k] = sd[k].add(g, alpha=-lr)
    if detail is not None:
        detail.update(feature_map=fm.detach(), rpn_trunk=y.detach(), score_map=score_map.detach(),
                      delta_map=delta_map.detach(), pooled=pooled.detach(), fc1=None if h1 is None else h1.detach(), fc2=h2.detach(),
                      classes=classes.detach(), deltas=deltas.detach(), minibatch=minibatch)
    return losses, grads, new_sd, bufs
