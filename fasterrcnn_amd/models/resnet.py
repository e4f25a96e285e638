"""
ResNet backbones, mirroring pytorch/FasterRCNN/models/resnet.py:27-185 (Architecture,
FeatureExtractor, PoolToFeatureVector, ResNetBackbone) with the reference's attribute names, so
the state_dict keys are the reference's:
  _stage1_feature_extractor._feature_extractor.{0=conv1,1=bn1,4=layer1,5=layer2,6=layer3}....
  _stage3_detector_network._pool_to_feature_vector._layer4....
(inner names as torchvision's ResNet: N.convK.weight, N.bnK.{weight,bias,running_mean,running_var,
num_batches_tracked}, N.downsample.{0,1}....).

The reference wraps `torchvision.models.resnet{50,101,152}(weights=IMAGENET1K_V1)` (resnet.py:144-149).
torchvision is a third-party dependency that is not available here and there is no network for the
ImageNet weights, so `_ResNetParams` below restates torchvision's v1.5 Bottleneck architecture as a
PARAMETER HOLDER (same module tree, torchvision's default initialisation); trained weights are
loaded through `load_state_dict` exactly as with the reference.  All arithmetic runs in
csrc/conv_gather.hip / conv.hip: BatchNorm is frozen in eval mode by the reference
(resnet.py:58-77,100-107) and is folded into the preceding convolution when the weights are packed.
"""
from enum import Enum
from math import ceil

import torch as t
from torch import nn

from .. import _native as nv
from .. import runtime as rt
from ..datasets import image
from .backbone import Backbone


class Architecture(Enum):
    ResNet50 = "ResNet50"
    ResNet101 = "ResNet101"
    ResNet152 = "ResNet152"


_BLOCKS = {Architecture.ResNet50: (3, 4, 6, 3), Architecture.ResNet101: (3, 4, 23, 3), Architecture.ResNet152: (3, 8, 36, 3)}


class _Bottleneck(nn.Module):
    """torchvision.models.resnet.Bottleneck (v1.5: the stride sits on the 3x3 conv), parameters only."""
    expansion = 4

    def __init__(self, inplanes, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or inplanes != planes * 4:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))
        self.stride = stride


class _ResNetParams(nn.Module):
    """The module tree of torchvision's ResNet (conv1, bn1, relu, maxpool, layer1..layer4)."""
    def __init__(self, blocks):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        inplanes = 64
        layers = []
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), blocks)):
            stride = 1 if i == 0 else 2
            mods = []
            for b in range(n):
                mods.append(_Bottleneck(inplanes, planes, stride if b == 0 else 1))
                inplanes = planes * 4
            layers.append(nn.Sequential(*mods))
        self.layer1, self.layer2, self.layer3, self.layer4 = layers
        for m in self.modules():          # torchvision's default initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


def fold_conv_bn(conv, bn):
    """(packed weight, bias) of conv followed by frozen BatchNorm, via frcnn_fold_bn_pack."""
    w = rt.as_f32_cuda(conv.weight.detach(), "conv weight")
    cout, cin, k = int(w.shape[0]), int(w.shape[1]), int(w.shape[2])
    wp = t.empty((cin * k * k, cout) if cin == 3 else (k * k, cout, cin), dtype=t.float32, device=w.device)
    bp = t.empty((cout,), dtype=t.float32, device=w.device)
    args = [rt.as_f32_cuda(x.detach(), "bn tensor") for x in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]
    with t.cuda.device(w.device):
        nv.check(nv.lib().frcnn_fold_bn_pack(nv.ptr(w), nv.ptr(args[0]), nv.ptr(args[1]), nv.ptr(args[2]), nv.ptr(args[3]),
                                             float(bn.eps), cout, cin, k, nv.ptr(wp), nv.ptr(bp), nv.stream_ptr()),
                 "frcnn_fold_bn_pack")
    return wp, bp, args


def _bn_params(bn):
    return [bn.weight, bn.bias, bn.running_mean, bn.running_var]


def fold_conv_bn_winograd(conv, bn, fused=False):
    """fused=True: the one-launch kernel's flat bank (frcnn_pack_conv3x3_winograd_fused), else
    (transformed filter bank [16][cout][cin], bias) of a 3x3 conv followed by frozen BatchNorm: the same float32 fold as
    frcnn_fold_bn_pack (scale = gamma / sqrt(var + eps) applied to the filter rows, bias = beta - mean * scale), then G g G^T."""
    w = rt.as_f32_cuda(conv.weight.detach(), "conv weight")
    cout, cin = int(w.shape[0]), int(w.shape[1])
    args = [rt.as_f32_cuda(x.detach(), "bn tensor") for x in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]
    scale = t.empty((cout,), dtype=t.float32, device=w.device)
    shift = t.empty((cout,), dtype=t.float32, device=w.device)
    u = t.empty((16 * cout * cin,) if fused else (16, cout, cin), dtype=t.float32, device=w.device)
    with t.cuda.device(w.device):
        lib = nv.lib()
        nv.check(lib.frcnn_bn_scale_shift(nv.ptr(args[0]), nv.ptr(args[1]), nv.ptr(args[2]), nv.ptr(args[3]), float(bn.eps), cout,
                                          nv.ptr(scale), nv.ptr(shift), nv.stream_ptr()), "frcnn_bn_scale_shift")
        pack = lib.frcnn_pack_conv3x3_winograd_fused if fused else lib.frcnn_pack_conv3x3_winograd
        nv.check(pack(nv.ptr(w), nv.ptr(scale), nv.ptr(u), cout, cin, nv.stream_ptr()), "frcnn_pack_conv3x3_winograd")
    return u, shift, args + [scale]


def x6_conv1x1_ok(cin, cout):
    """The 1x1 convolutions worth running as f32x6 GEMMs (csrc/gemm_x6t.hip): enough reduction depth for the 16-k stages to
    amortise a block's prologue / epilogue and enough output channels to fill a 128-column tile.  layer1's 64-channel
    convolutions and layer2's 128 -> 512 expansions stay on the exact-f32 gather kernel."""
    return cin % 16 == 0 and cout % 4 == 0 and cin >= 256 and cout >= 128


def records_x6t(wp, cout, cin):
    """folded float32 [1][cout][cin] pack of a 1x1 convolution -> the x6t records of the [cout][cin] matrix (uint8)."""
    lib = nv.lib()
    rows = (cout + nv.X6T_COL_TILE - 1) // nv.X6T_COL_TILE * nv.X6T_COL_TILE
    rec = t.empty((int(lib.frcnn_x6t_record_bytes(rows, cin)),), dtype=t.uint8, device=wp.device)
    with t.cuda.device(wp.device):
        nv.check(lib.frcnn_split_rows_x6t(nv.ptr(wp), cin, 0, nv.ptr(rec), cout, rows, cin, 1, nv.stream_ptr()), "frcnn_split_rows_x6t")
    return rec


def blob_x3t(wp, cout, cin):
    """folded float32 [1][cout][cin] pack -> the packed f32x3 operand (int8: records of the row-scaled [cout][cin] matrix, then its scales)."""
    from .vgg16 import pack_rows_x3t
    rows = (cout + nv.X6T_COL_TILE - 1) // nv.X6T_COL_TILE * nv.X6T_COL_TILE
    return pack_rows_x3t(wp.reshape(cout, cin), rows)


def conv_nhwc_x3g(x, wp, bp, n, h, w, cin, cout, k, stride, pad, relu, xmax, wmax, ymax=None, residual=None, tickets=None, wsplit=False):
    """frcnn_conv_nhwc_x3g: the same convolution in the f32x3 arithmetic under one scale per tensor; xmax / wmax: one-element CUDA float
    tensors bounding |x| and |wp|, ymax: a zeroed one that receives max|y| (or None).  tickets: a ZEROED int32 CUDA tensor of
    nv.X3G_TILE_COUNTERS elements = frcnn_conv_nhwc_x3g_tickets (a split reduction is finished inside the kernel; the tensor is zero
    again afterwards).  wsplit: wp is a pack_x3g_weights image (FRCNN_X3G_WSPLIT).  Returns (y, ho, wo)."""
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    y = t.empty((n, ho, wo, cout), dtype=t.float32, device=x.device)
    lib = nv.lib()
    wsb = int(lib.frcnn_conv_workspace_bytes(n, h, w, cin, cout, k, stride, pad))
    ws = t.empty((max(wsb, 4) // 4,), dtype=t.float32, device=x.device)
    fl = (nv.RELU if relu else 0) | (nv.X3G_WSPLIT if wsplit else 0)
    with t.cuda.device(x.device):
        if tickets is not None:
            assert tickets.numel() >= nv.X3G_TILE_COUNTERS and tickets.dtype == t.int32
            nv.check(lib.frcnn_conv_nhwc_x3g_tickets(nv.ptr(x), nv.ptr(wp), nv.ptr(bp), nv.ptr(residual), nv.ptr(y), n, h, w, cin, cout,
                                                     k, stride, pad, fl, nv.ptr(xmax), nv.ptr(wmax), nv.ptr(ymax), nv.ptr(ws),
                                                     wsb, nv.ptr(tickets), nv.stream_ptr()), "frcnn_conv_nhwc_x3g_tickets")
        else:
            nv.check(lib.frcnn_conv_nhwc_x3g(nv.ptr(x), nv.ptr(wp), nv.ptr(bp), nv.ptr(residual), nv.ptr(y), n, h, w, cin, cout,
                                             k, stride, pad, fl, nv.ptr(xmax), nv.ptr(wmax), nv.ptr(ymax), nv.ptr(ws), wsb,
                                             nv.stream_ptr()), "frcnn_conv_nhwc_x3g")
    return y, ho, wo


def tensor_absmax(x):
    """one-element CUDA tensor max|x| through frcnn_tensor_absmax"""
    out = t.zeros((1,), dtype=t.float32, device=x.device)
    with t.cuda.device(x.device):
        nv.check(nv.lib().frcnn_tensor_absmax(nv.ptr(x), x.numel(), nv.ptr(out), nv.stream_ptr()), "frcnn_tensor_absmax")
    return out


def pack_block_g3(block):
    """The block for frcnn_bottleneck_weights.g3: the four plain BN-folded float32 packs + `wmax` (their absolute maxima)."""
    w1, b1, k1 = fold_conv_bn(block.conv1, block.bn1)
    w2, b2, k2 = fold_conv_bn(block.conv2, block.bn2)
    w3, b3, k3 = fold_conv_bn(block.conv3, block.bn3)
    out = {"w1": w1, "b1": b1, "w2": w2, "b2": b2, "w3": w3, "b3": b3, "wd": None, "bd": None,
           "cin": block.conv1.in_channels, "width": block.conv1.out_channels, "cout": block.conv3.out_channels,
           "stride": block.stride, "keep": [k1, k2, k3], "x6_mask": 0, "x3_mask": 0, "g3": 1}
    packs = [w1, w2, w3]
    if block.downsample is not None:
        out["wd"], out["bd"], kd = fold_conv_bn(block.downsample[0], block.downsample[1])
        out["keep"].append(kd)
        packs.append(out["wd"])
    out["wmax"] = t.cat([tensor_absmax(p) for p in packs] + ([] if len(packs) == 4 else [t.ones((1,), dtype=t.float32, device=w1.device)]))
    if G3_PRESPLIT:
        # round 6 (ABI 15): the packs split ONCE, here, into the kernel's operand format (frcnn_pack_conv_x3g_weights: same size, same
        # results bit for bit) -- the kernel then copies a weight piece into LDS instead of splitting it in every block of every launch
        for i, key in enumerate(["w1", "w2", "w3", "wd"][:len(packs)]):
            out[key] = pack_x3g_weights(out[key], out["wmax"][i:i + 1])
        out["g3"] = 2
    return out


G3_PRESPLIT = True          # (tools flip it for A/B runs; frcnn_bottleneck_weights.g3 = 2 / 1)


def pack_x3g_weights(wp, wmax):
    """frcnn_pack_conv_x3g_weights: the float32 pack [taps][cout][cin] -> its pre-split image (a float32-typed tensor of the same shape
    holding fp16 hi / lo pairs: only its bytes mean anything)."""
    taps, cout, cin = int(wp.shape[0]), int(wp.shape[1]), int(wp.shape[2])
    out = t.empty_like(wp)
    with t.cuda.device(wp.device):
        nv.check(nv.lib().frcnn_pack_conv_x3g_weights(nv.ptr(wp), nv.ptr(wmax), nv.ptr(out), taps, cout, cin, nv.stream_ptr()), "frcnn_pack_conv_x3g_weights")
    return out


def pack_block(block, math_mode="f32", single_map=False, x6=False, x3=False, g3=False):
    """dict of packed tensors + shape info for one Bottleneck.  single_map: the block runs on ONE map (layer1..3 of the feature
    extractor) -> its 3x3 is a one-launch Winograd layer in the f32_winograd mode; the per-RoI maps of layer4 use the batched form.
    x6 (f32_winograd mode only): the 1x1 convolutions with x6_conv1x1_ok() carry x6t record arrays instead of float32 packs
    (`x6_mask` bits FRCNN_X6_CONV1 / _CONV3 / _DOWN) and run as f32x6 GEMMs on the bf16 pipe; with x3 they carry f32x3 blobs instead
    (`x3_mask` = `x6_mask`: two fp16 terms per row-scaled operand, three MFMAs per product, csrc/gemm_x3t.hip)."""
    if g3:
        return pack_block_g3(block)
    w1, b1, k1 = fold_conv_bn(block.conv1, block.bn1)
    width = block.conv2.out_channels
    if math_mode == "f32_winograd" and nv.resnet_block_uses_winograd_fused(1 if single_map else 2, width, block.stride):
        w2, b2, k2 = fold_conv_bn_winograd(block.conv2, block.bn2, fused=True)
    elif math_mode == "f32_winograd" and nv.resnet_block_uses_winograd(width, block.stride):
        w2, b2, k2 = fold_conv_bn_winograd(block.conv2, block.bn2)
    else:
        w2, b2, k2 = fold_conv_bn(block.conv2, block.bn2)
    w3, b3, k3 = fold_conv_bn(block.conv3, block.bn3)
    out = {"w1": w1, "b1": b1, "w2": w2, "b2": b2, "w3": w3, "b3": b3, "wd": None, "bd": None,
           "cin": block.conv1.in_channels, "width": block.conv1.out_channels, "cout": block.conv3.out_channels,
           "stride": block.stride, "keep": [k1, k2, k3]}
    if block.downsample is not None:
        out["wd"], out["bd"], kd = fold_conv_bn(block.downsample[0], block.downsample[1])
        out["keep"].append(kd)
    out["x6_mask"] = 0
    if x6 and math_mode == "f32_winograd" and not single_map and width % 16 == 0 and width >= 256:
        # the per-RoI 3x3 of layer4 on the bf16 pipe: stride 1 = an x6 Winograd layer over the block's maps (frozen-BN scale folded
        # into the filter transform, bias = the BN shift), stride 2 = an im2col GEMM against the [cout][9 cin] matrix (tap-major)
        lib = nv.lib()
        if block.stride == 1:
            wsrc = rt.as_f32_cuda(block.conv2.weight.detach(), "conv weight")
            args = [rt.as_f32_cuda(x.detach(), "bn tensor") for x in _bn_params(block.bn2)]
            scale = t.empty((width,), dtype=t.float32, device=wsrc.device)
            shift = t.empty((width,), dtype=t.float32, device=wsrc.device)
            rec = t.empty((int(lib.frcnn_conv3x3_winograd_x6_pack_bytes(width, width)),), dtype=t.uint8, device=wsrc.device)
            with t.cuda.device(wsrc.device):
                nv.check(lib.frcnn_bn_scale_shift(nv.ptr(args[0]), nv.ptr(args[1]), nv.ptr(args[2]), nv.ptr(args[3]), float(block.bn2.eps), width,
                                                  nv.ptr(scale), nv.ptr(shift), nv.stream_ptr()), "frcnn_bn_scale_shift")
                if x3:
                    bank = t.empty((16, width, width), dtype=t.float32, device=wsrc.device)
                    rec = t.empty((int(lib.frcnn_conv3x3_winograd_x3_pack_bytes(width, width)),), dtype=t.int8, device=wsrc.device)
                    nv.check(lib.frcnn_pack_conv3x3_winograd(nv.ptr(wsrc), nv.ptr(scale), nv.ptr(bank), width, width, nv.stream_ptr()),
                             "frcnn_pack_conv3x3_winograd")
                    nv.check(lib.frcnn_pack_conv3x3_winograd_x3(nv.ptr(bank), nv.ptr(rec), width, width, nv.stream_ptr()),
                             "frcnn_pack_conv3x3_winograd_x3")
                else:
                    nv.check(lib.frcnn_pack_conv3x3_winograd_x6(nv.ptr(wsrc), nv.ptr(scale), nv.ptr(rec), width, width, nv.stream_ptr()),
                             "frcnn_pack_conv3x3_winograd_x6")
            out["w2"], out["b2"] = rec, shift
            out["keep"].append(args + [scale])
        else:
            wf, bf, kf = fold_conv_bn(block.conv2, block.bn2)                    # [9][cout][cin] folded float32
            mat = wf.permute(1, 0, 2).reshape(width, 9 * width).contiguous()
            out["w2"] = blob_x3t(mat, width, 9 * width) if x3 else records_x6t(mat, width, 9 * width)
            out["b2"] = bf
            out["keep"].append(kf)
        out["x6_mask"] |= 8
    if x6 and math_mode == "f32_winograd":
        for bit, key, ci, co in ((1, "w1", out["cin"], out["width"]), (2, "w3", out["width"], out["cout"]), (4, "wd", out["cin"], out["cout"])):
            if out[key] is not None and x6_conv1x1_ok(ci, co):
                out[key] = blob_x3t(out[key], co, ci) if x3 else records_x6t(out[key], co, ci)
                out["x6_mask"] |= bit
    out["x3_mask"] = out["x6_mask"] if x3 else 0
    return out


def block_params(block):
    ps = [block.conv1.weight, block.conv2.weight, block.conv3.weight] + _bn_params(block.bn1) + _bn_params(block.bn2) + _bn_params(block.bn3)
    if block.downsample is not None:
        ps += [block.downsample[0].weight] + _bn_params(block.downsample[1])
    return ps


def conv_nhwc(x, wp, bp, n, h, w, cin, cout, k, stride, pad, relu, residual=None, math=0):
    """frcnn_conv_nhwc on a flat NHWC CUDA tensor; returns (y, ho, wo).  math: nv.GRAD_MATHS value (the train step's reduced-precision
    forward: operands rounded to bfloat16, bf16 matrix pipe, float32 accumulation)."""
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    y = t.empty((n, ho, wo, cout), dtype=t.float32, device=x.device)
    lib = nv.lib()
    wsb = int(lib.frcnn_conv_workspace_bytes(n, h, w, cin, cout, k, stride, pad))
    ws = t.empty((max(wsb, 4) // 4,), dtype=t.float32, device=x.device)
    with t.cuda.device(x.device):
        nv.check(lib.frcnn_conv_nhwc_math(nv.ptr(x), nv.ptr(wp), nv.ptr(bp), nv.ptr(residual), nv.ptr(y), n, h, w, cin, cout,
                                          k, stride, pad, nv.RELU if relu else 0, int(math), nv.ptr(ws), wsb, nv.stream_ptr()),
                 "frcnn_conv_nhwc_math")
    return y, ho, wo


def conv1x1_x6(x, wrec, bp, n, h, w, cin, cout, stride, relu, residual=None):
    """1x1 convolution (stride 1 / 2) as an f32x6 GEMM: frcnn_split_pixels_x6t + frcnn_gemm_x6t; returns (y, ho, wo)."""
    lib = nv.lib()
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    m = n * ho * wo
    mp = (m + nv.X6T_ROW_TILE - 1) // nv.X6T_ROW_TILE * nv.X6T_ROW_TILE
    np_ = (cout + nv.X6T_COL_TILE - 1) // nv.X6T_COL_TILE * nv.X6T_COL_TILE
    rec = t.empty((int(lib.frcnn_x6t_record_bytes(mp, cin)),), dtype=t.uint8, device=x.device)
    y = t.empty((n, ho, wo, cout), dtype=t.float32, device=x.device)
    wsb = int(lib.frcnn_gemm_x6t_workspace_bytes(m, cout, cin, 1))
    ws = t.empty((max(wsb, 4),), dtype=t.uint8, device=x.device)
    with t.cuda.device(x.device):
        nv.check(lib.frcnn_split_pixels_x6t(nv.ptr(x), nv.ptr(rec), n, h, w, cin, stride, mp, nv.stream_ptr()), "frcnn_split_pixels_x6t")
        nv.check(lib.frcnn_gemm_x6t(nv.ptr(rec), mp, 0, nv.ptr(wrec), np_, 0, nv.ptr(bp), nv.ptr(residual), nv.ptr(y), cout, 0, m, cout, cin, 1,
                                    nv.RELU if relu else 0, nv.ptr(ws), wsb, nv.stream_ptr()), "frcnn_gemm_x6t")
    return y, ho, wo


def conv_x3(x, wblob, bp, n, h, w, cin, cout, stride, relu, residual=None, ksize=1):
    """1x1 (stride 1 / 2) or 3x3 / padding-1 convolution as an f32x3 GEMM: frcnn_pixel_absmax + frcnn_split_pixels_x3t /
    frcnn_split_patches3x3_x3t + frcnn_gemm_x3t; returns (y, ho, wo)."""
    lib = nv.lib()
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    k = cin * (9 if ksize == 3 else 1)
    m = n * ho * wo
    mp = (m + nv.X6T_ROW_TILE - 1) // nv.X6T_ROW_TILE * nv.X6T_ROW_TILE
    np_ = (cout + nv.X6T_COL_TILE - 1) // nv.X6T_COL_TILE * nv.X6T_COL_TILE
    rec = t.empty((int(lib.frcnn_x3t_record_bytes(mp, k)),), dtype=t.uint8, device=x.device)
    inv = t.empty((mp,), dtype=t.float32, device=x.device)
    cmax = t.empty((n * h * w,), dtype=t.float32, device=x.device)
    y = t.empty((n, ho, wo, cout), dtype=t.float32, device=x.device)
    wsb = int(lib.frcnn_gemm_x3t_workspace_bytes(m, cout, k, 1))
    ws = t.empty((max(wsb, 4),), dtype=t.uint8, device=x.device)
    wrec = int(lib.frcnn_x3t_record_bytes(np_, k))
    with t.cuda.device(x.device):
        nv.check(lib.frcnn_pixel_absmax(nv.ptr(x), nv.ptr(cmax), n * h * w, cin, nv.stream_ptr()), "frcnn_pixel_absmax")
        split = lib.frcnn_split_patches3x3_x3t if ksize == 3 else lib.frcnn_split_pixels_x3t
        nv.check(split(nv.ptr(x), nv.ptr(cmax), nv.ptr(rec), nv.ptr(inv), n, h, w, cin, stride, mp, nv.stream_ptr()), "frcnn_split_*_x3t")
        nv.check(lib.frcnn_gemm_x3t(nv.ptr(rec), nv.ptr(inv), mp, 0, 0, nv.ptr(wblob), wblob.data_ptr() + wrec, np_, 0, 0, nv.ptr(bp),
                                    nv.ptr(residual), nv.ptr(y), cout, 0, m, cout, k, 1, nv.RELU if relu else 0, nv.ptr(ws), wsb,
                                    nv.stream_ptr()), "frcnn_gemm_x3t")
    return y, ho, wo


def run_block(x, n, h, w, pb):
    """One Bottleneck on NHWC data through the C ABI (stage-level path; the fused model uses frcnn_resnet_forward)."""
    if pb.get("g3", 0):
        wm = pb["wmax"]
        xmax = tensor_absmax(x)
        m = t.zeros((3,), dtype=t.float32, device=x.device)
        sp = pb["g3"] == 2                                                  # pre-split packs (pack_x3g_weights)
        t1, _, _ = conv_nhwc_x3g(x, pb["w1"], pb["b1"], n, h, w, pb["cin"], pb["width"], 1, 1, 0, True, xmax, wm[0:1], m[0:1], wsplit=sp)
        t2, ho, wo = conv_nhwc_x3g(t1, pb["w2"], pb["b2"], n, h, w, pb["width"], pb["width"], 3, pb["stride"], 1, True, m[0:1], wm[1:2], m[1:2], wsplit=sp)
        identity = x
        if pb["wd"] is not None:
            identity, _, _ = conv_nhwc_x3g(x, pb["wd"], pb["bd"], n, h, w, pb["cin"], pb["cout"], 1, pb["stride"], 0, False, xmax, wm[3:4], wsplit=sp)
        out, _, _ = conv_nhwc_x3g(t2, pb["w3"], pb["b3"], n, ho, wo, pb["width"], pb["cout"], 1, 1, 0, True, m[1:2], wm[2:3], m[2:3], residual=identity, wsplit=sp)
        return out, ho, wo
    xm = pb.get("x6_mask", 0)
    x3 = pb.get("x3_mask", 0)
    conv1x1 = (lambda *a, **k: conv_x3(*a, **k)) if x3 else conv1x1_x6
    if xm & 1:
        t1, _, _ = conv1x1(x, pb["w1"], pb["b1"], n, h, w, pb["cin"], pb["width"], 1, True)
    else:
        t1, _, _ = conv_nhwc(x, pb["w1"], pb["b1"], n, h, w, pb["cin"], pb["width"], 1, 1, 0, True)
    if xm & 8:                                                          # layer4's 3x3 on the bf16 pipe (x6 Winograd / im2col GEMM)
        width = pb["width"]
        lib = nv.lib()
        if x3 and pb["stride"] == 1:
            ho, wo = h, w
            t2 = t.empty((n, h, w, width), dtype=t.float32, device=x.device)
            wsb = int(lib.frcnn_conv3x3_winograd_x3_workspace_bytes(n, h, w, width, width))
            ws = t.empty((wsb,), dtype=t.uint8, device=x.device)
            with t.cuda.device(x.device):
                nv.check(lib.frcnn_conv3x3_nhwc_winograd_x3(nv.ptr(t1), nv.ptr(pb["w2"]), nv.ptr(pb["b2"]), nv.ptr(t2), n, h, w, width, width,
                                                            nv.RELU, nv.ptr(ws), wsb, nv.stream_ptr()), "frcnn_conv3x3_nhwc_winograd_x3")
        elif x3:
            t2, ho, wo = conv_x3(t1, pb["w2"], pb["b2"], n, h, w, width, width, pb["stride"], True, ksize=3)
        elif pb["stride"] == 1:
            ho, wo = h, w
            t2 = t.empty((n, h, w, width), dtype=t.float32, device=x.device)
            wsb = int(lib.frcnn_conv3x3_winograd_x6_workspace_bytes(n, h, w, width, width))
            ws = t.empty((wsb,), dtype=t.uint8, device=x.device)
            with t.cuda.device(x.device):
                nv.check(lib.frcnn_conv3x3_nhwc_winograd_x6(nv.ptr(t1), nv.ptr(pb["w2"]), nv.ptr(pb["b2"]), nv.ptr(t2), n, h, w, width, width,
                                                            nv.RELU, nv.ptr(ws), wsb, nv.stream_ptr()), "frcnn_conv3x3_nhwc_winograd_x6")
        else:
            st = pb["stride"]
            ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
            m = n * ho * wo
            mp = (m + nv.X6T_ROW_TILE - 1) // nv.X6T_ROW_TILE * nv.X6T_ROW_TILE
            np_ = (width + nv.X6T_COL_TILE - 1) // nv.X6T_COL_TILE * nv.X6T_COL_TILE
            rec = t.empty((int(lib.frcnn_x6t_record_bytes(mp, 9 * width)),), dtype=t.uint8, device=x.device)
            t2 = t.empty((n, ho, wo, width), dtype=t.float32, device=x.device)
            wsb = int(lib.frcnn_gemm_x6t_workspace_bytes(m, width, 9 * width, 1))
            ws = t.empty((max(wsb, 4),), dtype=t.uint8, device=x.device)
            with t.cuda.device(x.device):
                nv.check(lib.frcnn_split_patches3x3_x6t(nv.ptr(t1), nv.ptr(rec), n, h, w, width, st, mp, nv.stream_ptr()), "frcnn_split_patches3x3_x6t")
                nv.check(lib.frcnn_gemm_x6t(nv.ptr(rec), mp, 0, nv.ptr(pb["w2"]), np_, 0, nv.ptr(pb["b2"]), None, nv.ptr(t2), width, 0, m, width,
                                            9 * width, 1, nv.RELU, nv.ptr(ws), wsb, nv.stream_ptr()), "frcnn_gemm_x6t")
    elif pb["w2"].dim() == 1:                                           # one-launch Winograd bank (f32_winograd mode, one map)
        assert n == 1
        width = pb["width"]
        ho, wo = h, w
        t2 = t.empty((n, h, w, width), dtype=t.float32, device=x.device)
        with t.cuda.device(x.device):
            nv.check(nv.lib().frcnn_conv3x3_nhwc_winograd_fused(nv.ptr(t1), nv.ptr(pb["w2"]), nv.ptr(pb["b2"]), nv.ptr(t2), h, w, width, width,
                                                                nv.RELU, nv.stream_ptr()), "frcnn_conv3x3_nhwc_winograd_fused")
    elif pb["w2"].dim() == 3 and int(pb["w2"].shape[0]) == 16:        # Winograd filter bank (f32_winograd mode)
        width = pb["width"]
        ho, wo = h, w
        t2 = t.empty((n, h, w, width), dtype=t.float32, device=x.device)
        lib = nv.lib()
        wsb = int(lib.frcnn_conv3x3_winograd_workspace_bytes(n, h, w, width, width))
        ws = t.empty((max(wsb, 4) // 4,), dtype=t.float32, device=x.device)
        with t.cuda.device(x.device):
            nv.check(lib.frcnn_conv3x3_nhwc_winograd(nv.ptr(t1), nv.ptr(pb["w2"]), nv.ptr(pb["b2"]), nv.ptr(t2), n, h, w, width, width,
                                                     nv.RELU, nv.ptr(ws), wsb, nv.stream_ptr()), "frcnn_conv3x3_nhwc_winograd")
    else:
        t2, ho, wo = conv_nhwc(t1, pb["w2"], pb["b2"], n, h, w, pb["width"], pb["width"], 3, pb["stride"], 1, True)
    identity = x
    if pb["wd"] is not None:
        if xm & 4:
            identity, _, _ = conv1x1(x, pb["wd"], pb["bd"], n, h, w, pb["cin"], pb["cout"], pb["stride"], False)
        else:
            identity, _, _ = conv_nhwc(x, pb["wd"], pb["bd"], n, h, w, pb["cin"], pb["cout"], 1, pb["stride"], 0, False)
    if xm & 2:
        out, _, _ = conv1x1(t2, pb["w3"], pb["b3"], n, ho, wo, pb["width"], pb["cout"], 1, True, residual=identity)
    else:
        out, _, _ = conv_nhwc(t2, pb["w3"], pb["b3"], n, ho, wo, pb["width"], pb["cout"], 1, 1, 0, True, residual=identity)
    return out, ho, wo


class FeatureExtractor(nn.Module):
    def __init__(self, resnet):
        super().__init__()
        # Feature extractor layers (resnet.py:38-46)
        self._feature_extractor = nn.Sequential(
            resnet.conv1,     # 0
            resnet.bn1,       # 1
            resnet.relu,      # 2
            resnet.maxpool,   # 3
            resnet.layer1,    # 4
            resnet.layer2,    # 5
            resnet.layer3     # 6
        )
        # Freeze initial layers and every batchnorm (resnet.py:48-55) -- inference is unaffected
        for layer in (resnet.conv1, resnet.bn1, resnet.layer1):
            for p in layer.parameters():
                p.requires_grad = False
        for m in self._feature_extractor.modules():
            if type(m) == nn.BatchNorm2d:
                for p in m.parameters():
                    p.requires_grad = False
        self._packed_key = None
        self._packed = None
        self.math_mode = "f32"
        self.x6_conv1x1 = False      # the eligible 1x1 convolutions as f32x6 GEMMs (f32_winograd mode)
        self.x3 = False              # ... in the f32x3 arithmetic instead
        self.g3 = False              # every bottleneck convolution in the f32x3 arithmetic under one scale per tensor (pack_block_g3)

    def _g3_active(self):
        """the f32x3-under-a-tensor-scale arithmetic belongs to the `f32_winograd` table like every split-operand layer (x6_conv1x1 / x3):
        `math_mode = "f32"` is the STRICT mode -- every convolution on the exact-f32 pipe"""
        return bool(self.g3) and self.math_mode == "f32_winograd"

    def blocks(self):
        fe = self._feature_extractor
        return [b for layer in (fe[4], fe[5], fe[6]) for b in layer]

    def packed(self):
        """{'stem': (w, b), 'blocks': [dict]} of BN-folded packed weights, rebuilt when parameters change."""
        fe = self._feature_extractor
        params = [fe[0].weight] + _bn_params(fe[1]) + [p for b in self.blocks() for p in block_params(b)]
        key = (self.math_mode, self.x6_conv1x1, self.x3, self.g3) + rt.param_key(params)
        if key != self._packed_key:
            sw, sb, keep = fold_conv_bn(fe[0], fe[1])
            self._packed = {"stem": (sw, sb), "keep": keep,
                            "blocks": [pack_block(b, self.math_mode, single_map=True, x6=self.x6_conv1x1, x3=self.x3, g3=self._g3_active()) for b in self.blocks()],
                            "n_blocks": [len(fe[4]), len(fe[5]), len(fe[6])]}
            self._packed_key = key
        return self._packed

    def forward(self, image_data):
        """image_data (1,3,H,W) float32 CUDA -> (1, 1024, ceil(H/16), ceil(W/16))."""
        assert image_data.shape[0] == 1, "Batch size must be 1"
        x = rt.as_f32_cuda(image_data, "image_data")
        pk = self.packed()
        h, w = int(x.shape[2]), int(x.shape[3])
        lib = nv.lib()
        h1, w1 = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = t.empty((h1, w1, 64), dtype=t.float32, device=x.device)
        with t.cuda.device(x.device):
            nv.check(lib.frcnn_conv7x7_s2_c3(nv.ptr(x), nv.ptr(pk["stem"][0]), nv.ptr(pk["stem"][1]), nv.ptr(y), h, w, 64,
                                             nv.RELU, nv.stream_ptr()), "frcnn_conv7x7_s2_c3")
            h2, w2 = (h1 - 1) // 2 + 1, (w1 - 1) // 2 + 1
            cur = t.empty((1, h2, w2, 64), dtype=t.float32, device=x.device)
            nv.check(lib.frcnn_maxpool3x3_s2_nhwc(nv.ptr(y), nv.ptr(cur), h1, w1, 64, nv.stream_ptr()), "frcnn_maxpool3x3_s2_nhwc")
        h, w = h2, w2
        for pb in pk["blocks"]:
            cur, h, w = run_block(cur, 1, h, w, pb)
        return cur[0].permute(2, 0, 1).unsqueeze(0)


class PoolToFeatureVector(nn.Module):
    def __init__(self, resnet):
        super().__init__()
        self._layer4 = resnet.layer4
        for m in self._layer4.modules():
            if type(m) == nn.BatchNorm2d:
                for p in m.parameters():
                    p.requires_grad = False
        self._packed_key = None
        self._packed = None
        self.math_mode = "f32"
        self.x6_conv1x1 = False      # the eligible 1x1 convolutions as f32x6 GEMMs (f32_winograd mode)
        self.x3 = False              # ... in the f32x3 arithmetic instead
        self.g3 = False              # pack_block_g3

    def _g3_active(self):
        """the f32x3-under-a-tensor-scale arithmetic belongs to the `f32_winograd` table like every split-operand layer (x6_conv1x1 / x3):
        `math_mode = "f32"` is the STRICT mode -- every convolution on the exact-f32 pipe"""
        return bool(self.g3) and self.math_mode == "f32_winograd"

    def packed(self):
        params = [p for b in self._layer4 for p in block_params(b)]
        key = (self.math_mode, self.x6_conv1x1, self.x3, self.g3) + rt.param_key(params)
        if key != self._packed_key:
            self._packed = [pack_block(b, self.math_mode, x6=self.x6_conv1x1, x3=self.x3, g3=self._g3_active()) for b in self._layer4]
            self._packed_key = key
        return self._packed

    def forward(self, rois):
        """rois (N, 1024, 7, 7) -> layer4 -> (N, 2048, 4, 4) -> mean over x then y -> (N, 2048)."""
        x = rt.as_f32_cuda(rois, "rois")
        n = int(x.shape[0])
        if n == 0:
            return t.empty((0, 2048), dtype=t.float32, device=x.device)
        cur = x.permute(0, 2, 3, 1).contiguous()
        h, w = int(x.shape[2]), int(x.shape[3])
        for pb in self.packed():
            cur, h, w = run_block(cur, n, h, w, pb)
        c = int(cur.shape[3])
        y = t.empty((n, c), dtype=t.float32, device=x.device)
        with t.cuda.device(x.device):
            nv.check(nv.lib().frcnn_spatial_mean_nhwc(nv.ptr(cur), nv.ptr(y), n, h, w, c, nv.stream_ptr()),
                     "frcnn_spatial_mean_nhwc")
        return y


class ResNetBackbone(Backbone):
    def __init__(self, architecture):
        super().__init__()
        # Backbone properties (resnet.py:138-141)
        self.feature_map_channels = 1024
        self.feature_pixels = 16
        self.feature_vector_size = 2048
        self.image_preprocessing_params = image.PreprocessingParams(
            channel_order=image.ChannelOrder.RGB, scaling=1.0 / 255.0, means=[0.485, 0.456, 0.406], stds=[0.229, 0.224, 0.225])
        if architecture not in _BLOCKS:
            raise ValueError("Invalid ResNet architecture value: %s" % getattr(architecture, "value", architecture))
        self.architecture = architecture
        resnet = _ResNetParams(_BLOCKS[architecture])
        self.feature_extractor = FeatureExtractor(resnet=resnet)
        self.pool_to_feature_vector = PoolToFeatureVector(resnet=resnet)

    def compute_feature_map_shape(self, image_shape):
        """(1024, ceil(H/16), ceil(W/16)) -- resnet.py:161-185."""
        image_width = image_shape[-1]
        image_height = image_shape[-2]
        return (self.feature_map_channels, ceil(image_height / self.feature_pixels), ceil(image_width / self.feature_pixels))
