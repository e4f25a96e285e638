"""
VGG-16 backbone, mirroring pytorch/FasterRCNN/models/vgg16.py:22-158: same class names, layer
attribute names (-> identical state_dict keys: `_block{1..5}_conv{1..3}.{weight,bias}`,
`_fc{1,2}.{weight,bias}`), same Backbone properties.  nn.Conv2d / nn.Linear objects are used as
PARAMETER HOLDERS only; the arithmetic runs in csrc/conv.hip and csrc/linear.hip.
"""
import torch as t
from torch import nn

from .. import _native as nv
from .. import runtime as rt
from ..datasets import image
from .backbone import Backbone

_LAYERS = [  # (name, cin, cout, pool_after)
    ("_block1_conv1", 3, 64, False), ("_block1_conv2", 64, 64, True),
    ("_block2_conv1", 64, 128, False), ("_block2_conv2", 128, 128, True),
    ("_block3_conv1", 128, 256, False), ("_block3_conv2", 256, 256, False), ("_block3_conv3", 256, 256, True),
    ("_block4_conv1", 256, 512, False), ("_block4_conv2", 512, 512, False), ("_block4_conv3", 512, 512, True),
    ("_block5_conv1", 512, 512, False), ("_block5_conv2", 512, 512, False), ("_block5_conv3", 512, 512, False),
]


def pack_conv3x3(conv, math_mode="f32", one_launch=False):
    """
    OIHW weight of a 3x3 nn.Conv2d -> tap-major [9][cout][cin] (or [27][cout] when cin == 3); in the
    "f32_winograd" mode, for cin >= 64 (% 16) and cout >= 64 (% 64) -> the transformed filters G g G^T in the one-launch
    kernel's [cin/16][cout/64][16][64][16] order (flat); "f32_winograd_x6" (a per-layer choice of the f32_winograd mode, round 3)
    -> the same bank as x6t records (uint8) for csrc/wino_x6.hip; "f32_winograd_3launch" (tests / experiments) keeps round 1's
    [16][cout][cin] bank of the three-launch form for cin >= 128 and cout >= 256.
    """
    w = conv.weight.detach()
    cout, cin = int(w.shape[0]), int(w.shape[1])
    w = rt.as_f32_cuda(w, "conv weight")
    if math_mode == "f32_winograd_x6":
        # x6 Winograd layer (csrc/wino_x6.hip): the transformed filter bank as x6t records, uint8 (dtype marks it)
        if not nv.uses_winograd_x6(cin, cout):
            raise ValueError("a %d -> %d 3x3 layer cannot run as an x6 Winograd layer (cin >= 256, cout %% 256 == 0)" % (cin, cout))
        lib = nv.lib()
        out = t.empty((int(lib.frcnn_conv3x3_winograd_x6_pack_bytes(cout, cin)),), dtype=t.uint8, device=w.device)
        with t.cuda.device(w.device):
            nv.check(lib.frcnn_pack_conv3x3_winograd_x6(nv.ptr(w), None, nv.ptr(out), cout, cin, nv.stream_ptr()),
                     "frcnn_pack_conv3x3_winograd_x6")
        return out
    if math_mode == "f32_winograd_x3":
        # the same layer in the f32x3 arithmetic (csrc/wino_x3.hip): the float32 bank [16][cout][cin], then records + row scales in one blob
        # (int8: the dtype marks it)
        # (`one_launch`: the blob feeds csrc/wino_x3f.hip, whose shape rule is wider: cin % 32 == 0, cout % 64 == 0)
        if not (nv.uses_winograd_x3f(cin, cout) if one_launch else nv.uses_winograd_x6(cin, cout)):
            raise ValueError("a %d -> %d 3x3 layer cannot run as an x6 / x3 Winograd layer (cin >= 256, cout %% 256 == 0; one-launch x3: cin %% 32 == 0, "
                             "cout %% 64 == 0)" % (cin, cout))
        lib = nv.lib()
        bank = t.empty((16, cout, cin), dtype=t.float32, device=w.device)
        out = t.empty((int(lib.frcnn_conv3x3_winograd_x3_pack_bytes(cout, cin)),), dtype=t.int8, device=w.device)
        with t.cuda.device(w.device):
            nv.check(lib.frcnn_pack_conv3x3_winograd(nv.ptr(w), None, nv.ptr(bank), cout, cin, nv.stream_ptr()), "frcnn_pack_conv3x3_winograd")
            nv.check(lib.frcnn_pack_conv3x3_winograd_x3(nv.ptr(bank), nv.ptr(out), cout, cin, nv.stream_ptr()), "frcnn_pack_conv3x3_winograd_x3")
        return out
    if math_mode == "f32_winograd" and nv.uses_winograd_fused(cin, cout):
        # one-launch Winograd layer (csrc/winofused.hip): [cin/16][cout/64][16][64][16], kept flat (dim() == 1 marks it)
        out = t.empty((16 * cout * cin,), dtype=t.float32, device=w.device)
        with t.cuda.device(w.device):
            nv.check(nv.lib().frcnn_pack_conv3x3_winograd_fused(nv.ptr(w), None, nv.ptr(out), cout, cin, nv.stream_ptr()),
                     "frcnn_pack_conv3x3_winograd_fused")
        return out
    if math_mode == "f32_winograd_3launch" and nv.uses_winograd(cin, cout):
        out = t.empty((16, cout, cin), dtype=t.float32, device=w.device)
        with t.cuda.device(w.device):
            nv.check(nv.lib().frcnn_pack_conv3x3_winograd(nv.ptr(w), None, nv.ptr(out), cout, cin, nv.stream_ptr()),
                     "frcnn_pack_conv3x3_winograd")
        return out
    out = t.empty((27, cout) if cin == 3 else (9, cout, cin), dtype=t.float32, device=w.device)
    with t.cuda.device(w.device):
        if cin == 3:
            nv.check(nv.lib().frcnn_pack_conv3x3_c3(nv.ptr(w), nv.ptr(out), cout, nv.stream_ptr()), "frcnn_pack_conv3x3_c3")
        else:
            nv.check(nv.lib().frcnn_pack_conv3x3(nv.ptr(w), nv.ptr(out), cout, cin, nv.stream_ptr()), "frcnn_pack_conv3x3")
    return out


def conv3x3(x_hwc, wp, b, cin, cout, relu=True, pool=False, one_launch=False):
    """One 3x3 'same' convolution (+ReLU, + fused 2x2 max-pool) on an NHWC CUDA tensor via frcnn_conv3x3_nhwc
    (frcnn_conv3x3_nhwc_winograd when `wp` is a [16][cout][cin] transformed filter bank, the x6 / x3 Winograd entry points when it is a
    uint8 record bank / blob)."""
    h, w = int(x_hwc.shape[0]), int(x_hwc.shape[1])
    oh, ow = (h // 2, w // 2) if pool else (h, w)
    y = t.empty((oh, ow, cout), dtype=t.float32, device=x_hwc.device)
    lib = nv.lib()
    if wp.dtype == t.uint8:                              # x6 Winograd layer: record bank + scratch (V records, M, split-K partials)
        flags = (nv.RELU if relu else 0) | (nv.POOL2 if pool else 0)
        ws_bytes = int(lib.frcnn_conv3x3_winograd_x6_workspace_bytes(1, h, w, cin, cout))
        ws = t.empty((ws_bytes,), dtype=t.uint8, device=x_hwc.device)
        with t.cuda.device(x_hwc.device):
            nv.check(lib.frcnn_conv3x3_nhwc_winograd_x6(nv.ptr(x_hwc), nv.ptr(wp), nv.ptr(b), nv.ptr(y), 1, h, w, cin, cout, flags,
                                                        nv.ptr(ws), ws_bytes, nv.stream_ptr()), "frcnn_conv3x3_nhwc_winograd_x6")
        return y
    if wp.dtype == t.int8 and one_launch:                # one-launch x3 Winograd layer (csrc/wino_x3f.hip): scratch = the input's channel maxima
        flags = (nv.RELU if relu else 0) | (nv.POOL2 if pool else 0)
        ws_bytes = int(lib.frcnn_conv3x3_winograd_x3_fused_workspace_bytes(1, h, w))
        ws = t.empty((ws_bytes,), dtype=t.uint8, device=x_hwc.device)
        with t.cuda.device(x_hwc.device):
            nv.check(lib.frcnn_conv3x3_nhwc_winograd_x3_fused(nv.ptr(x_hwc), nv.ptr(wp), nv.ptr(b), nv.ptr(y), 1, h, w, cin, cout, flags,
                                                              nv.ptr(ws), ws_bytes, nv.stream_ptr()), "frcnn_conv3x3_nhwc_winograd_x3_fused")
        return y
    if wp.dtype == t.int8:                               # x3 Winograd layer: packed x3t bank + scratch
        flags = (nv.RELU if relu else 0) | (nv.POOL2 if pool else 0)
        ws_bytes = int(lib.frcnn_conv3x3_winograd_x3_workspace_bytes(1, h, w, cin, cout))
        ws = t.empty((ws_bytes,), dtype=t.uint8, device=x_hwc.device)
        with t.cuda.device(x_hwc.device):
            nv.check(lib.frcnn_conv3x3_nhwc_winograd_x3(nv.ptr(x_hwc), nv.ptr(wp), nv.ptr(b), nv.ptr(y), 1, h, w, cin, cout, flags,
                                                        nv.ptr(ws), ws_bytes, nv.stream_ptr()), "frcnn_conv3x3_nhwc_winograd_x3")
        return y
    if wp.dtype == t.float32 and wp.dim() == 1:          # one-launch Winograd bank
        flags = (nv.RELU if relu else 0) | (nv.POOL2 if pool else 0)
        with t.cuda.device(x_hwc.device):
            nv.check(lib.frcnn_conv3x3_nhwc_winograd_fused(nv.ptr(x_hwc), nv.ptr(wp), nv.ptr(b), nv.ptr(y), h, w, cin, cout, flags,
                                                           nv.stream_ptr()), "frcnn_conv3x3_nhwc_winograd_fused")
        return y
    winograd = wp.dtype == t.float32 and wp.dim() == 3 and int(wp.shape[0]) == 16
    if winograd:
        ws_bytes = int(lib.frcnn_conv3x3_winograd_workspace_bytes(1, h, w, cin, cout))
    else:
        ws_bytes = int(lib.frcnn_conv3x3_workspace_bytes(h, w, cin, cout))
    ws = t.empty((max(ws_bytes, 4) // 4,), dtype=t.float32, device=x_hwc.device)
    flags = (nv.RELU if relu else 0) | (nv.POOL2 if pool else 0)
    with t.cuda.device(x_hwc.device):
        if winograd:
            nv.check(lib.frcnn_conv3x3_nhwc_winograd(nv.ptr(x_hwc), nv.ptr(wp), nv.ptr(b), nv.ptr(y), 1, h, w, cin, cout, flags,
                                                     nv.ptr(ws), ws_bytes, nv.stream_ptr()), "frcnn_conv3x3_nhwc_winograd")
            return y
        nv.check(lib.frcnn_conv3x3_nhwc(nv.ptr(x_hwc), nv.ptr(wp), nv.ptr(b), nv.ptr(y), h, w, cin, cout, flags,
                                        nv.ptr(ws), ws_bytes, nv.stream_ptr()), "frcnn_conv3x3_nhwc")
    return y


class FeatureExtractor(nn.Module):
    def __init__(self):
        super().__init__()
        for name, cin, cout, _ in _LAYERS:
            setattr(self, name, nn.Conv2d(in_channels=cin, out_channels=cout, kernel_size=(3, 3), stride=1, padding="same"))
        # Freeze first two convolutional blocks (vgg16.py:49-58) -- no effect on inference
        for name in ("_block1_conv1", "_block1_conv2", "_block2_conv1", "_block2_conv2"):
            getattr(self, name).weight.requires_grad = False
            getattr(self, name).bias.requires_grad = False
        self._packed_key = None
        self._packed = None
        self.math_mode = "f32"
        self.x6_layers = ()          # names ("conv4_1", ...) of the layers that run as x6 Winograd layers in the f32_winograd mode
        self.x3_layers = ()          # the subset of x6_layers whose GEMMs run in the f32x3 arithmetic (csrc/wino_x3.hip)
        self.x3f_layers = ()         # layers (disjoint from x6_layers) that run as ONE-launch f32x3 Winograd layers (csrc/wino_x3f.hip)

    @staticmethod
    def layer_name(i):
        return "conv%s_%s" % (_LAYERS[i][0][6], _LAYERS[i][0][-1])

    def layer_math(self, i):
        """Pack / arithmetic kind of layer i in the current mode."""
        name = self.layer_name(i)
        if self.math_mode == "f32_winograd" and name in self.x3f_layers:
            return "f32_winograd_x3"
        if self.math_mode == "f32_winograd" and name in self.x6_layers:
            return "f32_winograd_x3" if name in self.x3_layers else "f32_winograd_x6"
        return self.math_mode

    def convs(self):
        return [getattr(self, name) for name, _, _, _ in _LAYERS]

    def packed(self):
        """[(packed_weight, bias)] x 13 on the parameters' device, rebuilt when parameters change."""
        params = [p for c in self.convs() for p in (c.weight, c.bias)]
        key = (self.math_mode, tuple(sorted(self.x6_layers)), tuple(sorted(self.x3_layers)), tuple(sorted(self.x3f_layers))) + rt.param_key(params)
        if key != self._packed_key:
            x3f = self.x3f_layers if self.math_mode == "f32_winograd" else ()
            self._packed = [(pack_conv3x3(c, self.layer_math(i), one_launch=self.layer_name(i) in x3f), rt.as_f32_cuda(c.bias.detach(), "conv bias"))
                            for i, c in enumerate(self.convs())]
            self._packed_key = key
        return self._packed

    def packed_direct(self):
        """Fresh tap-major [9][cout][cin] packs whatever the inference math mode is (the train step's masters)."""
        return [(pack_conv3x3(c, "f32"), rt.as_f32_cuda(c.bias.detach(), "conv bias").clone()) for c in self.convs()]

    def forward(self, image_data):
        """
        image_data (1, 3, H, W) float32 CUDA -> feature map (1, 512, H // 16, W // 16), as the
        reference returns it (NCHW view of the NHWC result).
        """
        assert image_data.shape[0] == 1, "Batch size must be 1"
        x = rt.as_f32_cuda(image_data, "image_data")
        packed = self.packed()
        lib = nv.lib()
        h, w = int(x.shape[2]), int(x.shape[3])
        cur = None
        for i, (name, cin, cout, pool) in enumerate(_LAYERS):
            wp, b = packed[i]
            if i == 0:
                cur = t.empty((h, w, cout), dtype=t.float32, device=x.device)
                with t.cuda.device(x.device):
                    nv.check(lib.frcnn_conv3x3_c3(nv.ptr(x), nv.ptr(wp), nv.ptr(b), nv.ptr(cur), h, w, cout, nv.RELU,
                                                  nv.stream_ptr()), "frcnn_conv3x3_c3")
            else:
                cur = conv3x3(cur, wp, b, cin, cout, relu=True, pool=pool,
                              one_launch=self.math_mode == "f32_winograd" and self.layer_name(i) in self.x3f_layers)
        return cur.permute(2, 0, 1).unsqueeze(0)


class PoolToFeatureVector(nn.Module):
    def __init__(self, dropout_probability):
        super().__init__()
        self._fc1 = nn.Linear(in_features=512 * 7 * 7, out_features=4096)
        self._fc2 = nn.Linear(in_features=4096, out_features=4096)
        # Dropout is identity at inference; kept so the module tree matches the reference
        self._dropout1 = nn.Dropout(p=dropout_probability)
        self._dropout2 = nn.Dropout(p=dropout_probability)
        self._packed_key = None
        self._packed = None
        self.fc_math_mode = "f32"

    def packed_direct(self):
        """float32 packs whatever the inference arithmetic is (the train step's masters): fc1 with its input dimension permuted
        (C,7,7) -> (7,7,C); fc2 and biases as stored."""
        w1 = rt.as_f32_cuda(self._fc1.weight.detach(), "fc1 weight")
        w1p = t.empty_like(w1)
        with t.cuda.device(w1.device):
            nv.check(nv.lib().frcnn_pack_fc_chw_to_hwc(nv.ptr(w1), nv.ptr(w1p), 4096, 512, 49, nv.stream_ptr()),
                     "frcnn_pack_fc_chw_to_hwc")
        return (w1p, rt.as_f32_cuda(self._fc1.bias.detach(), "fc1 bias"),
                rt.as_f32_cuda(self._fc2.weight.detach(), "fc2 weight"),
                rt.as_f32_cuda(self._fc2.bias.detach(), "fc2 bias"))

    def packed(self, mode=None):
        """(fc1 weight, fc1 bias, fc2 weight, fc2 bias) in the layout of `mode` (default: `fc_math_mode`): float32 matrices ("f32") or
        their x6 / x3 records.  Only the pack of the mode last asked for is kept (0.4 - 0.6 GB each: switching modes must not
        accumulate device memory)."""
        mode = mode or self.fc_math_mode
        params = [self._fc1.weight, self._fc1.bias, self._fc2.weight, self._fc2.bias]
        key = rt.param_key(params)
        if key != self._packed_key or self._packed is None:
            self._packed = {}
            self._packed_key = key
        if mode not in self._packed:
            self._packed = {}                        # drop the other modes' packs BEFORE allocating this one
            w1p, b1, w2, b2 = self.packed_direct()
            if mode == "f32x6":
                w1p, w2 = split_rows_x6t(w1p, 4096), split_rows_x6t(w2, 4096)
            elif mode == "f32x3":
                w1p, w2 = pack_rows_x3t(w1p, 4096), pack_rows_x3t(w2, 4096)
            self._packed[mode] = (w1p, b1, w2, b2)
        return self._packed[mode]

    def forward(self, rois):
        """rois (N, 512, 7, 7) -> (N, 4096): fc1+ReLU, fc2+ReLU (dropout = identity at inference)."""
        if self.training and (self._dropout1.p > 0 or self._dropout2.p > 0):
            raise NotImplementedError("training-mode dropout is outside the inference hot path")
        x = rt.as_f32_cuda(rois, "rois")
        n = int(x.shape[0])
        x = x.permute(0, 2, 3, 1).contiguous().reshape(n, 49 * 512)   # layout plumbing: (C,7,7) -> (7,7,C)
        mode = self.fc_math_mode
        w1p, b1, w2, b2 = self.packed(mode)
        if mode == "f32x6":
            h1 = linear_x6t(x, w1p, b1, 4096, relu=True)
            return linear_x6t(h1, w2, b2, 4096, relu=True)
        if mode == "f32x3":
            h1 = linear_x3t(x, w1p, b1, 4096, relu=True)
            return linear_x3t(h1, w2, b2, 4096, relu=True)
        return linear(linear(x, w1p, b1, 4096, relu=True), w2, b2, 4096, relu=True)


def split_rows_x6t(a, rows_padded):
    """float32 (R, K) CUDA matrix -> its x6t TILE records (uint8, [K/16][rows_padded/32][3][1 KB]: csrc/gemm_x6t.hip); rows beyond R zero."""
    r, k = int(a.shape[0]), int(a.shape[1])
    a = a.contiguous()
    lib = nv.lib()
    rec = t.empty((int(lib.frcnn_x6t_record_bytes(rows_padded, k)),), dtype=t.uint8, device=a.device)
    with t.cuda.device(a.device):
        nv.check(lib.frcnn_split_rows_x6t(nv.ptr(a), k, 0, nv.ptr(rec), r, rows_padded, k, 1, nv.stream_ptr()), "frcnn_split_rows_x6t")
    return rec


def linear_x6t(x, w_rec, b, n_out, relu):
    """y = act(x @ w.T + b) in the f32x6 arithmetic through frcnn_gemm_x6t; x (M, K) float32 CUDA, w_rec = x6t records of w [n_out][K]
    (rows padded to a multiple of 256).  Any M: the activation records are padded to the GEMM's 320-row tiles."""
    m, k = int(x.shape[0]), int(x.shape[1])
    y = t.empty((m, n_out), dtype=t.float32, device=x.device)
    if m == 0:
        return y
    lib = nv.lib()
    mp = (m + nv.X6T_ROW_TILE - 1) // nv.X6T_ROW_TILE * nv.X6T_ROW_TILE
    np_ = (n_out + nv.X6T_COL_TILE - 1) // nv.X6T_COL_TILE * nv.X6T_COL_TILE
    a_rec = split_rows_x6t(x, mp)
    wsb = int(lib.frcnn_gemm_x6t_workspace_bytes(m, n_out, k, 1))
    ws = t.empty((max(wsb, 4),), dtype=t.uint8, device=x.device)
    with t.cuda.device(x.device):
        nv.check(lib.frcnn_gemm_x6t(nv.ptr(a_rec), mp, 0, nv.ptr(w_rec), np_, 0, nv.ptr(b), None, nv.ptr(y), n_out, 0, m, n_out, k, 1,
                                    nv.RELU if relu else 0, nv.ptr(ws), wsb, nv.stream_ptr()), "frcnn_gemm_x6t")
    return y


def pack_rows_x3t(a, rows_padded):
    """float32 (R, K) CUDA matrix -> its packed x3t operand (int8: [K/16][rows_padded/32][2][1 KB] fp16 records of the row-scaled matrix,
    then rows_padded float32 scales 2^-e: csrc/gemm_x3t.hip)."""
    r, k = int(a.shape[0]), int(a.shape[1])
    a = a.contiguous()
    lib = nv.lib()
    blob = t.empty((int(lib.frcnn_x3t_blob_bytes(rows_padded, k, 1)),), dtype=t.int8, device=a.device)
    with t.cuda.device(a.device):
        nv.check(lib.frcnn_pack_rows_x3t(nv.ptr(a), k, 0, nv.ptr(blob), r, rows_padded, k, 1, nv.stream_ptr()), "frcnn_pack_rows_x3t")
    return blob


def linear_x3t(x, w_blob, b, n_out, relu):
    """y = act(x @ w.T + b) in the f32x3 arithmetic through frcnn_gemm_x3t; x (M, K) float32 CUDA, w_blob = pack_rows_x3t of w [n_out][K]
    (rows padded to a multiple of 256).  Any M."""
    m, k = int(x.shape[0]), int(x.shape[1])
    y = t.empty((m, n_out), dtype=t.float32, device=x.device)
    if m == 0:
        return y
    lib = nv.lib()
    mp = (m + nv.X6T_ROW_TILE - 1) // nv.X6T_ROW_TILE * nv.X6T_ROW_TILE
    np_ = (n_out + nv.X6T_COL_TILE - 1) // nv.X6T_COL_TILE * nv.X6T_COL_TILE
    a_blob = pack_rows_x3t(x, mp)
    a_rec, b_rec = int(lib.frcnn_x3t_record_bytes(mp, k)), int(lib.frcnn_x3t_record_bytes(np_, k))
    wsb = int(lib.frcnn_gemm_x3t_workspace_bytes(m, n_out, k, 1))
    ws = t.empty((max(wsb, 4),), dtype=t.uint8, device=x.device)
    with t.cuda.device(x.device):
        nv.check(lib.frcnn_gemm_x3t(nv.ptr(a_blob), a_blob.data_ptr() + a_rec, mp, 0, 0, nv.ptr(w_blob), w_blob.data_ptr() + b_rec, np_, 0, 0,
                                    nv.ptr(b), None, nv.ptr(y), n_out, 0, m, n_out, k, 1, nv.RELU if relu else 0, nv.ptr(ws), wsb,
                                    nv.stream_ptr()), "frcnn_gemm_x3t")
    return y


def linear(x, w, b, n_out, relu):
    """y = act(x @ w[:n_out].T + b) through frcnn_linear; x (M,K) CUDA float32, w row-major [>=n_out][K]."""
    m, k = int(x.shape[0]), int(x.shape[1])
    y = t.empty((m, n_out), dtype=t.float32, device=x.device)
    if m == 0:
        return y
    lib = nv.lib()
    ws_bytes = int(lib.frcnn_linear_workspace_bytes(m, n_out, k))
    ws = t.empty((max(ws_bytes, 4) // 4,), dtype=t.float32, device=x.device)
    with t.cuda.device(x.device):
        nv.check(lib.frcnn_linear(nv.ptr(x), k, nv.ptr(w), nv.ptr(b), nv.ptr(y), n_out, m, n_out, k,
                                  nv.RELU if relu else 0, nv.ptr(ws), ws_bytes, nv.stream_ptr()), "frcnn_linear")
    return y


class VGG16Backbone(Backbone):
    def __init__(self, dropout_probability):
        super().__init__()
        self.feature_map_channels = 512
        self.feature_pixels = 16
        self.feature_vector_size = 4096
        self.image_preprocessing_params = image.PreprocessingParams(
            channel_order=image.ChannelOrder.BGR, scaling=1.0, means=[103.939, 116.779, 123.680], stds=[1, 1, 1])
        self.feature_extractor = FeatureExtractor()
        self.pool_to_feature_vector = PoolToFeatureVector(dropout_probability=dropout_probability)

    def compute_feature_map_shape(self, image_shape):
        image_width = image_shape[-1]
        image_height = image_shape[-2]
        return (self.feature_map_channels, image_height // self.feature_pixels, image_width // self.feature_pixels)
