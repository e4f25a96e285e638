"""
GPU parity tests of the Winograd F(2x2,3x3) float32 layers (csrc/winograd.hip) and of the
"f32_winograd" math mode of the fused forward.

Tolerances (float32 path; north_star: boxes within 1e-3 px of the PyTorch reference):
  filter transform      : 1e-7 relative to numpy float64 G g G^T rounded to float32
  one layer             : error against a float64 convolution <= 5x the direct exact-f32 kernel's own
                          error + 3e-6 of max|y| (Winograd adds a few fp32 roundings per output)
  end to end            : the SAME thresholds as the default mode in tests/test_model_gpu.py
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fasterrcnn_amd import _native as nv
from fasterrcnn_amd import synthetic
from oracle import frcnn_oracle as O

pytestmark = pytest.mark.gpu

CASES = [("600x1000_s0", True), ("224x320_s3", True), ("333x517_s5_noedge", False)]
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)


def load_case(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "vgg16_%s.npz" % tag))
    img = synthetic.image(int(g["seed"]), int(g["height"]), int(g["width"])).unsqueeze(0)
    return g, img


def iou_matrix(a, b):
    tl = np.maximum(a[:, None, 0:2], b[None, :, 0:2])
    br = np.minimum(a[:, None, 2:4], b[None, :, 2:4])
    wh = np.clip(br - tl, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    aa = np.prod(a[:, 2:4] - a[:, 0:2], axis=1)
    ab = np.prod(b[:, 2:4] - b[:, 0:2], axis=1)
    return inter / (aa[:, None] + ab[None, :] - inter + 1e-12)


def match_rows(ours, ref):
    if len(ref) == 0 or len(ours) == 0:
        return np.zeros((0,), int), np.full((len(ref),), np.inf)
    j = iou_matrix(ref[:, :4].astype(np.float64), ours[:, :4].astype(np.float64)).argmax(axis=1)
    return j, np.abs(ours[j, :4] - ref[:, :4]).max(axis=1)


def run_layer(kind, x, w_oihw, b, relu, pool, scale=None):
    """kind: 'direct' | 'winograd'; x NHWC CUDA ([H][W][C] or [N][H][W][C] for winograd), returns NHWC CUDA."""
    lib = nv.lib()
    n = int(x.shape[0]) if x.dim() == 4 else 1
    h, wd, cin = (int(v) for v in x.shape[-3:])
    cout = int(w_oihw.shape[0])
    s = nv.stream_ptr()
    oh, ow = (h // 2, wd // 2) if pool else (h, wd)
    y = torch.full(((n, oh, ow, cout) if x.dim() == 4 else (oh, ow, cout)), float("nan"), device=x.device)
    flags = (nv.RELU if relu else 0) | (nv.POOL2 if pool else 0)
    if kind == "winograd":
        u = torch.empty((16, cout, cin), device=x.device)
        nv.check(lib.frcnn_pack_conv3x3_winograd(nv.ptr(w_oihw), nv.ptr(scale), nv.ptr(u), cout, cin, s), "pack_winograd")
        wsb = int(lib.frcnn_conv3x3_winograd_workspace_bytes(n, h, wd, cin, cout))
        assert wsb == 16 * n * ((h + 1) // 2) * ((wd + 1) // 2) * (cin + cout) * 4
        ws = torch.empty((wsb // 4,), device=x.device)
        nv.check(lib.frcnn_conv3x3_nhwc_winograd(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), n, h, wd, cin, cout, flags,
                                                 nv.ptr(ws), wsb, s), "conv_winograd")
    else:
        wp = torch.empty((9, cout, cin), device=x.device)
        nv.check(lib.frcnn_pack_conv3x3(nv.ptr(w_oihw), nv.ptr(wp), cout, cin, s), "pack")
        wsb = int(lib.frcnn_conv3x3_workspace_bytes(h, wd, cin, cout))
        ws = torch.empty((max(wsb, 4) // 4,), device=x.device)
        nv.check(lib.frcnn_conv3x3_nhwc(nv.ptr(x), nv.ptr(wp), nv.ptr(b), nv.ptr(y), h, wd, cin, cout, flags,
                                        nv.ptr(ws), wsb, s), "conv")
    torch.cuda.synchronize()
    return y


def test_filter_transform_matches_numpy():
    lib = nv.lib()
    rng = np.random.RandomState(5)
    g = rng.randn(128, 48, 3, 3).astype(np.float32)
    gd = torch.from_numpy(g).cuda()
    u = torch.empty((16, 128, 48), device="cuda")
    nv.check(lib.frcnn_pack_conv3x3_winograd(nv.ptr(gd), None, nv.ptr(u), 128, 48, nv.stream_ptr()), "pack_winograd")
    torch.cuda.synchronize()
    ref = np.einsum("ia,kcab,jb->ijkc", G, g.astype(np.float64), G).reshape(16, 128, 48)
    got = u.cpu().numpy().astype(np.float64)
    assert np.abs(got - ref).max() <= 1e-7 * np.abs(ref).max()
    # positions (0,0), (0,3), (3,0), (3,3) are plain copies of the corner taps
    assert np.array_equal(u[0].cpu().numpy(), g[:, :, 0, 0]) and np.array_equal(u[15].cpu().numpy(), g[:, :, 2, 2])
    assert np.array_equal(u[3].cpu().numpy(), g[:, :, 0, 2]) and np.array_equal(u[12].cpu().numpy(), g[:, :, 2, 0])


@pytest.mark.parametrize("h,w,cin,cout,relu,pool", [
    (37, 62, 512, 512, True, False),      # block 5 / RPN trunk
    (75, 125, 256, 512, True, False),     # conv4_1: odd height and width
    (75, 125, 512, 512, True, True),      # conv4_3 with the fused pool (floor: 37 x 62)
    (150, 250, 256, 256, True, True),     # conv3_3
    (9, 11, 256, 128, False, False),      # tiny, odd, no ReLU (negative values must survive)
    (2, 2, 16, 128, True, True),          # a single tile, a single pooled pixel
    (1, 5, 32, 128, True, False),         # one row
    (14, 14, 64, 256, False, True),
])
def test_layer_against_float64_and_direct(h, w, cin, cout, relu, pool):
    gen = torch.Generator().manual_seed(h * 1000 + w)
    x = torch.randn((h, w, cin), generator=gen)
    wt = torch.randn((cout, cin, 3, 3), generator=gen) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn((cout,), generator=gen) * 0.1
    ref = F.conv2d(x.permute(2, 0, 1).unsqueeze(0).double(), wt.double(), b.double(), padding=1)
    if relu:
        ref = F.relu(ref)
    if pool:
        ref = F.max_pool2d(ref, 2, 2)
    ref = ref[0].permute(1, 2, 0).numpy()
    xd, wd_, bd = x.cuda(), wt.cuda(), b.cuda()
    yw = run_layer("winograd", xd, wd_, bd, relu, pool).cpu().numpy().astype(np.float64)
    assert yw.shape == ref.shape and np.isfinite(yw).all()          # every output written
    scale = max(float(np.abs(ref).max()), 1e-30)
    ew = float(np.abs(yw - ref).max()) / scale
    if cin % 16 == 0 and cout % 64 == 0 and h >= 2 and w >= 2:
        yd = run_layer("direct", xd, wd_, bd, relu, pool).cpu().numpy().astype(np.float64)
        ed = float(np.abs(yd - ref).max()) / scale
    else:
        ed = 0.0
    print("winograd %dx%d %d->%d pool=%d: max err / max|y| = %.3g (direct kernel %.3g)" % (h, w, cin, cout, pool, ew, ed))
    assert ew <= 5 * ed + 3e-6
    # run-to-run identical
    yw2 = run_layer("winograd", xd, wd_, bd, relu, pool).cpu().numpy().astype(np.float64)
    assert np.array_equal(yw, yw2)


def test_batched_maps_with_folded_batchnorm_scale():
    """The ResNet layer4 use: 3x3 512->512 on a batch of per-RoI 4x4 maps, filter rows pre-multiplied by the frozen-BN scale."""
    gen = torch.Generator().manual_seed(77)
    n, h, w, cin, cout = 7, 4, 4, 512, 512
    x = torch.randn((n, h, w, cin), generator=gen)
    wt = torch.randn((cout, cin, 3, 3), generator=gen) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn((cout,), generator=gen) * 0.1
    scale = torch.rand((cout,), generator=gen) + 0.5
    folded = (wt * scale[:, None, None, None]).double()               # the float32 product, as the pack forms it
    ref = F.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), folded, b.double(), padding=1)).permute(0, 2, 3, 1).numpy()
    y = run_layer("winograd", x.cuda(), wt.cuda(), b.cuda(), True, False, scale=scale.cuda()).cpu().numpy().astype(np.float64)
    assert y.shape == ref.shape and np.isfinite(y).all()
    err = float(np.abs(y - ref).max()) / float(np.abs(ref).max())
    print("winograd batched %dx%dx%d %d->%d with BN scale: max err / max|y| = %.3g" % (n, h, w, cin, cout, err))
    assert err <= 3e-6
    # odd map size in a batch (7x7), pooled output
    x2 = torch.randn((3, 7, 7, 128), generator=gen)
    w2 = torch.randn((256, 128, 3, 3), generator=gen) * 0.03
    b2 = torch.zeros((256,))
    ref2 = F.max_pool2d(F.conv2d(x2.permute(0, 3, 1, 2).double(), w2.double(), None, padding=1), 2, 2).permute(0, 2, 3, 1).numpy()
    y2 = run_layer("winograd", x2.cuda(), w2.cuda(), b2.cuda(), False, True).cpu().numpy().astype(np.float64)
    assert y2.shape == ref2.shape == (3, 3, 3, 256)
    assert float(np.abs(y2 - ref2).max()) / float(np.abs(ref2).max()) <= 3e-6


def test_rejects_unsupported_shapes():
    lib = nv.lib()
    assert lib.frcnn_conv3x3_winograd_workspace_bytes(1, 8, 8, 24, 128) == 0        # cin % 16
    assert lib.frcnn_conv3x3_winograd_workspace_bytes(1, 8, 8, 32, 64) == 0         # cout % 128
    x = torch.zeros((8, 8, 32), device="cuda")
    u = torch.zeros((16, 64, 32), device="cuda")
    b = torch.zeros((64,), device="cuda")
    y = torch.zeros((8, 8, 64), device="cuda")
    ws = torch.zeros((1 << 16,), device="cuda")
    rc = lib.frcnn_conv3x3_nhwc_winograd(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), 1, 8, 8, 32, 64, 0, nv.ptr(ws), 1 << 18,
                                         nv.stream_ptr())
    assert rc == -4
    u2 = torch.zeros((16, 128, 32), device="cuda")
    b2 = torch.zeros((128,), device="cuda")
    y2 = torch.zeros((8, 8, 128), device="cuda")
    rc = lib.frcnn_conv3x3_nhwc_winograd(nv.ptr(x), nv.ptr(u2), nv.ptr(b2), nv.ptr(y2), 1, 8, 8, 32, 128, 0, nv.ptr(ws), 64,
                                         nv.stream_ptr())
    assert rc == -1                                                                # scratch too small


@pytest.fixture(scope="module")
def oracle_runs(golden_dir, sd_cpu):
    runs = {}
    for tag, allow_edge in CASES:
        g, img = load_case(golden_dir, tag)
        detail = {}
        O.forward(sd_cpu, img, allow_edge_proposals=allow_edge, detail=detail)
        runs[tag] = detail
    return runs


@pytest.fixture(scope="module")
def models(sd_cpu):
    from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
    from fasterrcnn_amd.models.vgg16 import VGG16Backbone
    out = {}
    for edge in (True, False):
        m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0), allow_edge_proposals=edge)
        m.load_state_dict(sd_cpu, strict=True)
        m = m.cuda().eval()
        m.math_mode = "f32_winograd"
        out[edge] = m
    return out


@pytest.mark.parametrize("tag,allow_edge", CASES)
def test_winograd_mode_end_to_end(models, golden_dir, oracle_runs, tag, allow_edge):
    """The thresholds of tests/test_model_gpu.py (default mode), applied to the f32_winograd mode."""
    g, img = load_case(golden_dir, tag)
    model = models[allow_edge]
    assert model.math_mode == "f32_winograd"
    fm = model._stage1_feature_extractor(image_data=img.cuda()).cpu()
    ref = oracle_runs[tag]["feature_map"]
    err = float((fm - ref).abs().max()) / float(ref.abs().max())
    print("winograd feature map %s: max rel err %.3g" % (tag, err))
    assert err <= 2e-5
    props, classes, deltas = model(image_data=img.cuda())
    j, e = match_rows(props.cpu().numpy(), g["proposals"])
    ok = e <= 1e-3
    print("winograd forward %s: %.1f%% of the reference's proposals within 1e-3 px" % (tag, 100 * ok.mean()))
    assert props.shape[0] == g["proposals"].shape[0] and ok.mean() >= 0.99      # the held-out sweep's floor (tests/test_model_gpu.py)
    assert np.abs(classes.cpu().numpy()[j[ok]] - g["classes"][ok]).max() <= 1e-4
    assert np.abs(deltas.cpu().numpy()[j[ok]] - g["box_deltas"][ok]).max() <= 1e-3
    det = model.predict(image_data=img.cuda(), score_threshold=float(g["score_threshold"]))
    assert sorted(det.keys()) == list(range(1, 21))
    refd = g["detections"]
    n_ok = 0
    for c in range(1, 21):
        r = refd[refd[:, 0] == c][:, 1:]
        if len(r) and len(det[c]):
            jj, ee = match_rows(det[c], r)
            n_ok += int(((ee <= 1e-3) & (np.abs(det[c][jj, 4] - r[:, 4]) <= 1e-4)).sum())
    n_ours = sum(len(v) for v in det.values())
    print("winograd predict %s: %d/%d reference detections reproduced within 1e-3 px / 1e-4 score (ours: %d rows)" % (
        tag, n_ok, len(refd), n_ours))
    assert n_ok >= 0.99 * len(refd) and n_ours == len(refd)   # the held-out sweep's floor; no extra, no missing row


def test_winograd_mode_is_deterministic_and_layerwise_equals_fused(models):
    model = models[True]
    img = synthetic.image(11, 352, 480).unsqueeze(0).cuda()
    a = model(image_data=img)
    b = model(image_data=img)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    # the layer-wise Python path (FeatureExtractor.forward) runs the same kernels as the fused C entry point
    fm_layerwise = model._stage1_feature_extractor(image_data=img)
    model(image_data=img)
    fused = model.context(0).tensor(0)                                   # feature map of the last fused forward, NHWC
    assert fused.numel() == fm_layerwise.numel()
    assert torch.equal(fm_layerwise[0].permute(1, 2, 0).contiguous().reshape(-1), fused)
    # async slots (throughput regime) give the same detections as the synchronous call
    d0 = model.predict(image_data=img, score_threshold=0.05)
    d1 = model.predict_async(img, 0.05, slot=1).result()
    same = total = 0
    for c in d0:
        total += len(d0[c])
        if len(d0[c]) and len(d1[c]):
            jj, ee = match_rows(d1[c], d0[c])
            same += int((ee <= 1e-3).sum())
    assert total > 0 and same >= 0.97 * total


def test_mode_switch_repacks_and_restores(models, sd_cpu):
    model = models[True]
    img = synthetic.image(4, 224, 320).unsqueeze(0).cuda()
    w = model(image_data=img)
    model.math_mode = "f32"
    try:
        d = model(image_data=img)
    finally:
        model.math_mode = "f32_winograd"
    w2 = model(image_data=img)
    for u, v in zip(w, w2):
        assert torch.equal(u, v)
    # the two modes agree to fp32 rounding on the class scores of matching proposals
    j, e = match_rows(w[0].cpu().numpy(), d[0].cpu().numpy())
    ok = e <= 1e-3
    assert ok.mean() >= 0.95
    assert np.abs(w[1].cpu().numpy()[j[ok]] - d[1].cpu().numpy()[ok]).max() <= 1e-4
