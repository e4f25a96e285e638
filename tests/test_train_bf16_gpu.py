"""
The reduced-precision train step (BASELINE.json configs[4]): `FasterRCNNModel.grad_math = "bf16"`
runs every gradient GEMM of the step (csrc/gemm_tn.hip: all weight gradients, the data gradients of the dense layers and of the
RPN's 1x1 heads) on the bf16 matrix pipe with operands rounded to bfloat16 and float32 accumulation.  The reference trains in
float32 (faster_rcnn.py:355), so the checker is the restatement in oracle/train_oracle.py (`grad_math="bf16"`: the same GEMMs on
`.bfloat16()`-rounded operands): products of bf16 values are exact in float32, so kernel and oracle may differ only by the
accumulation order -- float32-class tolerances, stated per test.
"""
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fasterrcnn_amd import _native as nv
from fasterrcnn_amd import synthetic
from fasterrcnn_amd import training as T
from fasterrcnn_amd.datasets.training_sample import Box
from oracle import frcnn_oracle as O
from oracle import train_oracle as TO
from test_train_gpu import canonical_grads, canonical_grads_resnet, sample_positions

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
BF16 = nv.GRAD_MATHS["bf16"]


def S():
    return nv.stream_ptr()


def gpu(x):
    return torch.as_tensor(x).to(DEV).contiguous()


def r16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def rel_err(got, truth64):
    return float(np.abs(got.astype(np.float64) - truth64).max()) / max(float(np.abs(truth64).max()), 1e-30)


@pytest.mark.parametrize("M,N,R,lda,ldb", [(128, 4096, 37, 128, 4096), (101, 512, 128, 104, 512), (300, 130, 77, 300, 132),
                                           (2294, 512, 128, 2296, 512), (128, 512, 5000, 128, 512), (7, 6, 3, 8, 8),
                                           (256, 256, 4096, 256, 256), (4096, 25088, 128, 4096, 25088)])
def test_gemm_tn_bf16(M, N, R, lda, ldb):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + R)
    a = torch.randn((R, lda), generator=g)
    b = torch.randn((R, ldb), generator=g)
    ar, br = r16(a[:, :M]), r16(b[:, :N])
    truth = (ar.double().T @ br.double()).numpy()
    yard = rel_err((ar.T @ br).numpy(), truth)                       # float32 torch GEMM of the same rounded operands
    plain = rel_err((a[:, :M].double().T @ b[:, :N].double()).numpy(), truth)
    da, db = gpu(a), gpu(b)
    lib = nv.lib()
    wsb = int(lib.frcnn_gemm_tn_workspace_bytes(M, N, R))
    ws = torch.empty((wsb // 4 + 1,), device=DEV)
    c = torch.full((M, N), float("nan"), device=DEV)
    nv.check(lib.frcnn_gemm_tn_math(nv.ptr(da), lda, nv.ptr(db), ldb, nv.ptr(c), N, M, N, R, BF16, nv.ptr(ws), wsb, S()), "gemm_tn_math")
    # exact products, float32 accumulation: the float32 GEMM's error class -- and far below the rounding of the operands itself
    tol = max(4 * yard, 1.2e-7 * R ** 0.5)
    e = rel_err(c.cpu().numpy(), truth)
    assert e <= tol, (e, yard)
    if R >= 64:
        assert plain > 20 * tol, "the operands' bf16 rounding must dominate: otherwise this test cannot tell the modes apart"
    c2 = torch.empty((M, N), device=DEV)
    nv.check(lib.frcnn_gemm_tn_math(nv.ptr(da), lda, nv.ptr(db), ldb, nv.ptr(c2), N, M, N, R, BF16, None, 0, S()), "gemm_tn_math")
    assert rel_err(c2.cpu().numpy(), truth) <= tol
    c3 = torch.empty((M, N), device=DEV)
    nv.check(lib.frcnn_gemm_tn_math(nv.ptr(da), lda, nv.ptr(db), ldb, nv.ptr(c3), N, M, N, R, BF16, nv.ptr(ws), wsb, S()), "gemm_tn_math")
    assert torch.equal(c, c3), "deterministic"
    # grad_math 0 is the float32 entry point, anything else is refused
    c4 = torch.empty((M, N), device=DEV)
    nv.check(lib.frcnn_gemm_tn_math(nv.ptr(da), lda, nv.ptr(db), ldb, nv.ptr(c4), N, M, N, R, 0, nv.ptr(ws), wsb, S()), "gemm_tn_math")
    c5 = torch.empty((M, N), device=DEV)
    nv.check(lib.frcnn_gemm_tn(nv.ptr(da), lda, nv.ptr(db), ldb, nv.ptr(c5), N, M, N, R, nv.ptr(ws), wsb, S()), "gemm_tn")
    assert torch.equal(c4, c5)
    assert lib.frcnn_gemm_tn_math(nv.ptr(da), lda, nv.ptr(db), ldb, nv.ptr(c4), N, M, N, R, 7, nv.ptr(ws), wsb, S()) == -1


@pytest.mark.parametrize("N,H,W,cin,cout,k,stride,pad", [(1, 20, 33, 64, 64, 3, 1, 1), (1, 37, 62, 128, 64, 3, 1, 1), (1, 9, 7, 16, 128, 3, 1, 1),
                                                         (1, 75, 125, 64, 128, 3, 1, 1), (1, 20, 33, 64, 128, 1, 1, 0),
                                                         (1, 21, 33, 64, 256, 1, 2, 0), (1, 20, 33, 64, 64, 3, 2, 1),
                                                         (5, 7, 7, 128, 64, 3, 2, 1), (128, 4, 4, 64, 256, 1, 1, 0)])
def test_conv_wgrad_bf16(N, H, W, cin, cout, k, stride, pad):
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + k + stride)
    x = torch.randn((N, cin, H, W), generator=g)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dz = torch.randn((N, cout, Ho, Wo), generator=g)
    truth = torch.nn.grad.conv2d_weight(r16(x).double(), (cout, cin, k, k), r16(dz).double(), stride=stride, padding=pad).numpy()
    yard = rel_err(torch.nn.grad.conv2d_weight(r16(x), (cout, cin, k, k), r16(dz), stride=stride, padding=pad).numpy(), truth)
    lib = nv.lib()
    x_n, dz_n = gpu(x.permute(0, 2, 3, 1)), gpu(dz.permute(0, 2, 3, 1))
    dwp = torch.full((k * k, cout, cin), float("nan"), device=DEV)
    wsb = int(lib.frcnn_conv_wgrad_workspace_bytes(N, H, W, cin, cout, k, stride, pad))
    ws = torch.empty((wsb // 4 + 1,), device=DEV)
    nv.check(lib.frcnn_conv_wgrad_math(nv.ptr(x_n), nv.ptr(dz_n), nv.ptr(dwp), N, H, W, cin, cout, k, stride, pad, BF16, nv.ptr(ws), wsb,
                                       S()), "conv_wgrad_math")
    got = dwp.permute(1, 2, 0).reshape(cout, cin, k, k).cpu().numpy()
    e = rel_err(got, truth)
    assert e <= max(4 * yard, 2e-6), (e, yard)
    if k == 3 and stride == 1 and N == 1:
        dwp2 = torch.full((9, cout, cin), float("nan"), device=DEV)
        wsb = int(lib.frcnn_conv3x3_wgrad_workspace_bytes(H, W, cin, cout))
        ws = torch.empty((wsb // 4 + 1,), device=DEV)
        nv.check(lib.frcnn_conv3x3_wgrad_math(nv.ptr(x_n), nv.ptr(dz_n), nv.ptr(dwp2), H, W, cin, cout, BF16, nv.ptr(ws), wsb, S()),
                 "conv3x3_wgrad_math")
        assert torch.equal(dwp, dwp2)


def _vgg_case(sd_cpu, h, w, seed):
    img = synthetic.image(seed, h, w).unsqueeze(0)
    gts = synthetic.ground_truth(seed, h, w)
    boxes = [Box(class_index=c, class_name="x", corners=k) for c, k in gts]
    am, vm = O.generate_anchor_maps((3, h, w), (512, h // 16, w // 16), 16)
    rmap, obj, bg = O.generate_rpn_map(am, vm, np.stack([k for _, k in gts]))
    return img, gts, boxes, am, vm, rmap, obj, bg


def test_bf16_backward_chain_elementwise_on_injected_oracle_activations(sd_cpu):
    """The whole backward chain in the bf16 gradient arithmetic on PRESCRIBED forward activations (the mechanism of
    test_train_gpu.test_backward_chain_elementwise_on_injected_oracle_activations): kernel and oracle round identical float32
    values to bfloat16, so every gradient tensor must agree elementwise to float32 summation-order accuracy -- while differing
    from the float32 step's gradients by the bf16 rounding, which is what shows the mode is in effect."""
    from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
    from fasterrcnn_amd.models.vgg16 import VGG16Backbone
    h, w, seed = 352, 480, 4
    img, gts, boxes, am, vm, rmap, obj, bg = _vgg_case(sd_cpu, h, w, seed)
    runs = {}
    for gm in ("bf16", "f32"):
        random.seed(5); torch.manual_seed(5)
        od = {}
        runs[gm] = TO.train_step(sd_cpu, img, am, vm, torch.from_numpy(rmap).unsqueeze(0), obj, bg, np.stack([k for _, k in gts]),
                                 np.array([c for c, _ in gts]), 21, 1e-6, 0.9, 5e-4, detail=od, grad_math=gm) + (od,)
    o_losses, o_grads, _, _, od = runs["bf16"]
    assert runs["f32"][0] == o_losses, "the forward pass (and the losses) do not depend on grad_math"
    hwc = lambda x: x.detach()[0].permute(1, 2, 0).contiguous().cuda()
    names = [n for n, _ in TO.VGG_LAYERS]
    inject = {"conv%d" % i: hwc(od[names[i]]) for i in range(4, 13)}
    inject["conv4_in"] = hwc(F.max_pool2d(od[names[3]], 2, 2))
    p = sd_cpu
    fm_o = od[names[12]].detach()
    inject["rpn_trunk"] = hwc(torch.relu(F.conv2d(fm_o, p["_stage2_region_proposal_network._rpn_conv1.weight"],
                                                  p["_stage2_region_proposal_network._rpn_conv1.bias"], padding=1)))
    pooled_o = TO.roi_pool_autograd(fm_o, od["sampled"][0]).detach()
    n_s = pooled_o.shape[0]
    inject["roi_out"] = pooled_o.permute(0, 2, 3, 1).contiguous().reshape(n_s, 49 * 512).cuda()
    pv = "_stage3_detector_network._pool_to_feature_vector."
    h1_o = torch.relu(F.linear(pooled_o.reshape(n_s, -1), p[pv + "_fc1.weight"], p[pv + "_fc1.bias"]))
    h2_o = torch.relu(F.linear(h1_o, p[pv + "_fc2.weight"], p[pv + "_fc2.bias"]))
    inject["fc1"], inject["fc2"] = h1_o.cuda().contiguous(), h2_o.cuda().contiguous()
    model = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
    model.load_state_dict(sd_cpu, strict=True)
    model = model.cuda()
    assert model.grad_math == "f32"
    with pytest.raises(ValueError):
        model.grad_math = "fp8"
    model.grad_math = "bf16"
    opt = T.create_optimizer(model, learning_rate=1e-6)
    random.seed(5); torch.manual_seed(5)
    detail = {"inject": inject}
    loss = T.train_step(model, opt, img.cuda(), am, vm, torch.from_numpy(rmap).unsqueeze(0), [obj], [bg], [boxes], detail=detail)
    assert np.array_equal(detail["sample_idx"].numpy(), od["proposal_sample_indices"])
    got = np.array([loss.rpn_class, loss.rpn_regression, loss.detector_class, loss.detector_regression])
    want = np.array([o_losses[k] for k in ("rpn_class", "rpn_regression", "detector_class", "detector_regression")])
    assert np.all(np.abs(got - want) <= 1e-5 * np.abs(want) + 1e-7), (got, want)
    grads = canonical_grads(detail["grads"])
    worst, least_shift = (0.0, ""), (1.0, "")
    for k, g_ref in o_grads.items():
        gmax = float(g_ref.abs().max())
        e = float((grads[k].cpu() - g_ref).abs().max()) / gmax
        shift = float((runs["f32"][1][k] - g_ref).abs().max()) / gmax          # what the bf16 rounding itself moves
        worst = max(worst, (e, k))
        least_shift = min(least_shift, (shift, k))
        # float32 accumulation-order accuracy (the f32 step's bound on the same test is 1e-5) plus the few operand values that
        # sit on a bfloat16 rounding boundary and round the other way because the incoming gradient differs in its last float32
        # bits (each such flip moves one product by 2^-8): measured 5.1e-5 on the deepest tensor (conv3_1), 45x below the shift
        # that the rounding itself causes
        assert e <= 1e-4, (k, e)
        assert shift >= 5 * e, (k, shift, e)
    print("bf16 injected-activation backward: worst elementwise gradient error / max|g| = %.2e (%s); the bf16 rounding moves the "
          "gradients by >= %.2e (%s)" % (worst[0], worst[1], least_shift[0], least_shift[1]))


def test_bf16_train_step_full_size_is_deterministic_and_learns(sd_cpu):
    """BASELINE configs[1]'s image size in the bf16 gradient arithmetic: two runs from the same seeds are bit-identical, the loss
    falls as in the float32 step (the step times are printed, not asserted: bench.py times them)."""
    import time
    from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
    from fasterrcnn_amd.models.vgg16 import VGG16Backbone
    h, w, seed = 600, 1000, 3
    img, gts, boxes, am, vm, rmap, obj, bg = _vgg_case(sd_cpu, h, w, seed)

    def run(gm, steps=10):
        model = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
        model.load_state_dict(sd_cpu, strict=True)
        model = model.cuda()
        model.grad_math = gm
        opt = T.create_optimizer(model, learning_rate=1e-6)      # the synthetic weights' calibrated step size (bench.py, tools/train_bench.py)
        random.seed(11); torch.manual_seed(11)
        losses = []
        x = img.cuda()
        for i in range(steps):
            if i == 2:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            losses.append(model.train_step(opt, x, am, vm, torch.from_numpy(rmap).unsqueeze(0), [obj], [bg], [boxes]).total)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / (steps - 2) * 1e3
        return losses, {k: v.clone() for k, v in model.state_dict().items()}, ms
    l1, sd1, ms_bf16 = run("bf16")
    l2, sd2, _ = run("bf16")
    lf, _, ms_f32 = run("f32")
    assert l1 == l2 and all(torch.equal(sd1[k], sd2[k]) for k in sd1), "deterministic"
    assert l1[0] == pytest.approx(lf[0], rel=1e-6), "step 0's forward is the float32 forward"
    assert all(np.isfinite(l1)) and l1[-1] < 0.8 * l1[0], l1
    assert abs(l1[-1] - lf[-1]) <= 0.1 * lf[-1], (l1, lf)             # the same trajectory up to the gradient noise
    print("600x1000 VGG-16 train step: grad_math bf16 %.2f ms, f32 %.2f ms; total loss %.4f -> %.4f (f32: %.4f)"
          % (ms_bf16, ms_f32, l1[0], l1[-1], lf[-1]))


def test_bf16_resnet50_train_step_matches_the_bf16_oracle():
    """ResNet-50 (frozen BatchNorm folded into each convolution: training.py _TrainConv), one step at 352x480 against the oracle's
    bf16 restatement run here on the same seeds: identical selections and losses; gradients under the float32 step's criteria
    (tests/test_train_gpu.py: a tight median, looser L2 / norm bounds because two float32 forwards flip a few ReLU decisions) --
    and closer to the bf16 oracle than to the float32 oracle, which is what shows the arithmetic is the restated one."""
    from fasterrcnn_amd.models import resnet
    from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
    h, w, seed = 352, 480, 4
    sd0 = synthetic.resnet_state_dict(1234, "ResNet50")
    img = synthetic.image_rgb(seed, h, w).unsqueeze(0)
    gts = synthetic.ground_truth(seed, h, w)
    boxes = [Box(class_index=c, class_name="x", corners=k) for c, k in gts]
    am, vm = O.generate_anchor_maps((3, h, w), (1024, -(-h // 16), -(-w // 16)), 16)
    rmap, obj, bg = O.generate_rpn_map(am, vm, np.stack([k for _, k in gts]))
    ref = {}
    for gm in ("bf16", "f32"):
        random.seed(5); torch.manual_seed(5)
        od = {}
        ref[gm] = TO.train_step(sd0, img, am, vm, torch.from_numpy(rmap).unsqueeze(0), obj, bg, np.stack([k for _, k in gts]),
                                np.array([c for c, _ in gts]), 21, 1e-6, 0.9, 5e-4, detail=od, grad_math=gm) + (od,)
    o_losses, o_grads, _, _, od = ref["bf16"]
    model = FasterRCNNModel(num_classes=21, backbone=resnet.ResNetBackbone(resnet.Architecture.ResNet50))
    model.load_state_dict(sd0, strict=True)
    model = model.cuda()
    model.grad_math = "bf16"
    opt = T.create_optimizer(model, learning_rate=1e-6)
    random.seed(5); torch.manual_seed(5)
    detail = {}
    loss = T.train_step(model, opt, img.cuda(), am, vm, torch.from_numpy(rmap).unsqueeze(0), [obj], [bg], [boxes], detail=detail)
    assert np.array_equal(detail["sample_idx"].numpy(), od["proposal_sample_indices"])
    got = np.array([loss.rpn_class, loss.rpn_regression, loss.detector_class, loss.detector_regression])
    want = np.array([o_losses[k] for k in ("rpn_class", "rpn_regression", "detector_class", "detector_regression")])
    assert np.all(np.abs(got - want) <= 2e-5 * np.abs(want) + 1e-7), (got, want)
    grads = canonical_grads_resnet(detail["grads"])
    assert sorted(grads) == sorted(o_grads)
    gscale = max(float(g.norm()) for g in o_grads.values())
    worst = {"median": (0.0, ""), "L2": (0.0, ""), "norm": (0.0, "")}
    closer = 0
    for k, g_ref in o_grads.items():
        g = grads[k].cpu().double().reshape(-1)
        r = g_ref.double().reshape(-1)
        r32 = ref["f32"][1][k].double().reshape(-1)
        ref_max = max(float(r.abs().max()), 1e-7 * gscale)
        med = float((g - r).abs().median()) / ref_max
        l2 = float((g - r).norm()) / max(float(r.norm()), 1e-7 * gscale)
        nrm = abs(float(g.norm()) - float(r.norm())) / max(float(r.norm()), 1e-7 * gscale)
        for name, val in (("median", med), ("L2", l2), ("norm", nrm)):
            worst[name] = max(worst[name], (val, k.split(".")[-3] + "." + k.split(".")[-2]))
        assert med <= 1e-4 and l2 <= 1e-2 and nrm <= 5e-3, (k, med, l2, nrm)
        closer += float((g - r).norm()) < float((g - r32).norm())
    print("bf16 ResNet-50 step vs the bf16 oracle: worst gradient errors %s; closer to the bf16 than to the f32 oracle on %d of %d tensors"
          % (", ".join("%s %.2e (%s)" % (n, v[0], v[1]) for n, v in worst.items()), closer, len(o_grads)))
    assert closer >= 0.9 * len(o_grads)


def test_bf16_resnet101_roialign_train_step_full_size():
    """BASELINE configs[4] on one GPU: ResNet-101, 600x1000, RoIAlign, bf16 gradient GEMMs: deterministic, finite, the loss falls
    along the float32 trajectory (the step times are printed, not asserted: bench.py times them)."""
    import time
    from fasterrcnn_amd.models import resnet
    from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
    h, w, seed = 600, 1000, 2
    sd0 = synthetic.resnet_state_dict(1234, "ResNet101")
    img = synthetic.image_rgb(seed, h, w).unsqueeze(0).cuda()
    gts = synthetic.ground_truth(seed, h, w)
    boxes = [Box(class_index=c, class_name="x", corners=k) for c, k in gts]
    am, vm = O.generate_anchor_maps((3, h, w), (1024, -(-h // 16), -(-w // 16)), 16)
    rmap, obj, bg = O.generate_rpn_map(am, vm, np.stack([k for _, k in gts]))
    rmap_t = torch.from_numpy(rmap).unsqueeze(0).cuda()

    def run(gm, steps=6):
        model = FasterRCNNModel(num_classes=21, backbone=resnet.ResNetBackbone(resnet.Architecture.ResNet101), roi_pooling="align")
        model.load_state_dict(sd0, strict=True)
        model = model.cuda()
        model.grad_math = gm
        opt = T.create_optimizer(model, learning_rate=3e-6)
        random.seed(5); torch.manual_seed(5)
        losses = []
        for i in range(steps):
            if i == 2:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            losses.append(model.train_step(opt, img, am, vm, rmap_t, [obj], [bg], [boxes]).total)
        torch.cuda.synchronize()
        return losses, {k: v.clone() for k, v in model.state_dict().items()}, (time.perf_counter() - t0) / (steps - 2) * 1e3
    l1, s1, ms_bf16 = run("bf16")
    l2, s2, _ = run("bf16")
    lf, _, ms_f32 = run("f32")
    assert l1 == l2 and all(torch.equal(s1[k], s2[k]) for k in s1), "deterministic"
    assert all(np.isfinite(l1)) and l1[-1] < l1[0]
    assert l1[0] == pytest.approx(lf[0], rel=1e-6) and abs(l1[-1] - lf[-1]) <= 0.05 * abs(lf[-1]), (l1, lf)
    print("ResNet-101 600x1000 RoIAlign train step: grad_math bf16 %.2f ms, f32 %.2f ms; total loss %s (f32: %s)"
          % (ms_bf16, ms_f32, ["%.4f" % x for x in l1], ["%.4f" % x for x in lf]))
