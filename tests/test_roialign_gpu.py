"""
RoIAlign (csrc/roialign.hip) -- the pooling BASELINE.json's north_star / configs[4] name; the reference itself uses RoIPool
(models/detector.py:16,27), so this is an option beyond it: FasterRCNNModel(..., roi_pooling="align").

The oracle (oracle/frcnn_oracle.py: roi_align / roi_align_backward) restates torchvision.ops.roi_align's published algorithm
(torchvision is a third-party dependency that is not in /root/reference: parity unpinned, as for nms and RoIPool).
Tolerances: forward float32 in the same operation order -> <= 2e-7 of max|y| (bit-exact on most inputs); backward against the
float64 accumulation of the same sampling plan <= 2e-6 of max|d|, run-to-run identical (gathered per cell, no atomics);
train step with RoIAlign against the oracle's autograd step: losses <= 2e-5 relative, gradients as for the RoIPool fixtures.
"""
import random

import numpy as np
import pytest
import torch

from fasterrcnn_amd import _native as nv
from fasterrcnn_amd import synthetic
from fasterrcnn_amd import training as T
from fasterrcnn_amd.datasets.training_sample import Box
from oracle import frcnn_oracle as O
from oracle import train_oracle as TO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_rois(rng, n, h, w):
    """(n, 4) float32 (y1, x1, y2, x2) in image pixels: ordinary boxes plus the edge cases (tiny, clipped, larger than the map)."""
    y1 = rng.uniform(0, h * 16 - 40, n); x1 = rng.uniform(0, w * 16 - 40, n)
    rois = np.stack([y1, x1, y1 + rng.uniform(16, h * 12, n), x1 + rng.uniform(16, w * 12, n)], 1)
    rois[:, 2] = np.minimum(rois[:, 2], h * 16); rois[:, 3] = np.minimum(rois[:, 3], w * 16)
    special = np.array([[0, 0, h * 16, w * 16], [5.3, 7.9, 6.1, 9.2], [h * 16 - 3, w * 16 - 3, h * 16, w * 16],
                        [-40, -40, 30, 50], [h * 16 - 10, 10, h * 16 + 60, 90], [100, 100, 100, 100]], dtype=np.float64)
    rois[:len(special)] = special[:n]
    return rois.astype(np.float32)


def run_forward(fm_chw, rois, sampling_ratio, aligned=False, max_rois=None):
    c, h, w = fm_chw.shape
    n = rois.shape[0]
    max_rois = max_rois or n
    x = torch.from_numpy(fm_chw).permute(1, 2, 0).contiguous().to(DEV)
    r = torch.zeros((max_rois, 4), device=DEV)
    r[:n] = torch.from_numpy(rois).to(DEV)
    out = torch.full((max_rois, 7, 7, c), float("nan"), device=DEV)
    cnt = torch.tensor([n], dtype=torch.int32, device=DEV)
    nv.check(nv.lib().frcnn_roi_align(nv.ptr(x), h, w, c, nv.ptr(r), nv.ptr(cnt), max_rois, 7, 1.0 / 16.0, sampling_ratio,
                                      1 if aligned else 0, nv.ptr(out), nv.stream_ptr()), "frcnn_roi_align")
    torch.cuda.synchronize()
    return out.permute(0, 3, 1, 2).cpu().numpy()


@pytest.mark.parametrize("c,h,w,n,sr,aligned", [(64, 37, 62, 40, 2, False), (16, 12, 20, 12, 1, False), (32, 9, 7, 9, -1, False),
                                                (8, 38, 63, 10, 2, True), (1024, 5, 6, 7, 2, False)])
def test_forward_matches_oracle(c, h, w, n, sr, aligned):
    rng = np.random.RandomState(c + h + n)
    fm = rng.randn(c, h, w).astype(np.float32)
    rois = make_rois(rng, n, h, w)
    got = run_forward(fm, rois, sr, aligned, max_rois=n + 3)
    xyxy = np.zeros((n, 5), np.float32)
    xyxy[:, 1:] = rois[:, [1, 0, 3, 2]]
    want = O.roi_align(fm[None], xyxy, 7, 1.0 / 16.0, sr, aligned)
    assert (got[n:] == 0).all()                                        # rows >= n_rois are zeros
    err = np.abs(got[:n] - want).max() / np.abs(want).max()
    print("roi_align c=%d %dx%d n=%d sr=%d aligned=%d: max err / max|y| %.3g, bit-identical %.1f%%" % (
        c, h, w, n, sr, aligned, err, 100.0 * (got[:n] == want).mean()))
    assert err <= 2e-7
    # a constant map pools to the constant wherever the bin lies inside the map
    ones = run_forward(np.ones((c, h, w), np.float32), rois[:1], sr, aligned)
    assert np.abs(ones - 1.0).max() <= 1e-6


@pytest.mark.parametrize("c,h,w,n,sr", [(64, 37, 62, 128, 2), (16, 12, 20, 12, 1), (1024, 9, 7, 20, 2), (8, 20, 30, 16, -1)])
def test_backward_matches_oracle_and_is_deterministic(c, h, w, n, sr):
    rng = np.random.RandomState(c * 3 + n)
    rois = make_rois(rng, n, h, w)
    g = rng.randn(n, c, 7, 7).astype(np.float32)
    xyxy = np.zeros((n, 5), np.float32)
    xyxy[:, 1:] = rois[:, [1, 0, 3, 2]]
    want = O.roi_align_backward(g, (1, c, h, w), xyxy, 7, 1.0 / 16.0, sr, False)[0]
    dg = torch.from_numpy(g).permute(0, 2, 3, 1).contiguous().to(DEV)
    dr = torch.from_numpy(rois).to(DEV)
    outs = []
    for _ in range(2):
        d = torch.full((h, w, c), float("nan"), device=DEV)
        nv.check(nv.lib().frcnn_roi_align_backward(nv.ptr(dr), n, h, w, c, 7, 1.0 / 16.0, sr, 0, nv.ptr(dg), nv.ptr(d), 0,
                                                   nv.stream_ptr()), "frcnn_roi_align_backward")
        outs.append(d.permute(2, 0, 1).cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    err = np.abs(outs[0] - want).max() / np.abs(want).max()
    print("roi_align backward c=%d %dx%d n=%d sr=%d: max err / max|d| %.3g" % (c, h, w, n, sr, err))
    assert err <= 2e-6
    # adjoint identity on the device results: <g, A x> == <A^T g, x>
    x = rng.randn(c, h, w).astype(np.float32)
    y = run_forward(x, rois, sr)
    lhs, rhs = float((g.astype(np.float64) * y).sum()), float((outs[0].astype(np.float64) * x).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0)
    # accumulate = 1 adds onto the existing gradient
    d2 = torch.ones((h, w, c), device=DEV)
    nv.check(nv.lib().frcnn_roi_align_backward(nv.ptr(dr), n, h, w, c, 7, 1.0 / 16.0, sr, 0, nv.ptr(dg), nv.ptr(d2), 1,
                                               nv.stream_ptr()), "frcnn_roi_align_backward")
    assert np.abs(d2.permute(2, 0, 1).cpu().numpy() - (outs[0] + 1.0)).max() <= 1e-5 * max(1.0, np.abs(outs[0]).max())


def make_model(sd_cpu, **kw):
    from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
    from fasterrcnn_amd.models.vgg16 import VGG16Backbone
    m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0), **kw)
    m.load_state_dict(sd_cpu, strict=True)
    return m.cuda()


def test_model_with_roi_align_matches_oracle(sd_cpu):
    model = make_model(sd_cpu, roi_pooling="align").eval()
    assert model._stage3_detector_network.pooling == "align" and sorted(model.state_dict().keys()) == sorted(sd_cpu.keys())
    img = synthetic.image(3, 224, 320).unsqueeze(0)
    detail = {}
    o_props, o_classes, o_deltas = O.forward(sd_cpu, img, detail=detail, roi_pooling="align")
    props, classes, deltas = model(image_data=img.cuda())
    assert props.shape[0] == o_props.shape[0]
    d = np.abs(props.cpu().numpy()[:, None, :] - o_props.numpy()[None, :, :]).max(axis=2)
    j = d.argmin(axis=0)
    ok = d[j, np.arange(len(j))] <= 1e-3
    assert ok.mean() >= 0.97
    assert np.abs(classes.cpu().numpy()[j[ok]] - o_classes.numpy()[ok]).max() <= 1e-4
    assert np.abs(deltas.cpu().numpy()[j[ok]] - o_deltas.numpy()[ok]).max() <= 1e-3
    # stage level on the oracle's own inputs: pooled features within float32 rounding
    det = model._stage3_detector_network
    pooled = det.roi_pool(detail["feature_map"].cuda(), o_props.cuda()).cpu()
    assert float((pooled - detail["pooled"]).abs().max()) <= 2e-7 * float(detail["pooled"].abs().max())
    # RoIAlign differs from RoIPool (the option is really in effect) and predict() works
    pooled_max = make_model(sd_cpu)._stage3_detector_network.roi_pool(detail["feature_map"].cuda(), o_props.cuda()).cpu()
    assert float((pooled - pooled_max).abs().max()) > 1e-2
    out = model.predict(image_data=img.cuda(), score_threshold=0.05)
    ref = O.detections(o_props.numpy(), o_classes.numpy(), o_deltas.numpy(), 224, 320, 0.05)
    assert abs(sum(len(v) for v in out.values()) - sum(len(v) for v in ref.values())) <= 3
    with pytest.raises(ValueError):
        make_model(sd_cpu, roi_pooling="max")


def test_train_step_with_roi_align_matches_oracle_autograd_step(sd_cpu):
    """BASELINE configs[4]'s "RoIAlign grad": forward + backward + SGD with RoIAlign in the detector stage, against the oracle's
    step (torch-CPU autograd around the restated roi_align / roi_align_backward), same seeds -> same samples."""
    h, w, seed = 352, 480, 4
    img = synthetic.image(seed, h, w).unsqueeze(0)
    gts = synthetic.ground_truth(seed, h, w)
    boxes = [Box(class_index=c, class_name="x", corners=k) for c, k in gts]
    am, vm = O.generate_anchor_maps((3, h, w), (512, h // 16, w // 16), 16)
    rmap, obj, bg = O.generate_rpn_map(am, vm, np.stack([k for _, k in gts]))
    lr, mom, wd = 1e-6, 0.9, 5e-4
    random.seed(11); torch.manual_seed(11)
    odetail = {}
    o_losses, o_grads, o_sd, _ = TO.train_step(sd_cpu, img, am, vm, torch.from_numpy(rmap).unsqueeze(0), obj, bg,
                                               np.stack([k for _, k in gts]), np.array([c for c, _ in gts]), 21, lr, mom, wd,
                                               detail=odetail, roi_pooling="align")
    model = make_model(sd_cpu, roi_pooling="align")
    opt = T.create_optimizer(model, learning_rate=lr, momentum=mom, weight_decay=wd)
    random.seed(11); torch.manual_seed(11)
    detail = {}
    loss = T.train_step(model, opt, img.cuda(), am, vm, torch.from_numpy(rmap).unsqueeze(0), [obj], [bg], [boxes], detail=detail)
    got = np.array([loss.rpn_class, loss.rpn_regression, loss.detector_class, loss.detector_regression, loss.total])
    want = np.array([float(o_losses[k]) for k in ("rpn_class", "rpn_regression", "detector_class", "detector_regression", "total")])
    print("RoIAlign train step losses:", got, "oracle:", want)
    assert np.all(np.abs(got - want) <= 2e-5 * np.abs(want) + 1e-7)
    sp, osp = detail["sampled_props"].cpu().numpy(), odetail["sampled"][0].numpy()
    assert sp.shape == osp.shape and np.abs(sp - osp).max() <= 1e-3
    new_sd = model.state_dict()
    # updated weights: the no-flip tensors (heads, fc2) elementwise; every trainable tensor by update norm
    for key in ("_stage3_detector_network._classifier.weight", "_stage3_detector_network._regressor.weight",
                "_stage3_detector_network._pool_to_feature_vector._fc2.weight"):
        du = (new_sd[key].cpu() - sd_cpu[key]).double()
        dw = (o_sd[key] - sd_cpu[key]).double()
        # (the update is ~5e-6 on weights of ~1e-2: a float32 ulp of the weight itself is 1e-9 of it, i.e. ~2e-4 of the update)
        assert float((du - dw).abs().max()) <= 2e-5 * float(dw.abs().max()) + 2.0 ** -23 * float(sd_cpu[key].abs().max()), key
    for key in TO.trainable_weight_keys(sd_cpu):
        du = (new_sd[key].cpu() - sd_cpu[key]).double()
        dw = (o_sd[key] - sd_cpu[key]).double()
        rel = float((du - dw).norm()) / max(float(dw.norm()), 1e-30)
        assert rel <= 1e-2, (key, rel)
