"""Host-side logic of the product package on CPU: model surface, error behaviour, mAP mirror."""
import os

import numpy as np
import pytest
import torch
import torch as t

from fasterrcnn_amd import statistics, synthetic, training
from fasterrcnn_amd.datasets.training_sample import Box
from fasterrcnn_amd.models import math_utils
from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
from fasterrcnn_amd.models.vgg16 import VGG16Backbone
from oracle import frcnn_oracle as O


@pytest.fixture(scope="module")
def cpu_model():
    return FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))


def test_state_dict_keys_match_reference(cpu_model, sd_cpu):
    keys = list(cpu_model.state_dict().keys())
    assert len(keys) == 40
    assert sorted(keys) == sorted(sd_cpu.keys())                      # names captured from the reference
    assert sum(p.numel() for p in cpu_model.parameters()) == 137057234
    cpu_model.load_state_dict(sd_cpu, strict=True)
    assert cpu_model.state_dict()["_stage3_detector_network._pool_to_feature_vector._fc1.weight"].shape == (4096, 25088)


def test_backbone_contract(cpu_model):
    b = cpu_model.backbone
    assert (b.feature_map_channels, b.feature_pixels, b.feature_vector_size) == (512, 16, 4096)
    assert b.compute_feature_map_shape((3, 600, 1000)) == (512, 37, 62)
    assert b.image_preprocessing_params.means == [103.939, 116.779, 123.680]
    assert b.image_preprocessing_params.channel_order.value == "BGR"


def test_no_cpu_fallback(cpu_model):
    img = torch.zeros((1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="MI355X"):
        cpu_model.predict(img, score_threshold=0.05)
    with pytest.raises(AssertionError, match="Batch size must be 1"):
        cpu_model.predict(torch.zeros((2, 3, 64, 64)), score_threshold=0.05)
    # training has no CPU path either: the packed masters are built on the device
    with pytest.raises(RuntimeError, match="MI355X"):
        cpu_model.train_step(training.create_optimizer(cpu_model), img, None, None, torch.zeros((1, 4, 4, 9, 6)), [np.zeros((1, 3))], [np.zeros((1, 3))], [[]])


def test_product_does_not_import_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "fasterrcnn_amd")):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, os.path.join(dirpath, f)


def test_iou_matches_oracle_pairwise():
    rng = np.random.RandomState(0)
    a = np.sort(rng.rand(7, 2, 2) * 100, axis=1).reshape(7, 4)[:, [0, 1, 2, 3]]
    b = np.sort(rng.rand(5, 2, 2) * 100, axis=1).reshape(5, 4)
    m = math_utils.intersection_over_union(a, b)
    for i in range(7):
        for j in range(5):
            assert m[i, j] == O.iou_pair(a[i], b[j])


def _stream(golden_dir):
    g = np.load(os.path.join(golden_dir, "small_ops.npz"))
    gts, preds = g["map_stream_gt"], g["map_stream_pred"]
    for i in range(int(gts[:, 0].max()) + 1):
        gt = [(int(r[1]), r[2:6].astype(np.float32)) for r in gts[gts[:, 0] == i]]
        p = {c: preds[(preds[:, 0] == i) & (preds[:, 1] == c)][:, 2:7] for c in range(1, 21)}
        yield gt, p
    return


def test_map_mirror_equals_reference_values(golden_dir):
    g = np.load(os.path.join(golden_dir, "small_ops.npz"))
    calc = statistics.PrecisionRecallCurveCalculator()
    for gt, p in _stream(golden_dir):
        calc.add_image_results(p, [Box(c, "x", k) for c, k in gt])
    assert float(calc.compute_mean_average_precision()) == float(g["map_stream_value"])
    # SURVEY 8(a14) known answer
    calc = statistics.PrecisionRecallCurveCalculator()
    preds = {7: np.array([[100, 200, 400, 700, .9], [110, 210, 390, 690, .8], [0, 0, 50, 50, .7]]),
             15: np.array([[50, 50, 300, 180, .6]])}
    for c in range(1, 21):
        preds.setdefault(c, np.zeros((0, 5)))
    calc.add_image_results(preds, [Box(7, "car", np.array([100, 200, 400, 700], np.float32)),
                                   Box(15, "person", np.array([50, 50, 300, 180], np.float32))])
    assert [tp for _, tp in calc._unsorted_predictions_by_class_index[7]] == [True, False, False]
    assert float(calc.compute_mean_average_precision()) == 1.0


def test_map_state_roundtrip_is_exact(golden_dir):
    whole = statistics.PrecisionRecallCurveCalculator()
    parts = [statistics.PrecisionRecallCurveCalculator() for _ in range(3)]
    for i, (gt, p) in enumerate(_stream(golden_dir)):
        boxes = [Box(c, "x", k) for c, k in gt]
        whole.add_image_results(p, boxes)
        parts[i % 3].add_image_results(p, boxes)
    merged = statistics.PrecisionRecallCurveCalculator()
    for part in parts:
        merged.merge_state(part.state())
    # image-interleaved sharding changes the insertion order of equal-score records only;
    # scores here are continuous, so mAP must agree to the last bit
    assert float(merged.compute_mean_average_precision()) == float(whole.compute_mean_average_precision())


def test_synthetic_workload_is_reproducible():
    a, b = synthetic.image(4), synthetic.image(4)
    assert a.shape == (3, 600, 1000) and torch.equal(a, b)
    assert not torch.equal(a, synthetic.image(5))
    gt = synthetic.ground_truth(4)
    assert 1 <= len(gt) <= 5 and all(1 <= c <= 20 and k.dtype == np.float32 for c, k in gt)


def test_resnet_state_dict_keys_match_reference():
    from fasterrcnn_amd.models import resnet
    m = FasterRCNNModel(num_classes=21, backbone=resnet.ResNetBackbone(resnet.Architecture.ResNet50))
    sd = synthetic.resnet_state_dict(1234, "ResNet50")            # key names captured from the reference
    assert len(m.state_dict()) == 328 and sorted(m.state_dict().keys()) == sorted(sd.keys())
    assert sum(p.numel() for p in m.parameters()) == 33199314
    m.load_state_dict(sd, strict=True)
    b = m.backbone
    assert (b.feature_map_channels, b.feature_pixels, b.feature_vector_size) == (1024, 16, 2048)
    assert b.compute_feature_map_shape((3, 600, 1000)) == (1024, 38, 63)
    assert b.image_preprocessing_params.channel_order.value == "RGB"
    with pytest.raises(ValueError):
        resnet.ResNetBackbone("ResNet18")
    m101 = resnet.ResNetBackbone(resnet.Architecture.ResNet101)
    assert len(m101.feature_extractor._feature_extractor[6]) == 23


def test_checkpoint_formats(tmp_path, cpu_model, sd_cpu):
    from fasterrcnn_amd import state
    # 1. the reference's own checkpoint format
    path = str(tmp_path / "fasterrcnn.pth")
    torch.save({"epoch": 3, "model_state_dict": sd_cpu}, path)
    fresh = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
    assert state.load(fresh, path) == []
    for k, v in fresh.state_dict().items():
        assert torch.equal(v, sd_cpu[k]), k
    state.save(fresh, path, epoch=4)
    assert torch.load(path)["epoch"] == 4
    with pytest.raises(KeyError):
        bad = str(tmp_path / "bad.pth")
        torch.save({"something": 1}, bad)
        state.load(fresh, bad)
    # 2. Caffe / torchvision-style VGG-16 file: conv blocks + fc1/fc2 land on the LIVE keys
    names = {"features.%d" % n: k for n, k in zip((0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28),
             ["_stage1_feature_extractor._block%d_conv%d" % (b, c) for b, n in ((1, 2), (2, 2), (3, 3), (4, 3), (5, 3)) for c in range(1, n + 1)])}
    caffe = {}
    for src, dst in names.items():
        caffe[src + ".weight"], caffe[src + ".bias"] = sd_cpu[dst + ".weight"] * 2, sd_cpu[dst + ".bias"] + 1
    p = "_stage3_detector_network._pool_to_feature_vector."
    caffe["classifier.0.weight"], caffe["classifier.0.bias"] = sd_cpu[p + "_fc1.weight"] * 2, sd_cpu[p + "_fc1.bias"] + 1
    caffe["classifier.3.weight"], caffe["classifier.3.bias"] = sd_cpu[p + "_fc2.weight"] * 2, sd_cpu[p + "_fc2.bias"] + 1
    cpath = str(tmp_path / "vgg16_caffe.pth")
    torch.save(caffe, cpath)
    left = state.load(fresh, cpath)
    assert sorted(left) == sorted(k for k in sd_cpu if "_rpn_" in k or "_classifier" in k or "_regressor" in k)
    assert torch.equal(fresh.state_dict()[p + "_fc1.weight"], sd_cpu[p + "_fc1.weight"] * 2)
    assert torch.equal(fresh.state_dict()["_stage1_feature_extractor._block3_conv2.bias"], sd_cpu["_stage1_feature_extractor._block3_conv2.bias"] + 1)
    tracker = state.BestWeightsTracker(str(tmp_path / "best.pth"))
    tracker.on_epoch_end(fresh, 1, 10.0)
    tracker.on_epoch_end(fresh, 2, 5.0)
    tracker.save_best_weights(fresh)
    assert torch.load(str(tmp_path / "best.pth"))["epoch"] == 1


def test_proposal_sampler_draws_like_the_reference():
    """training._sample_proposal_indices consumes torch's CPU generator exactly as faster_rcnn.py:512-561 does."""
    from fasterrcnn_amd import training
    from oracle import train_oracle as TO
    rng = np.random.RandomState(3)
    cls = t.from_numpy((rng.rand(900) < 0.08).astype(np.int64) * rng.randint(1, 21, 900))
    onehot = t.nn.functional.one_hot(cls, 21).float()
    props = t.arange(900, dtype=t.float32).reshape(-1, 1).repeat(1, 4)
    deltas = t.zeros((900, 2, 80))
    for max_props in (128, 2000, 0):
        t.manual_seed(77)
        idx = training._sample_proposal_indices(cls, max_props, 0.25)
        t.manual_seed(77)
        rp, rc, _ = TO.sample_proposals(props, onehot, deltas, max_props, 0.25)
        assert t.equal(props[idx], rp) and t.equal(onehot[idx], rc)
    # no positives -> empty batch (faster_rcnn.py:552-553)
    assert training._sample_proposal_indices(t.zeros((50,), dtype=t.int64), 128, 0.25).shape[0] == 0


def test_optimizer_hyper_parameters_from_torch_sgd():
    from fasterrcnn_amd import training
    p = t.nn.Parameter(t.zeros(3))
    q = t.nn.Parameter(t.zeros(3))
    opt = t.optim.SGD([{"params": [p], "weight_decay": 5e-4}, {"params": [q], "weight_decay": 5e-4}], lr=1e-3, momentum=0.9)
    assert training.sgd_hyper_parameters(opt) == (1e-3, 0.9, 5e-4)
    assert training.sgd_hyper_parameters(training.create_optimizer(None)) == (1e-3, 0.9, 5e-4)
    bad = t.optim.SGD([{"params": [p], "weight_decay": 0.0}, {"params": [q], "weight_decay": 5e-4}], lr=1e-3, momentum=0.9)
    with pytest.raises(NotImplementedError):
        training.sgd_hyper_parameters(bad)


# ---- bench.py's N > 1 launch path without hardware (VERDICT r2 #7b) ----------------------------------------------------------
def test_bench_multi_gpu_relaunch_command_and_environment(monkeypatch):
    """`python bench.py --gpus 8` without a launcher re-executes itself under torch.distributed.run: the command line must be the
    driver's form (one node, 8 ranks, 127.0.0.1 rendezvous, flags passed through) and the children's environment must carry
    dmabuf IPC + the HIP queue count.  The launcher itself is mocked."""
    import subprocess
    import sys
    import bench
    seen = {}

    def fake_call(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = list(cmd), dict(env)
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "2"])
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "HSA_ENABLE_IPC_MODE_LEGACY"):
        monkeypatch.delenv(k, raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    else:
        raise AssertionError("bench.main() must hand over to the launcher")
    cmd, env = seen["cmd"], seen["env"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    script = cmd.index(bench.os.path.abspath(bench.__file__))
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "7", "--warmup", "2"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["GPU_MAX_HW_QUEUES"] == "16"
    # a rank of that job reads its identity (and its device = LOCAL_RANK) from the launcher's variables
    assert bench.rank_environment({"RANK": "5", "LOCAL_RANK": "5", "WORLD_SIZE": "8"}) == (5, 5, 8)
    assert bench.rank_environment({}) == (0, 0, 1)
    # an explicit user setting wins over the defaults
    _, env2 = bench.launcher_command(2, [], {"GPU_MAX_HW_QUEUES": "4", "HSA_ENABLE_IPC_MODE_LEGACY": "1"})
    assert env2["GPU_MAX_HW_QUEUES"] == "4" and env2["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"


def test_bench_layer_arithmetic_discloses_pipes():
    """Every GEMM-shaped layer of the image appears once with the pipe it runs on; the per-pipe sums equal the documented figures."""
    import bench
    rows = bench.layer_arithmetic("f32_winograd", "f32x6")
    names = [r[0] for r in rows]
    assert names == bench._CONV_NAMES + ["rpn_heads_1x1", "fc1", "fc2", "detector_heads"]
    by = {r[0]: r for r in rows}
    assert by["fc1"][2] == "bf16" and by["fc1"][3] == 6.0 * by["fc1"][4] and by["fc1"][1] == "gemm_x6t_kernel"
    assert all(by[n][2] == "f32" and by[n][1] == "wino_fused_kernel" for n in bench._CONV_NAMES)
    pf = bench.pipe_flops_per_image("f32_winograd", "f32x6")
    assert abs(pf["f32"] - (1.683e11 + 2.0 * 512 * 45 * 37 * 62 + 300 * 2.0 * 4096 * 101)) / pf["f32"] < 1e-3
    assert abs(pf["bf16"] - 6.0 * 300 * 2.0 * (25088 * 4096 + 4096 * 4096)) / pf["bf16"] < 1e-12
    assert bench.pipe_flops_per_image("f32", "f32")["bf16"] == 0.0
    # algorithmic total == BASELINE.md section 3 minus conv1_1 (the VALU layer): 4.4922e11 - 2*27*64*600*1000
    alg = sum(r[4] for r in bench.layer_arithmetic("f32", "f32"))
    assert abs(alg - (4.4922e11 - 2.0 * 27 * 64 * 600 * 1000)) / alg < 2e-4


def test_bench_layer_arithmetic_f32x3_tables():
    """The f32x3 layers count three fp16 MFMAs per float32 product, the f32x6 layers six bf16 MFMAs; a name in the x3 table only counts
    when it is in the x6 table too; the default tables of the model are the ones bench.py accounts for."""
    import bench
    from fasterrcnn_amd import _native as nv
    x6, x3 = nv.DEFAULT_X6_LAYERS_VGG16, nv.DEFAULT_X3_LAYERS_VGG16
    x3f = nv.DEFAULT_X3F_LAYERS_VGG16
    assert x3 == x6 and "conv5_1" in x3                     # round 4: the whole x6 table in f32x3 (chosen on the held-out set: DESIGN.md section 4)
    assert x3f == ("conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3") and not set(x3f) & set(x6)
    rows = {r[0]: r for r in bench.layer_arithmetic("f32_winograd", "f32x3", x6, x3=x3, x3f=x3f)}
    for n in x3f:                                            # the one-launch f32x3 layers: three fp16 MFMAs per product, no x6 entry needed
        ci, co, h, w = dict(zip(bench._CONV_NAMES, bench._MFMA_CONVS))[n]
        assert rows[n][1].startswith("wino_x3d_kernel") and rows[n][2] == "f16" and rows[n][3] == 3.0 * bench.winograd_gemm_flops(ci, co, h, w)
    assert [n for n, _ in bench.winograd_layers("f32_winograd", x6, named=True, x3f=x3f)] == []          # no float32 one-launch layer is left
    r0 = {r[0]: r for r in bench.layer_arithmetic("f32_winograd", "f32x3", x6, x3=x3, x3f=x3f[1:])}
    assert r0["conv1_2"][1] == "wino_fused_kernel" and r0["conv1_2"][2] == "f32"
    assert [n for n, _ in bench.x3f_winograd_layers("f32_winograd", x6, x3f)] == list(x3f)
    rows = {r[0]: r for r in bench.layer_arithmetic("f32_winograd", "f32x3", x6, x3=tuple(n for n in x3 if n != "conv5_1"))}
    x3 = tuple(n for n in x3 if n != "conv5_1")
    for n in x6:
        if n == "rpn_trunk":
            continue
        kern, pipe, ex = rows[n][1], rows[n][2], rows[n][3]
        ci, co, h, w = dict(zip(bench._CONV_NAMES, bench._MFMA_CONVS))[n]
        g = bench.winograd_gemm_flops(ci, co, h, w)
        assert (pipe, ex) == (("f16", 3.0 * g) if n in x3 else ("bf16", 6.0 * g)), n
        assert kern.startswith("gemm_x3t_kernel" if n in x3 else "gemm_x6t_kernel")
    assert rows["fc1"][1] == "gemm_x3t_kernel" and rows["fc1"][2] == "f16" and rows["fc1"][3] == 3.0 * rows["fc1"][4]
    # an x3 name outside the x6 table stays on the float32 kernel
    r2 = {r[0]: r for r in bench.layer_arithmetic("f32_winograd", "f32", ("conv4_2",), x3=("conv4_2", "conv5_2"))}
    assert r2["conv4_2"][2] == "f16" and r2["conv5_2"][2] == "f32" and r2["fc1"][2] == "f32"
    pf = bench.pipe_flops_per_image("f32_winograd", "f32x3", x6, x3=x3)
    assert pf["f16"] > 0 and pf["bf16"] > 0 and abs(pf["f32"] + pf["bf16"] + pf["f16"] - bench.executed_mfma_flops_per_image("f32_winograd", "f32x3", x6, x3)) < 1.0
    x3 = nv.DEFAULT_X3_LAYERS_VGG16
    m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
    assert m.winograd_x6_layers == x6 and m.winograd_x3_layers == x3 and m.fc_math_mode == "f32x3" and m.winograd_x3f_layers == x3f
    assert m._x3_mask() & ~m._x6_mask() == 0 and m._x3f_mask() & m._x6_mask() == 0
    alone = nv.DEFAULT_ALONE_X3F_LAYERS_VGG16
    fe_alone = tuple(n for n in alone if n != "rpn_trunk")
    assert m._x3f_mask() == sum(1 << nv.X6_LAYER_BITS[n] for n in x3f) and m._stage1_feature_extractor.x3f_layers == x3f + fe_alone
    # ONE table for every slot since round 5: the f32x3 layers of the x6 table run in the one-launch form in the in-flight slots
    # (inflight_winograd_x3f_layers) AND in slot 0 (alone_winograd_x3f_layers); () for the latter restores round 4's three launches there
    assert m.inflight_winograd_x3f_layers == nv.DEFAULT_INFLIGHT_X3F_LAYERS_VGG16 == x6 and m.alone_winograd_x3f_layers == alone == x6
    assert m.layer_tables(0) == ((), (), x3f + x6) == m.layer_tables(1)
    bits = lambda names: sum(1 << nv.X6_LAYER_BITS[n] for n in names)
    assert m._slot_masks(0) == (0, 0, bits(x3f + x6)) == m._slot_masks(2)
    rpn = m._stage2_region_proposal_network
    assert rpn.x3f_trunk and not rpn.x3_trunk and not rpn.x6_trunk and m._stage1_feature_extractor.x3_layers == ()
    m.alone_winograd_x3f_layers = ("conv5_1", "rpn_trunk")
    assert m.layer_tables(0) == (tuple(n for n in x6 if n not in ("conv5_1", "rpn_trunk")),) * 2 + (x3f + ("conv5_1", "rpn_trunk"),)
    m.alone_winograd_x3f_layers = ()
    assert m.layer_tables(0) == (x6, x3, x3f) and rpn.x3_trunk and not rpn.x3f_trunk and m._stage1_feature_extractor.x3f_layers == x3f
    assert m._slot_masks(0) == (m._x6_mask(), m._x3_mask(), m._x3f_mask())
    m6, m3, mf = m._slot_masks(3)
    assert m6 == 0 and m3 == 0 and mf == m._x3f_mask() | m._x6_mask() and mf & (1 << nv.X6_RPN_TRUNK_BIT)
    p0, p1 = m._forward_params(0), m._forward_params(2)
    assert (p0.winograd_x6_mask, p0.winograd_x3_mask, p0.winograd_x3f_mask) == m._slot_masks(0)
    assert (p1.winograd_x6_mask, p1.winograd_x3_mask, p1.winograd_x3f_mask) == m._slot_masks(2)
    m.winograd_x3_layers = tuple(n for n in x3 if n != "conv5_1")      # an f32x6 layer is not moved (its bank is not an x3 blob)
    assert m.layer_tables(1) == (("conv5_1",), (), x3f + tuple(n for n in x6 if n != "conv5_1"))
    m.winograd_x3_layers = x3
    m.inflight_winograd_x3f_layers = ("conv4_2",)
    assert m.layer_tables(1)[2] == x3f + ("conv4_2",) and "conv4_2" not in m.layer_tables(1)[0]
    with pytest.raises(ValueError):
        m.inflight_winograd_x3f_layers = ("conv9_9",)
    m.inflight_winograd_x3f_layers = x6
    m.winograd_x6_layers = x6 + ("conv3_2",)                  # a layer in both tables runs as the three-launch x6 / x3 layer
    assert m._x3f_mask() & (1 << nv.X6_LAYER_BITS["conv3_2"]) == 0 and "conv3_2" not in m._stage1_feature_extractor.x3f_layers
    m.winograd_x6_layers = x6
    m.math_mode = "f32"
    assert m._x3f_mask() == 0
    m.math_mode = "f32_winograd"
    with pytest.raises(ValueError):
        m.winograd_x3f_layers = ("rpn_trunk",)
    m.winograd_x6_layers = ("conv4_2",)                       # the x3 table is an overlay: names outside the x6 table have no effect ...
    assert m._x3_mask() == 1 << nv.X6_LAYER_BITS["conv4_2"] and m._stage1_feature_extractor.x3_layers == ("conv4_2",)
    m.winograd_x6_layers = x6                                 # ... and come back with it
    assert m._x3_mask() == sum(1 << nv.X6_LAYER_BITS[n] for n in x3)
    with pytest.raises(ValueError):
        m.winograd_x3_layers = ("conv1_2",)


def test_resnet_default_modes_and_conv_table():
    """ResNet defaults: layer4 head + RPN trunk as f32x3 GEMMs; bench.py's convolution table follows the model's modes."""
    import bench
    from fasterrcnn_amd.models import resnet
    m = FasterRCNNModel(num_classes=21, backbone=resnet.ResNetBackbone(resnet.Architecture.ResNet50))
    assert (m.x6_conv1x1, m.x6_conv1x1_arith, m.winograd_x6_layers, m.winograd_x3_layers) == ("head", "f32x3", ("rpn_trunk",), ("rpn_trunk",))
    assert m.bottleneck_g3 == "backbone" and m._stage1_feature_extractor.g3 and not m._stage3_detector_network._pool_to_feature_vector.g3
    m101 = FasterRCNNModel(num_classes=21, backbone=resnet.ResNetBackbone(resnet.Architecture.ResNet101))
    assert m101.bottleneck_g3 == "off" and not m101._stage1_feature_extractor.g3        # admitted by the criterion, not the default
    del m101
    assert m._x6_mask() == m._x3_mask() == 1 << 13
    table = bench.resnet_conv_table(m)
    head = [r for r in table if r[0] == "head" and r[2].startswith("gemm_")]
    assert head and all(r[2].startswith("gemm_x3t_kernel") and r[3] == "f16" and abs(r[4] / r[5] - 3.0) < 1e-9 or "Winograd" in r[2] for r in head)
    trunk = [r for r in table if r[1] == "rpn_trunk"][0]
    assert trunk[2].startswith("gemm_x3t_kernel") and trunk[3] == "f16"
    assert all(r[2] == "conv_gather_x3_kernel" and r[3] == "f16" and abs(r[4] / r[5] - 3.0) < 1e-9 for r in table if r[0] == "backbone" and "stem" not in r[1])
    m.bottleneck_g3 = "off"
    assert all(r[3] in ("f32", "valu") for r in bench.resnet_conv_table(m) if r[0] == "backbone")
    with pytest.raises(ValueError):
        m.bottleneck_g3 = "layer3"
    m.x6_conv1x1_arith = "f32x6"
    m.winograd_x3_layers = ()
    table6 = bench.resnet_conv_table(m)
    assert [r for r in table6 if r[1] == "rpn_trunk"][0][3] == "bf16"
    assert all(r[3] == "bf16" for r in table6 if r[0] == "head" and r[2].startswith("gemm_"))
    with pytest.raises(NotImplementedError):
        FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0)).x6_conv1x1_arith = "f32x3"


class _FakeHandle:
    def __init__(self, log, tag):
        self.log, self.tag = log, tag

    def result(self):
        self.log.append(("collect",) + self.tag)
        return {1: np.zeros((0, 5))}


class _FakeBatchModel:
    """Stands in for a ResNet FasterRCNNModel: records which lane every batch went to and refuses a lane whose images are uncollected."""
    _is_resnet = True

    def __init__(self):
        self.log, self.busy = [], {}

    def predict_batch_async(self, images, score_threshold, lane=0):
        assert not self.busy.get(lane), "lane %d reused while its images are in flight" % lane
        n = int(images.shape[0])
        self.log.append(("batch", lane, n, tuple(images.shape[1:])))
        handles = [_FakeHandle(self.log, (lane, i)) for i in range(n)]
        self.busy[lane] = n
        model = self

        def make(hd):
            inner = hd.result

            def result():
                model.busy[lane] -= 1
                return inner()
            hd.result = result
            return hd
        return [make(h) for h in handles]


def test_evaluate_stream_batches_group_by_shape_and_never_reuse_a_busy_lane():
    from fasterrcnn_amd import evaluate as ev
    model = _FakeBatchModel()
    shapes = [(3, 8, 8)] * 5 + [(3, 8, 12)] * 3 + [(3, 8, 8)] * 4          # a shape change in the middle of a batch, twice
    samples = [(i, torch.zeros((1,) + shp), None) for i, shp in enumerate(shapes)]
    got = []
    ev.evaluate_stream(model, samples, inflight=8, batch=4, on_result=lambda i, d: got.append(i))
    assert got == list(range(12))                                          # results in image order
    batches = [e for e in model.log if e[0] == "batch"]
    assert [b[2] for b in batches] == [4, 1, 3, 4] and [b[1] for b in batches] == [0, 1, 0, 1]
    assert all(v == 0 for v in model.busy.values())
    # one lane only: every batch is collected before the next is enqueued; ranks take every world-th sample
    model = _FakeBatchModel()
    got = []
    ev.evaluate_stream(model, samples, inflight=2, batch=4, rank=1, world=2, on_result=lambda i, d: got.append(i))
    assert got == [i for i in range(12) if i % 2 == 1] and all(b[1] == 0 for b in model.log if b[0] == "batch")
    # a VGG-16 model ignores `batch`
    class _Vgg:
        _is_resnet = False

        def __init__(self):
            self.calls = 0

        def predict_async(self, image, thr, slot):
            self.calls += 1
            return _FakeHandle([], (slot, 0))
    v = _Vgg()
    ev.evaluate_stream(v, samples[:3], inflight=2, batch=4)
    assert v.calls == 3


def test_runtime_feature_map_shape_matches_the_backbone_contract():
    """runtime.feature_map_shape (used to address the per-image maps of a batch) == ResNetBackbone.compute_feature_map_shape
    (models/resnet.py:161-185: ceil(H / 16), ceil(W / 16)) == the four stride-2 stages applied one by one."""
    from fasterrcnn_amd import runtime as rt
    from fasterrcnn_amd.models import resnet
    bb = resnet.ResNetBackbone(resnet.Architecture.ResNet50)
    for h in (32, 33, 47, 224, 333, 599, 600, 601, 1000, 1333):
        for w in (32, 49, 320, 517, 1000, 1001):
            c, fh, fw = bb.compute_feature_map_shape((3, h, w))
            assert rt.feature_map_shape(h, w) == (fh, fw) and c == 1024


def test_one_layer_table_round_trips_through_the_deprecated_attributes():
    """set_layer_forms / layer_forms (round 6): one {layer: form} table for every slot; the five per-kind attributes are its storage."""
    from fasterrcnn_amd import _native as nv
    from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
    from fasterrcnn_amd.models.vgg16 import VGG16Backbone
    m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
    default = m.layer_forms(0)
    assert default == m.layer_forms(1) == {n: "f32x3_one_launch" for n in nv.X6_LAYER_BITS}      # round 5: one table, all 13 layers one-launch f32x3
    masks = (m._slot_masks(0), m._slot_masks(1))
    m.set_layer_forms(default)
    assert (m._slot_masks(0), m._slot_masks(1)) == masks and m.layer_forms(1) == default
    mixed = {"conv1_2": "f32x3_one_launch", "conv3_2": "f32x6", "conv4_1": "f32x3", "conv5_3": "f32x3_one_launch", "rpn_trunk": "f32x3"}
    m.set_layer_forms(mixed)
    want = {n: mixed.get(n, "f32") for n in nv.X6_LAYER_BITS}
    assert m.layer_forms(0) == want and m.layer_forms(2) == want
    import pytest
    with pytest.raises(ValueError):
        m.set_layer_forms({"conv1_1": "f32x3"})
    with pytest.raises(ValueError):
        m.set_layer_forms({"conv4_1": "bf16"})
