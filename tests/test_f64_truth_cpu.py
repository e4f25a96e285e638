"""
CPU checks of the float64 yardstick (oracle/f64_truth.py) and of the held-out fixtures it produced (tests/golden/holdout/, written by
oracle/make_holdout.py from the imported reference in the build container).  No GPU, no /root/reference.
"""
import glob
import os

import numpy as np
import torch

from fasterrcnn_amd import synthetic
from oracle import f64_truth as T
from oracle import frcnn_oracle as O

HOLDOUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "holdout")


def test_truth_is_the_float32_path_up_to_float32_noise(sd_cpu):
    """On a small image: the float64 evaluation and the oracle's float32 one are the same function -- feature map within float32 noise,
    the same proposals row for row (no near-tie flips on this image) within 1e-3 px, every float32 row the decode of a truth candidate,
    detections matched class by class."""
    img = synthetic.image(3, 224, 320).unsqueeze(0)
    detail = {}
    props, classes, deltas = O.forward(sd_cpu, img, detail=detail)
    truth = T.forward(sd_cpu, img, score_threshold=0.05)
    fm64 = truth["feature_map"]
    assert fm64.dtype == torch.float64 and truth["proposals"].dtype == np.float64
    rel = float((detail["feature_map"].double() - fm64).abs().max()) / float(fm64.abs().max())
    assert 0 < rel <= 5e-6                                     # float32 rounding is visible (> 0) and small
    assert float((detail["scores"].double() - truth["scores"]).abs().max()) <= 1e-5
    cand = truth["clipped"][torch.from_numpy(truth["sorted_idx"])].numpy()
    err, idx = T.proposal_errors(props.numpy(), cand)
    s = T.summarize(err)
    assert s["n_far"] == 0 and s["max"] <= 1e-3 and s["median"] <= 2e-4
    assert truth["proposals"].shape == tuple(props.shape)
    assert float(np.abs(truth["proposals"] - props.numpy().astype(np.float64)).max()) <= 1e-3      # same rows, same order
    det32 = O.detections(props.numpy(), classes.numpy(), deltas.numpy(), 224, 320, 0.05)
    rows32 = np.vstack([np.hstack([np.full((len(v), 1), float(c)), v]) for c, v in sorted(det32.items()) if len(v)])
    rows64 = np.vstack([np.hstack([np.full((len(v), 1), float(c)), v]) for c, v in sorted(truth["detections"].items()) if len(v)])
    b_err, s_err = T.detection_errors(rows32, rows64)
    assert np.isfinite(b_err).mean() >= 0.98 and np.median(b_err[np.isfinite(b_err)]) <= 2e-4


def test_error_helpers():
    truth = np.array([[0.0, 0.0, 10.0, 10.0], [5.0, 5.0, 50.0, 60.0]])
    run = np.array([[5.0 + 2e-4, 5.0, 50.0, 60.0 - 1e-4], [0.0, 0.0, 10.0, 10.0], [100.0, 100.0, 200.0, 200.0]], dtype=np.float32)
    err, idx = T.proposal_errors(run, truth)
    assert idx.tolist()[:2] == [1, 0] and abs(err[0] - 2e-4) < 1e-5 and err[1] == 0.0 and err[2] > 50
    s = T.summarize(err)
    assert s["n"] == 3 and s["n_far"] == 1 and s["beyond_gate"] == 0
    d = np.array([[3.0, 0, 0, 10, 10, 0.9], [4.0, 0, 0, 10, 10, 0.8]])
    t_ = np.array([[3.0, 0, 0, 10, 10 + 1e-4, 0.9]])
    b, sc = T.detection_errors(d, t_)
    assert abs(b[0] - 1e-4) < 1e-9 and sc[0] == 0.0 and np.isinf(b[1])


def test_holdout_fixtures_are_complete_and_self_consistent():
    """32 cases; in every one the reference's rows are decodes of truth candidates (recorded error == recomputed error), the reference's
    distance from the truth is float32 noise (median < 3e-4 px), and seeds are the held-out ones."""
    files = sorted(glob.glob(os.path.join(HOLDOUT, "*.npz")))
    assert len(files) == 32
    archs = {}
    for f in files:
        g = np.load(f)
        arch = str(g["arch"])
        archs[arch] = archs.get(arch, 0) + 1
        assert int(g["height"]) == 600 and int(g["width"]) == 1000 and int(g["seed"]) >= 101
        err, idx = T.proposal_errors(g["ref_proposals"], g["truth_cand_boxes"])
        assert np.array_equal(idx, g["ref_prop_candidate"]) and np.allclose(err, g["ref_prop_err"], rtol=0, atol=1e-12)
        s = T.summarize(err)
        assert s["n_far"] == 0 and s["median"] <= 3e-4 and s["max"] <= 2e-3
        assert g["truth_cand_boxes"].dtype == np.float64 and g["truth_prop_pos"].max() < len(g["truth_cand_anchor"])
    assert archs == {"VGG16": 16, "ResNet50": 8, "ResNet101": 8}


def test_one_holdout_fixture_regenerates_from_the_seeds(sd_cpu):
    """The oracle's float32 run and the float64 truth, recomputed HERE from the seeds of one fixture, give the fixture's vectors: the truth
    candidates to 1e-9 px (float64 BLAS summation order may differ between hosts), the reference's proposals to float32 noise."""
    g = np.load(os.path.join(HOLDOUT, "vgg16_600x1000_s101_w1234.npz"))
    img = synthetic.image(int(g["seed"]), 600, 1000).unsqueeze(0)
    props, _, _ = O.forward(sd_cpu, img)
    assert props.shape == g["ref_proposals"].shape
    assert float(np.abs(props.numpy() - g["ref_proposals"]).max()) <= 1e-3
    truth = T.forward(sd_cpu, img, score_threshold=0.05)
    cand = truth["clipped"][torch.from_numpy(g["truth_cand_anchor"].astype(np.int64))].numpy()
    assert float(np.abs(cand - g["truth_cand_boxes"]).max()) <= 1e-9
