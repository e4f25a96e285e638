"""CPU side of the stress parity set (tests/golden/stress/, oracle/make_stress.py): the fixtures are there, the recipes that regenerate their
inputs on the GPU box are deterministic, and the oracle reproduces the imported reference's outputs stored in a fixture bit for bit."""
import glob
import hashlib
import os

import numpy as np
import torch

from fasterrcnn_amd import synthetic
from oracle import frcnn_oracle as O

STRESS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stress")


def test_fixture_inventory():
    vgg = sorted(glob.glob(os.path.join(STRESS, "vgg16_*.npz")))
    r50 = sorted(glob.glob(os.path.join(STRESS, "resnet50_*.npz")))
    assert len(vgg) >= 8 and len(r50) >= 4
    kinds = {str(np.load(f)["kind"]) for f in vgg + r50}
    assert kinds == set(synthetic.STRESS_KINDS)
    for f in vgg + r50:
        g = np.load(f)
        assert g["ref_proposals"].shape == (300, 4) and g["ref_detections"].shape[0] > 0, f
        assert int(g["n_unique_top_scores"]) >= 0.99 * int(g["n_top_scores"]), "the recipe must not test the sort's tie rule: %s" % f
        if str(g["kind"]) == "outlier":
            assert float(g["planted_max"]) >= 2.0 ** 12 * float(g["planted_median"]), f


def test_recipes_are_deterministic_and_what_they_say():
    a = synthetic.stress_frame_u8(504, "edges")
    b = synthetic.stress_frame_u8(504, "edges")
    assert torch.equal(a, b) and a.dtype == torch.uint8 and tuple(a.shape) == (3, 600, 1000)
    assert int((a == 0).all(dim=0).sum()) > 10000 and int((a == 255).all(dim=0).sum()) > 10000        # flat black and saturated regions
    dx = (a[:, :, 1:].to(torch.int16) - a[:, :, :-1].to(torch.int16)).abs().amax(dim=0)
    assert int((dx >= 128).sum()) > 2000                                                                 # hard edges
    assert hashlib.sha256(a.numpy().tobytes()).hexdigest()[:16] == hashlib.sha256(b.numpy().tobytes()).hexdigest()[:16]
    sd = synthetic.stress_vgg16_state_dict(7003, "outlier")
    w = sd["_stage1_feature_extractor._block3_conv2.weight"]
    norms = w.flatten(1).norm(dim=1)
    assert float(norms.max() / norms.median()) > 3000.0                                                 # ONE output channel x4096
    sh = synthetic.stress_vgg16_state_dict(7001, "heavy")
    k = sh["_stage1_feature_extractor._block3_conv1.weight"].flatten()
    kurt = float(((k - k.mean()) ** 4).mean() / k.var() ** 2)
    assert kurt > 20.0                                                                                   # heavy tails (a normal: 3)
    assert ("VGG16", "heavy", 7001) in synthetic.STRESS_CALIBRATION and ("ResNet50", "outlier", 7103) in synthetic.STRESS_CALIBRATION


def test_oracle_reproduces_a_stress_fixture():
    g = np.load(os.path.join(STRESS, "vgg16_edges_s504_w1234.npz"))
    sd = synthetic.stress_vgg16_state_dict(int(g["weights_seed"]), str(g["kind"]))
    img = synthetic.stress_image(int(g["seed"]), str(g["kind"])).unsqueeze(0)
    with torch.no_grad():
        props, classes, deltas = O.forward(sd, img)
    assert np.array_equal(props.numpy(), g["ref_proposals"])
