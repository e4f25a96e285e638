"""oracle/train_oracle.py, grad_math="bf16" (the checker of the reduced-precision train step): the forward pass and the losses are the
float32 step's bit for bit; every gradient moves by the bfloat16 rounding of the GEMM operands -- relative 2^-9 per operand, averaged
down by the reduction -- and by nothing larger."""
import random

import numpy as np
import torch

from fasterrcnn_amd import synthetic
from oracle import frcnn_oracle as O
from oracle import train_oracle as TO


def test_bf16_oracle_rounds_only_the_gradient_gemms(sd_cpu):
    h, w, seed = 352, 480, 4
    img = synthetic.image(seed, h, w).unsqueeze(0)
    gts = synthetic.ground_truth(seed, h, w)
    am, vm = O.generate_anchor_maps((3, h, w), (512, h // 16, w // 16), 16)
    rmap, obj, bg = O.generate_rpn_map(am, vm, np.stack([k for _, k in gts]))
    res = {}
    for gm in ("f32", "bf16"):
        random.seed(5); torch.manual_seed(5)
        res[gm] = TO.train_step(sd_cpu, img, am, vm, torch.from_numpy(rmap).unsqueeze(0), obj, bg, np.stack([k for _, k in gts]),
                                np.array([c for c, _ in gts]), 21, 1e-6, 0.9, 5e-4, grad_math=gm)
    assert res["f32"][0] == res["bf16"][0]
    for k, g in res["f32"][1].items():
        shift = float((g - res["bf16"][1][k]).abs().max()) / float(g.abs().max())
        assert 1e-4 <= shift <= 2e-2, (k, shift)
    assert O.CONV_BN is None, "the conv+BN hook is restored after the step"
    x = torch.randn(3, 7)
    assert torch.equal(TO._bf16r(x), x.to(torch.bfloat16).float())
