"""tools/exp_r50_global_scale.py -- CPU experiment (build container or anywhere with the fixtures): would a ResNet backbone whose convolutions
split their operands into two fp16 terms under ONE power-of-two scale PER TENSOR (instead of per row / tile) still pass the held-out
criterion?  Emulates the operand perturbation in the oracle (float32 accumulation by torch) and measures the proposals against the float64
truth stored in tests/golden/holdout/.  Development aid; not a test."""
import glob
import os
import sys

sys.path.insert(0, ".")
import numpy as np
import torch as t
from torch.nn import functional as F

from fasterrcnn_amd import synthetic
from oracle import f64_truth as T
from oracle import frcnn_oracle as O


def split2(x):
    m = float(x.abs().max())
    if m == 0.0:
        return x
    s = 2.0 ** (14 - int(np.floor(np.log2(m))))          # max |x| s in [2^14, 2^15)
    xs = x * s
    hi = xs.half().float()
    lo = (xs - hi).half().float()
    return (hi + lo) / s


def conv_bn_x3(x, sd, wkey, bn_prefix, stride, padding):
    scale = sd[bn_prefix + "weight"] / t.sqrt(sd[bn_prefix + "running_var"] + 1e-5)
    shift = sd[bn_prefix + "bias"] - sd[bn_prefix + "running_mean"] * scale
    wf = sd[wkey] * scale.reshape(-1, 1, 1, 1)
    return F.conv2d(split2(x), split2(wf), stride=stride, padding=padding) + shift.reshape(1, -1, 1, 1)


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "ResNet50"
    files = sorted(glob.glob(os.path.join("tests", "golden", "holdout", "%s_*.npz" % arch.lower())))[: int(sys.argv[2]) if len(sys.argv) > 2 else 3]
    t.set_num_threads(16)
    rows = []
    for f in files:
        g = np.load(f)
        sd = synthetic.resnet_state_dict(int(g["weights_seed"]), arch)
        img = synthetic.image_rgb(int(g["seed"]), int(g["height"]), int(g["width"])).unsqueeze(0)
        out = {}
        for name, hook in (("f32", None), ("x3_global", conv_bn_x3)):
            O.CONV_BN = hook
            try:
                with t.no_grad():
                    props, classes, deltas = O.forward(sd, img)
            finally:
                O.CONV_BN = None
            err, _ = T.proposal_errors(props.numpy(), g["truth_cand_boxes"])
            out[name] = T.summarize(err)
        ref = T.summarize(g["ref_prop_err"])
        rows.append((out, ref))
        print("%s: reference med %.3g p95 %.3g | oracle f32 med %.3g p95 %.3g | per-tensor-scale f32x3 backbone med %.3g p95 %.3g (x%.2f / x%.2f of the reference)" % (
            os.path.basename(f), ref["median"], ref["p95"], out["f32"]["median"], out["f32"]["p95"], out["x3_global"]["median"], out["x3_global"]["p95"],
            out["x3_global"]["median"] / ref["median"], out["x3_global"]["p95"] / ref["p95"]), flush=True)
    med = np.median([o["x3_global"]["median"] for o, _ in rows]) / np.median([r["median"] for _, r in rows])
    p95 = np.median([o["x3_global"]["p95"] for o, _ in rows]) / np.median([r["p95"] for _, r in rows])
    print("pooled: x%.2f / x%.2f" % (med, p95))


if __name__ == "__main__":
    main()
