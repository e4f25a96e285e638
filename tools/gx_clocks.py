"""tools/gx_clocks.py -- where a block of conv_gather_x3_kernel (csrc/conv_gather.hip) spends its time, per ResNet-50 backbone shape.
Needs a library built with -DGX_CLOCKS:  SRC=conv_gather tools/build_ablate.sh gxclk -DGX_CLOCKS ; FRCNN_LIB_PATH=build/libfrcnn_gxclk.so"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from fasterrcnn_amd import _native as nv
from fasterrcnn_amd.models import resnet as R
from exp_conv_x3g import SHAPES


# --contiguity: 3x3 convolutions whose operand rows are (cin = 32: a pixel's stage is its whole 128-byte row, neighbouring pixels and filter
# rows adjoin) or are not (cin = 64, 256: 128 bytes out of every 256 / 1024) contiguous in memory -- does the fetch rate depend on it?
CONTIGUITY = [("3x3 cin 32 > 64", 150, 250, 32, 64, 3, 1, False, 1), ("3x3 cin 64 > 64", 150, 250, 64, 64, 3, 1, False, 1),
              ("3x3 cin 32 > 128", 75, 125, 32, 128, 3, 1, False, 1), ("3x3 cin 128 > 128", 75, 125, 128, 128, 3, 1, False, 1),
              ("3x3 cin 32 > 256", 38, 63, 32, 256, 3, 1, False, 1), ("3x3 cin 256 > 256", 38, 63, 256, 256, 3, 1, False, 1)]


def main():
    lib = nv.lib()
    for name, h, w, cin, cout, k, stride, res, count in (CONTIGUITY if "--contiguity" in sys.argv else SHAPES):
        pad = 1 if k == 3 else 0
        ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        x = torch.randn(1, h, w, cin, device="cuda").relu()
        wp = torch.randn(k * k, cout, cin, device="cuda") / (cin * k * k) ** 0.5
        b = torch.randn(cout, device="cuda")
        r = torch.randn(1, ho, wo, cout, device="cuda") if res else None
        nrec = 8 * 8 * 4096
        y = torch.zeros(ho * wo * cout + nrec, device="cuda")
        xm, wm, ym = R.tensor_absmax(x), R.tensor_absmax(wp), torch.zeros(1, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rep in range(10):
            if rep == 9:
                e0.record()
            # no workspace: the un-split form, whose blocks write the stamps
            nv.check(lib.frcnn_conv_nhwc_x3g(nv.ptr(x), nv.ptr(wp), nv.ptr(b), nv.ptr(r), nv.ptr(y), 1, h, w, cin, cout, k, stride, pad, nv.RELU,
                                             nv.ptr(xm), nv.ptr(wm), nv.ptr(ym), None, 0, nv.stream_ptr()), "x3g")
        e1.record()
        torch.cuda.synchronize()
        o = y[ho * wo * cout:].view(-1, 8).cpu().numpy().astype(np.float64)
        o = o[o[:, 7] == 1.0]
        span = ((o[:, 5].max() - o[:, 4].min()) % (1 << 24)) / 100.0
        print("%-24s %5.1f us launch-to-launch | %4d blocks, %3d stages | block medians: setup %5.2f  fetch+first %5.2f  loop %6.2f (%.3f / stage)  epilogue %5.2f us | "
              "block total median %6.2f max %6.2f | first in -> last out %6.2f us" % (
                  name, e0.elapsed_time(e1) * 1e3, len(o), int(o[0, 6]), np.median(o[:, 0]) / 100, np.median(o[:, 1]) / 100, np.median(o[:, 2]) / 100,
                  np.median(o[:, 2]) / 100 / max(o[0, 6], 1), np.median(o[:, 3]) / 100, np.median(o[:, :4].sum(1)) / 100, o[:, :4].sum(1).max() / 100, span), flush=True)


if __name__ == "__main__":
    main()
