"""tools/exp_conv_x3g.py -- every distinct convolution of the ResNet-50 feature extractor at 600x1000 (one map), timed in the exact-f32
gather kernel, the float32 one-launch Winograd kernel (stride-1 3x3) and the f32x3 tensor-scale kernel (frcnn_conv_nhwc_x3g), next to
the time its algorithmic bytes take at 8 TB/s.  Development aid."""
import sys

sys.path.insert(0, ".")
import torch

from fasterrcnn_amd import _native as nv
from fasterrcnn_amd.models import resnet as R

SHAPES = [  # name, h, w, cin, cout, k, stride, residual, count per image (ResNet-50)
    ("l1.conv1 64>64", 150, 250, 64, 64, 1, 1, False, 1),
    ("l1.conv1 256>64", 150, 250, 256, 64, 1, 1, False, 2),
    ("l1.conv2 3x3 64", 150, 250, 64, 64, 3, 1, False, 3),
    ("l1.conv3 64>256", 150, 250, 64, 256, 1, 1, True, 3),
    ("l1.down 64>256", 150, 250, 64, 256, 1, 1, False, 1),
    ("l2.0.conv1 256>128", 150, 250, 256, 128, 1, 1, False, 1),
    ("l2.0.conv2 3x3s2 128", 150, 250, 128, 128, 3, 2, False, 1),
    ("l2.down 256>512 s2", 150, 250, 256, 512, 1, 2, False, 1),
    ("l2.conv1 512>128", 75, 125, 512, 128, 1, 1, False, 3),
    ("l2.conv2 3x3 128", 75, 125, 128, 128, 3, 1, False, 3),
    ("l2.conv3 128>512", 75, 125, 128, 512, 1, 1, True, 4),
    ("l3.0.conv1 512>256", 75, 125, 512, 256, 1, 1, False, 1),
    ("l3.0.conv2 3x3s2 256", 75, 125, 256, 256, 3, 2, False, 1),
    ("l3.down 512>1024 s2", 75, 125, 512, 1024, 1, 2, False, 1),
    ("l3.conv1 1024>256", 38, 63, 1024, 256, 1, 1, False, 5),
    ("l3.conv2 3x3 256", 38, 63, 256, 256, 3, 1, False, 5),
    ("l3.conv3 256>1024", 38, 63, 256, 1024, 1, 1, True, 6),
]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    lib = nv.lib()
    tot = {"f32": 0.0, "x3g": 0.0, "best_f32": 0.0, "bytes": 0.0}
    for name, h, w, cin, cout, k, stride, res, count in SHAPES:
        pad = 1 if k == 3 else 0
        ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        x = torch.randn(1, h, w, cin, device="cuda").relu()
        wp = torch.randn(k * k, cout, cin, device="cuda") / (cin * k * k) ** 0.5
        b = torch.randn(cout, device="cuda")
        r = torch.randn(1, ho, wo, cout, device="cuda") if res else None
        y = torch.empty(1, ho, wo, cout, device="cuda")
        wsb = int(lib.frcnn_conv_workspace_bytes(1, h, w, cin, cout, k, stride, pad))
        ws = torch.empty(max(wsb, 4) // 4, device="cuda")
        xm, wm, ym = R.tensor_absmax(x), R.tensor_absmax(wp), torch.zeros(1, device="cuda")
        sp = nv.stream_ptr()

        def f32():
            nv.check(lib.frcnn_conv_nhwc(nv.ptr(x), nv.ptr(wp), nv.ptr(b), nv.ptr(r), nv.ptr(y), 1, h, w, cin, cout, k, stride, pad, nv.RELU, nv.ptr(ws), wsb, sp), "f32")

        def x3g():
            nv.check(lib.frcnn_conv_nhwc_x3g(nv.ptr(x), nv.ptr(wp), nv.ptr(b), nv.ptr(r), nv.ptr(y), 1, h, w, cin, cout, k, stride, pad, nv.RELU,
                                             nv.ptr(xm), nv.ptr(wm), nv.ptr(ym), nv.ptr(ws), wsb, sp), "x3g")
        t32, t3 = timeit(f32), timeit(x3g)
        tw = None
        if k == 3 and stride == 1 and nv.resnet_block_uses_winograd_fused(1, cin, 1):
            u = torch.empty(16 * cout * cin, device="cuda")
            ones = torch.ones(cout, device="cuda")
            w4 = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
            nv.check(lib.frcnn_pack_conv3x3_winograd_fused(nv.ptr(w4), nv.ptr(ones), nv.ptr(u), cout, cin, sp), "pack")

            def wino():
                nv.check(lib.frcnn_conv3x3_nhwc_winograd_fused(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), h, w, cin, cout, nv.RELU, sp), "wino")
            tw = timeit(wino)
        byts = 4.0 * (h * w * cin + ho * wo * cout * (2 if res else 1) + k * k * cin * cout)
        best = min(t32, tw) if tw else t32
        print("%-24s M=%6d  f32 %7.1f us  wino %s  x3g %7.1f us   bytes %6.1f MB = %5.1f us at 8 TB/s   x%d" % (
            name, ho * wo, t32, ("%7.1f" % tw) if tw else "      -", t3, byts / 1e6, byts / 8e6, count), flush=True)
        tot["f32"] += t32 * count
        tot["best_f32"] += best * count
        tot["x3g"] += t3 * count
        tot["bytes"] += byts / 8e6 * count
    print("per image: f32 gather %.0f us, best float32 %.0f us, x3g %.0f us, bytes at 8 TB/s %.0f us" % (tot["f32"], tot["best_f32"], tot["x3g"], tot["bytes"]))


if __name__ == "__main__":
    main()
