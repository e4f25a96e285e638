"""tools/power_probe.py -- board power and shader clock while ONE layer of the one-launch f32x3 Winograd kernel runs back to back
(round 6: is the kernel's wall time set by cycles or by the power cap?).  Use with FRCNN_LIB_PATH=<variant>:
    python tools/power_probe.py conv4_2 [seconds]
Prints: launches/s -> us per launch, mean / max board power (hwmon power1_average or rocm-smi), mean sclk (hwmon freq1_input or rocm-smi)."""
import glob
import json
import subprocess
import sys
import threading
import time

import torch as t

sys.path.insert(0, ".")
from fasterrcnn_amd import _native as nv  # noqa: E402

LAYERS = {"conv1_2": (64, 64, 600, 1000, True), "conv2_2": (128, 128, 300, 500, True), "conv3_2": (256, 256, 150, 250, False),
          "conv4_2": (512, 512, 75, 125, False), "conv5_x": (512, 512, 37, 62, False)}


def _hwmon():
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        p = glob.glob(d + "/power1_average") or glob.glob(d + "/power1_input")
        f = glob.glob(d + "/freq1_input")
        if p:
            return p[0], (f[0] if f else None)
    return None, None


def sampler(stop, out):
    p, f = _hwmon()
    while not stop.is_set():
        try:
            if p:
                w = int(open(p).read()) / 1e6
                mhz = int(open(f).read()) / 1e6 if f else float("nan")
            else:
                j = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True).stdout)
                c = next(iter(j.values()))
                w = float(next(v for k, v in c.items() if "ower" in k and "W" in k))
                mhz = float(next(v for k, v in c.items() if "sclk" in k).strip("()Mhz "))
            out.append((w, mhz))
        except Exception as e:  # noqa: BLE001
            out.append((float("nan"), float("nan")))
            if len(out) < 3:
                print("sampler:", repr(e), file=sys.stderr)
        time.sleep(0.1)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "conv4_2"
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
    cin, cout, h, w, pool = LAYERS[name]
    dev = t.device("cuda:0")
    lib = nv.lib()
    s = nv.stream_ptr()
    x = t.randn((h, w, cin), device=dev).clamp(min=0)
    wt = t.randn((cout, cin, 3, 3), device=dev) * 0.02
    b = t.zeros((cout,), device=dev)
    bank = t.empty((16, cout, cin), device=dev)
    u = t.empty((int(lib.frcnn_conv3x3_winograd_x3_pack_bytes(cout, cin)),), dtype=t.int8, device=dev)
    nv.check(lib.frcnn_pack_conv3x3_winograd(nv.ptr(wt), None, nv.ptr(bank), cout, cin, s), "pack")
    nv.check(lib.frcnn_pack_conv3x3_winograd_x3(nv.ptr(bank), nv.ptr(u), cout, cin, s), "pack_x3")
    oh, ow = (h // 2, w // 2) if pool else (h, w)
    y = t.zeros((oh * ow * cout + 16 * 4096,), device=dev)
    wsb = int(lib.frcnn_conv3x3_winograd_x3_fused_workspace_bytes(1, h, w))
    ws = t.empty((wsb,), dtype=t.uint8, device=dev)
    cm = t.empty((h, w), device=dev)
    nv.check(lib.frcnn_pixel_absmax(nv.ptr(x), nv.ptr(cm), h * w, cin, s), "absmax")
    flags = nv.RELU | (nv.POOL2 if pool else 0) | nv.X3F_WAVES4

    def launch(n):
        for _ in range(n):
            nv.check(lib.frcnn_conv3x3_nhwc_winograd_x3_chain(nv.ptr(x), nv.ptr(u), nv.ptr(b), nv.ptr(y), 1, h, w, cin, cout, flags, 1, nv.ptr(ws), wsb,
                                                              nv.ptr(cm), None, s), "x3_chain")
    launch(200)
    t.cuda.synchronize()
    t0 = time.time()
    launch(500)
    t.cuda.synchronize()
    per = (time.time() - t0) / 500
    n = max(500, int(secs / per))
    stop, out = threading.Event(), []
    th = threading.Thread(target=sampler, args=(stop, out))
    e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    th.start()
    e0.record()
    launch(n)
    e1.record()
    t.cuda.synchronize()
    stop.set()
    th.join()
    us = e0.elapsed_time(e1) * 1e3 / n
    ws_ = [o[0] for o in out[len(out) // 4:] if o[0] == o[0]]
    fs = [o[1] for o in out[len(out) // 4:] if o[1] == o[1]]
    print("%-8s %s: %7.1f us per launch over %d launches | board power mean %.0f W max %.0f W | sclk mean %.0f MHz (%d samples)"
          % (name, nv.LIB_PATH.split("/")[-1], us, n, sum(ws_) / max(1, len(ws_)), max(ws_ or [0]), sum(fs) / max(1, len(fs)), len(ws_)))


if __name__ == "__main__":
    main()
