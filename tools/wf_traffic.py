"""Launches every one-launch Winograd layer shape of VGG-16 at 600x1000 a few times -- the workload of the FETCH_SIZE / WRITE_SIZE
counter passes that compare block -> XCD mappings (FRCNN_WF_XCL=0..3):
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o p -- python tools/wf_traffic.py
  python tools/wf_traffic.py --summarize out/p_counter_collection.csv"""
import collections
import csv
import sys

LAYERS = [("conv1_2", 600, 1000, 64, 64), ("conv2_1", 300, 500, 64, 128), ("conv2_2", 300, 500, 128, 128), ("conv3_1", 150, 250, 128, 256),
          ("conv3_2", 150, 250, 256, 256), ("conv4_1", 75, 125, 256, 512), ("conv4_2", 75, 125, 512, 512), ("conv5_x", 37, 62, 512, 512)]


def summarize(path):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(list)
    for r in rows:
        if "wino_fused_kernel" in r["Kernel_Name"]:
            agg[r["Dispatch_Id"]].append(r)
    disp = sorted(agg, key=int)
    vals = [sum(float(r["Counter_Value"]) for r in agg[d]) for d in disp]
    reps = len(vals) // len(LAYERS)
    name = rows[0]["Counter_Name"]
    for i, (lname, h, w, cin, cout) in enumerate(LAYERS):
        v = vals[i * reps:(i + 1) * reps]
        alg = (h * w * cin + h * w * cout + 16 * cin * cout) * 4 / 1e6
        # FETCH_SIZE / WRITE_SIZE: kilobytes; gfx950 reports half of the fabric bytes (MI355X_MICROARCH.md, HBM section)
        print("%s %-8s last of %d launches: %.1f MB  (algorithmic in + bank + out %.1f MB)" % (name, lname, reps, 2 * v[-1] * 1024 / 1e6, alg))


def main():
    import torch as t
    sys.path.insert(0, ".")
    from fasterrcnn_amd.models import vgg16 as V
    dev = t.device("cuda:0")
    for name, h, w, cin, cout in LAYERS:
        x = t.randn(h, w, cin, device=dev)
        conv = t.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
        wp = V.pack_conv3x3(conv, "f32_winograd")
        b = t.zeros(cout, device=dev)
        for _ in range(3):
            V.conv3x3(x, wp, b, cin, cout, relu=True, pool=False)
        t.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
        summarize(sys.argv[2])
    else:
        main()
