"""tools/cu_time_model.py -- a CU-time budget of one image from a SINGLE-STREAM rocprofv3 kernel trace (one image at a time: a dispatch's
duration is its own).  For every dispatch: blocks resident per CU from its registers / LDS / block size, the fraction of the chip's block slots
its grid fills over its rounds of resident blocks, and  cu_time = duration x (rounds / ceil(rounds)), rounds = blocks / (256 x blocks per CU).  The sum over an image's dispatches is what the chip must
spend on the image however many images are in flight (kernels that leave CUs empty can overlap others'; kernels that fill it cannot).
    (1) rocprofv3 --kernel-trace -d DIR -- python tools/cu_time_model.py run vgg16|resnet50 [images]
    (2) python tools/cu_time_model.py report DIR
    (3) python tools/cu_time_model.py report DIR N      (every dispatch of the trace over N units of work: e.g. train steps)"""
import csv, glob, sys
from collections import defaultdict


# blocks per CU of the kernels whose limit is DYNAMIC LDS (csrc: XD_LDS_BYTES 145,664; HxCfg<5,2,2,4,NSUB>::LDS_BYTES 110,592 / 147,456;
# HxCfg<5,1,1,4,NSUB> 55,296 / 73,728; roi_pool_x3t_rows_kernel 66,048; conv_gather_x3 __launch_bounds__(256, 2))
KNOWN_BLOCKS_PER_CU = {"wino_x3d_kernel": 1, "gemm_x3t_kernel<5, 2, 2, 4": 1, "gemm_x3t_kernel<5, 1, 1, 4": 2, "roi_pool_x3t_rows_kernel": 2,
                       "conv_gather_x3_kernel": 2}


def run(arch, n):
    sys.path.insert(0, ".")
    import torch
    from fasterrcnn_amd import synthetic
    from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
    dev = torch.device("cuda", 0)
    if arch == "vgg16":
        from fasterrcnn_amd.models.vgg16 import VGG16Backbone
        m = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
        m.load_state_dict(synthetic.vgg16_state_dict(1234), strict=True)
        pool = [synthetic.image(s).unsqueeze(0).to(dev) for s in range(4)]
    else:
        from fasterrcnn_amd.models import resnet as _resnet
        m = FasterRCNNModel(num_classes=21, backbone=_resnet.ResNetBackbone(_resnet.Architecture.ResNet50))
        m.load_state_dict(synthetic.resnet_state_dict(1234, "ResNet50"), strict=True)
        pool = [synthetic.image_rgb(s).unsqueeze(0).to(dev) for s in range(4)]
    m = m.cuda(dev).eval()
    for i in range(n):
        m.predict_async(pool[i % 4], 0.05, slot=1).result()      # slot 1: the in-flight slots' table
    torch.cuda.synchronize()


def report(d, units=0):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if units > 0:
        # (3) report DIR N: every dispatch of the trace over N units of work (e.g. the train steps of `rocprofv3 ... python tools/train_bench.py --steps a --warmup b`, N = a + b)
        nimg = units
    else:
        # images = detections_kernel dispatches; keep the last 60 %
        det = [i for i, r in enumerate(rows) if "detections_kernel" in r["Kernel_Name"]]
        first = det[len(det) * 4 // 10]
        rows = rows[first + 1:det[-1] + 1]
        nimg = len(det) - 1 - len(det) * 4 // 10
    agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0])
    for r in rows:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
        grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        blocks = max(1, grid // max(1, wg))
        waves = (wg + 63) // 64
        def col(*names):
            for nm in names:
                if r.get(nm) not in (None, ""):
                    return int(r[nm])
            return 0
        regs = col("Arch_VGPR_Count", "VGPR_Count", "Arch_Vgpr_Count") + col("Accum_VGPR_Count", "Accum_Vgpr_Count")
        regs = max(8, (regs + 7) // 8 * 8)
        lds = col("LDS_Block_Size", "Group_Segment_Size", "Lds_Block_Size")
        waves_per_simd = min(8, 512 // regs)
        by_regs = max(1, (waves_per_simd * 4) // waves) if waves_per_simd * 4 >= waves else 1
        by_lds = max(1, 163840 // lds) if lds > 0 else 32
        by_waves = max(1, 32 // waves)
        bpc = min(by_regs, by_lds, by_waves)
        for key, val in KNOWN_BLOCKS_PER_CU.items():               # (the trace's LDS column does not include dynamic LDS)
            if key in r["Kernel_Name"]:
                bpc = val
        rounds = blocks / (256.0 * bpc)
        frac = rounds / max(1.0, float(-(-blocks // (256 * bpc))))          # mean fill over the launch's rounds of resident blocks (the last round is partial)
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("frcnn::", "")[:48]
        a = agg[name]
        a[0] += 1; a[1] += dur; a[2] += dur * frac; a[3] += frac; a[4] = bpc
    tot_d = sum(a[1] for a in agg.values()); tot_c = sum(a[2] for a in agg.values())
    print("%d images | per image: %.1f launches, kernel time %.0f us, CU-time budget %.0f us (= %.0f images/sec if the chip were never idle)" % (
        nimg, sum(a[0] for a in agg.values()) / nimg, tot_d / nimg, tot_c / nimg, 1e6 / (tot_c / nimg)))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        print("  %-48s %5.1f launches/img  %7.1f us/img  mean chip fill %.2f (blocks/CU %2d)  CU-time %7.1f us/img  %4.1f %%" % (
            name, a[0] / nimg, a[1] / nimg, a[3] / a[0], a[4], a[2] / nimg, 100 * a[2] / tot_c))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 40)
    else:
        report(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0)
