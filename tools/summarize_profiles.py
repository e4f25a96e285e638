"""tools/summarize_profiles.py <dir> -- condenses the rocprofv3 CSVs written by collect_profiles.sh
into a markdown summary (per-kernel time, MFMA busy, HBM bytes per launch)."""
import collections
import csv
import glob
import os
import sys


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("frcnn::", "")
    return n[:60]


def load(pattern):
    files = glob.glob(pattern)
    return list(csv.DictReader(open(files[0]))) if files else []


def main(d):
    print("# rocprofv3 summary (%s)\n" % os.path.basename(d.rstrip("/")))
    bench = os.path.join(d, "bench_default.json")
    if os.path.exists(bench):
        lines = [l for l in open(bench).read().splitlines() if l.startswith("{")]
        if lines:
            print("## bench.py (default flags)\n\n```json\n%s\n```\n" % lines[-1])
    stats = load(os.path.join(d, "trace", "*kernel_stats.csv"))
    if stats:
        print("## kernel trace + stats (bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary, default images in flight)\n")
        print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
        for r in stats[:24]:
            print("| %s | %s | %.2f | %.1f | %.2f |" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                     float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
        print()
    sstats = load(os.path.join(d, "trace_single", "*kernel_stats.csv"))
    if sstats:
        print("## single stream: kernel stats (bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary --inflight 1 "
              "--roofline-images 10 --map-images 0)\n")
        print("One image at a time, so a kernel's duration is its own: this is the regime bench.py's `roofline` block times with "
              "HIP events (`avg_launch_us` = mean over the wino_fused_kernel launches of both instantiations).\n")
        print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
        tot_ns, tot_calls = 0.0, 0
        for r in sstats[:20]:
            print("| %s | %s | %.2f | %.1f | %.2f |" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                     float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
        for fam in ("wino_x3d_kernel", "wino_fused_kernel", "gemm_x6t_kernel", "gemm_x3t_kernel", "wino_input_x6t_kernel", "wino_input_x3t_kernel", "wino_output_kernel", "linear_x6_kernel", "linear_mfma_kernel",
                    "conv3x3_mfma_kernel"):
            f_ns = sum(float(r["TotalDurationNs"]) for r in sstats if fam in r["Name"])
            f_calls = sum(int(r["Calls"]) for r in sstats if fam in r["Name"])
            if f_calls:
                print("\n%s, all instantiations: %d launches, mean %.1f us" % (fam, f_calls, f_ns / f_calls / 1e3))
        print()
    for sub, title in (("trace_r50_single", "ResNet-50 (BASELINE configs[2]), one image at a time: kernel stats (bench.py --backbone resnet50 "
                                             "--inflight 1; default modes: layer4 head + RPN trunk in f32x3)"),
                       ("trace_r50", "ResNet-50, 8 batch-1 images in flight: kernel stats (bench.py --backbone resnet50 --inflight 8)")):
        rs = load(os.path.join(d, sub, "*kernel_stats.csv"))
        if rs:
            print("## %s\n" % title)
            print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
            for r in rs[:22]:
                print("| %s | %s | %.2f | %.1f | %.2f |" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                         float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
            print()
    tb = os.path.join(d, "train_bench.json")
    if os.path.exists(tb):
        lines = [l for l in open(tb).read().splitlines() if l.startswith("{")]
        if lines:
            print("## tools/train_bench.py (train step, SURVEY section 8 row f3)\n\n```json\n%s\n```\n" % lines[-1])
    tstats = load(os.path.join(d, "trace_train", "*kernel_stats.csv"))
    if tstats:
        print("## train step: kernel stats (tools/train_bench.py --steps 8 --warmup 2 = 10 steps)\n")
        print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
        for r in tstats[:24]:
            print("| %s | %s | %.2f | %.1f | %.2f |" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                     float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
        print()
    for name, title in (("train_bench_r101_bf16.json", "tools/train_bench.py --backbone resnet101 --grad-math bf16 --roi align (BASELINE configs[4] on one GPU)"),
                        ("train_bench_r101_f32.json", "the same step in float32 (--grad-math f32)")):
        f = os.path.join(d, name)
        if os.path.exists(f):
            lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
            if lines:
                print("## %s\n\n```json\n%s\n```\n" % (title, lines[-1]))
    rstats = load(os.path.join(d, "trace_train_r101", "*kernel_stats.csv"))
    if rstats:
        print("## ResNet-101 + RoIAlign train step, grad_math bf16: kernel stats (10 steps)\n")
        print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
        for r in rstats[:20]:
            print("| %s | %s | %.2f | %.1f | %.2f |" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                     float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
        print()
    hold = sorted(glob.glob(os.path.join(d, "holdout_*.log")))
    if hold:
        print("## held-out parity sweep (tools/holdout_report.py; the asserted form: tests/test_holdout_gpu.py)\n\n```")
        for f in hold:
            for l in open(f):
                if l.startswith("=="):
                    print(l.rstrip())
        print("```\n")
    # per-layer conv durations by grid
    kt = load(os.path.join(d, "trace", "*kernel_trace.csv"))
    if kt:
        agg = collections.defaultdict(list)
        for r in kt:
            if any(x in r["Kernel_Name"] for x in ("conv3x3_mfma", "linear_mfma", "wino_fused", "linear_x6", "gemm_x6t", "gemm_x3t")):
                key = (short(r["Kernel_Name"]), r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
                agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        print("## MFMA kernels by grid (threads x, blocks y, z), multi-stream run\n")
        print("| kernel | grid | launches | avg us |\n|---|---|---|---|")
        for k, v in sorted(agg.items()):
            print("| %s | %s x %s x %s | %d | %.1f |" % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v)))
        print()
    for tag, title in (("pmc_mfma", "MFMA / wave-state counters"), ("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"),
                       ("pmc_lds", "LDS counters"), ("pmc_r50_mfma", "ResNet-50: MFMA / wave-state counters"),
                       ("pmc_r50_fetch", "ResNet-50: FETCH_SIZE"), ("pmc_r50_write", "ResNet-50: WRITE_SIZE")):
        rows = load(os.path.join(d, tag, "*counter_collection.csv"))
        if not rows:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(set)
        for r in rows:
            k = short(r["Kernel_Name"])
            if not any(x in k for x in ("conv3x3", "linear_", "wino_", "roi_", "topk", "nms_", "detections", "splitk", "conv_splitk", "split_rows",
                                        "gemm_x6t", "gemm_x3t", "conv_gather", "split_pixels", "split_patches")):
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
        print("## %s (single stream, per launch averages)\n" % title)
        names = sorted({c for v in agg.values() for c in v})
        print("| kernel | launches | " + " | ".join(names) + " |\n|---|---|" + "---|" * len(names))
        for k, v in sorted(agg.items()):
            n = max(len(cnt[k]), 1)
            print("| %s | %d | " % (k, n) + " | ".join("%.4g" % (v.get(c, 0.0) / n) for c in names) + " |")
        print()
        if tag in ("pmc_mfma", "pmc_r50_mfma"):
            print("derived (per kernel, summed over its launches): MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCD * 1024 SIMD); "
                  "clock = GRBM_GUI_ACTIVE/8 / duration\n")
            print("| kernel | MFMA busy | waves/SIMD avg (SQ_WAVE_CYCLES*4 / (1024 * GUI/8)) | WAIT_INST_ANY | WAIT_ANY | ACTIVE |\n|---|---|---|---|---|---|")
            for k, v in sorted(agg.items()):
                g = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
                wc = v.get("SQ_WAVE_CYCLES", 0.0)
                if g <= 0 or wc <= 0:
                    continue
                print("| %s | %.3f | %.2f | %.3f | %.3f | %.3f |" % (
                    k, v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (g * 1024), wc * 4 / (1024 * g),
                    v.get("SQ_WAIT_INST_ANY", 0.0) / wc, v.get("SQ_WAIT_ANY", 0.0) / wc, v.get("SQ_ACTIVE_INST_ANY", 0.0) / wc))
            print()


def traffic(d):
    """
    HBM bytes per launch from the separate FETCH_SIZE / WRITE_SIZE passes (KiB per dispatch), with the gfx950 correction of
    /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE reports half of the bytes of wide coalesced reads ->
    doubled; WRITE_SIZE taken as is.  Written to <dir>/traffic.json (bench.py reads the copy committed under profiles/rNN/):
    `hbm_bytes_per_launch` = conv3x3_mfma_kernel (all instantiations), `by_kernel` = the same figure per kernel family.
    """
    import json
    families = (("wino_x3d_kernel", lambda n: "wino_x3d_kernel" in n),
                ("wino_fused_kernel", lambda n: "wino_fused_kernel" in n),
                ("gemm_x6t_kernel", lambda n: "gemm_x6t_kernel" in n),
                ("gemm_x3t_kernel", lambda n: "gemm_x3t_kernel" in n),
                ("wino_input_x6t_kernel", lambda n: "wino_input_x6t_kernel" in n),
                ("wino_output_kernel", lambda n: "wino_output_kernel" in n),
                ("conv_gather_x3_kernel", lambda n: "conv_gather_x3_kernel" in n),
                ("conv_gather_mfma_kernel", lambda n: "conv_gather_mfma_kernel" in n),
                ("linear_x6_kernel", lambda n: "linear_x6_kernel" in n),
                ("conv3x3_mfma_kernel", lambda n: "conv3x3_mfma" in n),
                ("conv3x3_c3_kernel", lambda n: "conv3x3_c3" in n),
                ("linear_mfma_kernel", lambda n: "linear_mfma_kernel" in n and "true" not in n.split("(")[0]))
    raw = {}
    for tag, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        raw[name] = [r for r in load(os.path.join(d, tag, "*counter_collection.csv")) if r["Counter_Name"] == name]
        # the ResNet-50 passes contribute the kernels the VGG-16 run does not launch (conv_gather_mfma_kernel); Dispatch ids are made
        # unique per pass
        extra = [dict(r, Dispatch_Id="r50_" + r["Dispatch_Id"]) for r in load(os.path.join(d, tag.replace("pmc_", "pmc_r50_"), "*counter_collection.csv"))
                 if r["Counter_Name"] == name and "conv_gather" in r["Kernel_Name"]]
        raw[name] += extra
    if not raw["FETCH_SIZE"] or not raw["WRITE_SIZE"]:
        return
    by = {}
    for fam, pred in families:
        vals = {}
        for name in ("FETCH_SIZE", "WRITE_SIZE"):
            rows = [r for r in raw[name] if pred(r["Kernel_Name"])]
            disp = {r["Dispatch_Id"] for r in rows}
            vals[name] = (sum(float(r["Counter_Value"]) for r in rows) / max(len(disp), 1), len(disp))
        if vals["FETCH_SIZE"][1] == 0:
            continue
        f, w = vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0]
        by[fam] = {"launches": vals["FETCH_SIZE"][1], "FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w,
                   "hbm_bytes_per_launch": f * 1024 * 2.0 + w * 1024}
    head = "wino_x3d_kernel" if "wino_x3d_kernel" in by else "wino_fused_kernel" if "wino_fused_kernel" in by else ("conv3x3_mfma_kernel" if "conv3x3_mfma_kernel" in by else None)
    if head is None:
        if not by:
            return
        head = next(iter(by))              # a partial collection (ResNet passes only): tools/collect_profiles.sh PARTIAL=1
    c = by[head]
    rec = {"source": "%s (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, single stream)" % d,
           "kernel": "%s (all instantiations)" % head, "launches": c["launches"],
           "FETCH_SIZE_KiB_per_launch": c["FETCH_SIZE_KiB_per_launch"], "WRITE_SIZE_KiB_per_launch": c["WRITE_SIZE_KiB_per_launch"],
           "fetch_correction": 2.0, "hbm_bytes_per_launch": c["hbm_bytes_per_launch"], "by_kernel": by}
    json.dump(rec, open(os.path.join(d, "traffic.json"), "w"), indent=1)


def condense_pmc(d):
    """Per-dispatch counter dumps -> <dir>/pmc_counters_by_kernel.csv (pass, kernel, counter, dispatches, sum): what gets committed."""
    out = []
    for tag in ("mfma", "fetch", "write", "lds", "r50_mfma", "r50_fetch", "r50_write"):
        rows = load(os.path.join(d, "pmc_" + tag, "*counter_collection.csv"))
        agg = collections.defaultdict(float)
        cnt = collections.defaultdict(set)
        for r in rows:
            key = (tag, short(r["Kernel_Name"]), r["Counter_Name"])
            agg[key] += float(r["Counter_Value"])
            cnt[key].add(r["Dispatch_Id"])
        out += [(k[0], k[1], k[2], len(cnt[k]), v) for k, v in sorted(agg.items())]
    if out:
        with open(os.path.join(d, "pmc_counters_by_kernel.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["pass", "kernel", "counter", "dispatches", "sum_over_dispatches"])
            w.writerows(out)


if __name__ == "__main__":
    main(sys.argv[1])
    traffic(sys.argv[1])
    condense_pmc(sys.argv[1])
