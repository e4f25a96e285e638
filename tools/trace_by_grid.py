"""
tools/trace_by_grid.py -- group a rocprofv3 --kernel-trace CSV by (kernel, grid): mean / min duration per shape.

  rocprofv3 --kernel-trace -d gpurun_out/prof -o t --output-format csv -- python tools/layer_bench.py ...
  python tools/trace_by_grid.py gpurun_out/prof [substring-of-kernel-name ...]
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("frcnn::", "")[:70]


def main():
    d = sys.argv[1]
    want = sys.argv[2:]
    files = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
    if not files:
        print("no kernel_trace.csv under", d)
        return 1
    acc = defaultdict(list)
    order = []
    for f in files:
        for r in csv.DictReader(open(f)):
            k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), r["Grid_Size_Y"], r["Grid_Size_Z"])
            if want and not any(w in k[0] for w in want):
                continue
            if k not in acc:
                order.append(k)
            acc[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("%-72s %9s %5s %9s %9s" % ("kernel", "blocks.x", "n", "mean us", "min us"))
    for k in order:
        v = acc[k]
        print("%-72s %9d %5d %9.1f %9.1f" % (k[0], k[1], len(v), sum(v) / len(v), min(v)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
