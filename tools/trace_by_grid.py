"""tools/trace_by_grid.py -- rocprofv3 --kernel-trace CSV -> average duration per (kernel, grid size): tells the launches of one template
apart by shape.  python tools/trace_by_grid.py <dir with *_kernel_trace.csv> [substring]"""
import csv
import glob
import sys
from collections import defaultdict

d = defaultdict(list)
order = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][-60:]
        if len(sys.argv) > 2 and sys.argv[2] not in r["Kernel_Name"]:
            continue
        key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Z", ""))
        if key not in d:
            order.append(key)
        d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in order:
    v = sorted(d[k])
    print("%-62s grid %8s z %3s  n %4d  median %8.1f us  min %8.1f" % (k[0], k[1], k[2], len(v), v[len(v) // 2], v[0]))
