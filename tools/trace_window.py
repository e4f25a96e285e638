"""tools/trace_window.py <kernel_trace.csv> [n] -- prints n consecutive dispatches from the middle of a rocprofv3 kernel trace:
start / end (us from the first), queue, grid, LDS, kernel.  Development aid."""
import csv
import sys


def main(path, n=150):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    mid = len(rows) * 2 // 3
    win = rows[mid:mid + n]
    t0 = int(win[0]["Start_Timestamp"])
    print(list(rows[0].keys()))
    for r in win:
        print("%9.1f %9.1f q%-3s grid %-8s wg %-5s lds %-7s %s" % (
            (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")),
            r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")), r.get("LDS_Block_Size", "?"),
            r["Kernel_Name"].split("(")[0].replace("void ", "").replace("frcnn::", "")[:50]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 150)
