"""tools/trace_overlap.py <kernel_trace.csv> -- how the kernels of an in-flight run share the GPU (rocprofv3 --kernel-trace CSV):
per kernel family: launches, summed duration, duration while it is the ONLY kernel on the GPU, duration-weighted mean number of
co-running kernels; and for the whole trace: wall time, time with >= 1 kernel running, mean concurrency.  Development aid."""
import collections
import csv
import sys


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("frcnn::", "")
    return n[:44]


def main(path, skip_frac=0.3):
    rows = list(csv.DictReader(open(path)))
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows]
    ev.sort()
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    lo = t0 + int((t1 - t0) * skip_frac)            # skip the warm-up part of the run
    ev = [e for e in ev if e[0] >= lo]
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    points = []
    for i, (s, e, n) in enumerate(ev):
        points.append((s, 1, i))
        points.append((e, -1, i))
    points.sort()
    active = set()
    last = points[0][0]
    busy = 0
    conc_time = collections.Counter()
    alone = collections.Counter()
    weighted = collections.Counter()
    for t, d, i in points:
        dt = t - last
        if dt > 0 and active:
            busy += dt
            conc_time[len(active)] += dt
            for j in active:
                weighted[ev[j][2]] += dt * len(active)
                if len(active) == 1:
                    alone[ev[j][2]] += dt
        last = t
        if d == 1:
            active.add(i)
        else:
            active.discard(i)
    tot = collections.Counter()
    cnt = collections.Counter()
    for s, e, n in ev:
        tot[n] += e - s
        cnt[n] += 1
    wall = t1 - t0
    print("wall %.2f ms, >= 1 kernel running %.2f ms (%.1f %%), summed kernel time %.2f ms -> mean concurrency %.2f" % (
        wall / 1e6, busy / 1e6, 100.0 * busy / wall, sum(tot.values()) / 1e6, sum(tot.values()) / busy))
    print("time by number of kernels running: " + ", ".join("%d: %.1f %%" % (k, 100.0 * v / wall) for k, v in sorted(conc_time.items())))
    print("| kernel | launches | summed ms | %% of wall | alone ms | mean co-running |")
    for n, v in tot.most_common(16):
        print("| %s | %d | %.2f | %.1f | %.2f | %.2f |" % (n, cnt[n], v / 1e6, 100.0 * v / wall, alone[n] / 1e6, weighted[n] / v))


if __name__ == "__main__":
    main(sys.argv[1])
