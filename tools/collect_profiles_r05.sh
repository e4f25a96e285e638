#!/bin/bash
# tools/collect_profiles_r05.sh -- round 5's evidence run on the GPU box (via gpurun): tools/collect_profiles.sh (bench lines, kernel traces in
# flight / single stream, PMC, FETCH / WRITE, held-out sweeps, train steps) plus what round 5 added: the stress sweeps, the held-out / stress
# level of the all-exact-f32 table, the four- / eight-wave comparison of the one-launch kernel, the L2 counters of the ResNet-50 backbone.
set -u
TAG=${1:-r05}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh $TAG > $OUT/collect.log 2>&1; tail -5 $OUT/collect.log
for A in VGG16 ResNet50; do timeout 600 python tools/holdout_report.py --arch $A --stress --tables default,f32 --out $OUT/stress_$A.json > $OUT/stress_$A.log 2>&1; echo "stress $A exit $?"; done
timeout 600 python tools/holdout_report.py --arch VGG16 --stress --tables default --slot 1 --out $OUT/stress_VGG16_inflight.json > $OUT/stress_VGG16_inflight.log 2>&1
grep "^==" $OUT/stress_*.log | cut -c1-400
timeout 300 python tools/x3f_bench.py > $OUT/x3f_bench.txt 2>&1
FRCNN_LIB_PATH=build/libfrcnn_xdclk.so timeout 300 python tools/xd_clocks.py > $OUT/xd_clocks.txt 2>&1
timeout 300 python tools/exp_alone_tables.py > $OUT/alone_tables.txt 2>&1; tail -3 $OUT/alone_tables.txt
FRCNN_LIB_PATH=build/libfrcnn_detclk.so timeout 200 python tools/exp_det_clocks.py 2>&1 | grep "det cls" | sort -t' ' -k5 -n | tail -3 > $OUT/det_clocks.txt; cat $OUT/det_clocks.txt
R="python bench.py --backbone resnet50 --no-cpu-baseline --no-secondary --no-extra-legs --map-images 0 --roofline-images 1 --steps 8 --warmup 2 --ramp-seconds 0 --inflight 1 --min-timed-seconds 0"
timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $OUT/pmc_r50_tcc -o p -- $R > $OUT/pmc_r50_tcc.log 2>&1; echo "pmc tcc exit $?"
timeout 400 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/pmc_r50_tcp -o p -- $R > $OUT/pmc_r50_tcp.log 2>&1; echo "pmc tcp exit $?"
python - <<PY
import csv, glob, collections
for d in ("pmc_r50_tcc", "pmc_r50_tcp"):
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % d):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float); nd = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
            if "Start_Timestamp" in r and "End_Timestamp" in r:
                dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); nd[k] += 1
        with open("$OUT/%s_by_kernel.csv" % d, "w") as o:
            o.write("kernel,counter,dispatches,sum,mean_per_dispatch,mean_dispatch_us\n")
            for k in sorted(acc):
                for c in sorted(acc[k]):
                    o.write("%s,%s,%d,%.0f,%.1f,%.2f\n" % (k, c, n[(k, c)], acc[k][c], acc[k][c] / n[(k, c)], dur[k] / max(nd[k], 1) / 1e3))
PY
rm -f $OUT/pmc_*/*counter_collection.csv $OUT/pmc_*/*kernel_trace.csv $OUT/*/*.db $OUT/*/*/*.db
du -sh $OUT; ls $OUT
