set -u
OUT=gpurun_out/r05a
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 ./build/mfma_fill2 > $OUT/mfma_fill2.txt 2>&1; echo "micro exit $?"
cat $OUT/mfma_fill2.txt
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TCC|TCP|TCA|SQ)_[A-Z0-9_a-z\[\]]+" | sort -u > $OUT/counters_avail.txt; wc -l $OUT/counters_avail.txt
R="python bench.py --backbone resnet50 --no-cpu-baseline --no-secondary --no-extra-legs --map-images 0 --roofline-images 1 --steps 8 --warmup 2 --ramp-seconds 0 --inflight 1 --min-timed-seconds 0"
timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $OUT/pmc_r50_tcc -o p -- $R > $OUT/pmc_r50_tcc.log 2>&1; echo "pmc tcc exit $?"
timeout 400 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/pmc_r50_tcp -o p -- $R > $OUT/pmc_r50_tcp.log 2>&1; echo "pmc tcp exit $?"
tail -3 $OUT/pmc_r50_tcc.log $OUT/pmc_r50_tcp.log
ls -la $OUT/pmc_r50_tcc $OUT/pmc_r50_tcp 2>/dev/null | head
python - <<'PY'
import csv, glob, collections
for d in ("pmc_r50_tcc", "pmc_r50_tcp"):
    for f in glob.glob("gpurun_out/r05a/%s/*counter_collection.csv" % d):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        with open("gpurun_out/r05a/%s_by_kernel.csv" % d, "w") as o:
            o.write("kernel,counter,dispatches,sum,mean_per_dispatch\n")
            for k in sorted(acc):
                for c in sorted(acc[k]):
                    o.write("%s,%s,%d,%.0f,%.1f\n" % (k, c, n[(k, c)], acc[k][c], acc[k][c] / n[(k, c)]))
        print(open("gpurun_out/r05a/%s_by_kernel.csv" % d).read()[:3000])
PY
rm -f $OUT/pmc_*/*counter_collection.csv $OUT/pmc_*/*kernel_trace.csv $OUT/*/*.db $OUT/*/*/*.db
du -sh $OUT
