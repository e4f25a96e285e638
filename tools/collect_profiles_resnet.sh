#!/bin/bash
# tools/collect_profiles_resnet.sh <tag> -- the ResNet / train-step part of tools/collect_profiles.sh (after a change that leaves the VGG-16
# kernels alone): default bench line, ResNet-50 kernel stats (one image at a time, 8 in flight), its FETCH / WRITE passes, the held-out
# sweeps of the ResNets, the bf16 ResNet-101 train step.  Runs on the GPU box (via gpurun); writes gpurun_out/<tag>/...
set -u
TAG=${1:-r04p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?"
R="python bench.py --backbone resnet50 --no-cpu-baseline --no-secondary --no-extra-legs --map-images 0 --roofline-images 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_r50_single -o t -- $R --steps 30 --warmup 5 --inflight 1 > $OUT/trace_r50_single.log 2>&1; echo "r50 single trace exit $?"
rm -f $OUT/trace_r50_single/t_kernel_trace.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_r50 -o t -- $R --steps 60 --warmup 10 --inflight 8 > $OUT/trace_r50.log 2>&1; echo "r50 in-flight trace exit $?"
rm -f $OUT/trace_r50/t_kernel_trace.csv
RP="$R --steps 8 --warmup 2 --ramp-seconds 0 --inflight 1 --min-timed-seconds 0"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_r50_fetch -o p -- $RP > $OUT/pmc_r50_fetch.log 2>&1; echo "pmc r50 fetch exit $?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_r50_write -o p -- $RP > $OUT/pmc_r50_write.log 2>&1; echo "pmc r50 write exit $?"
for A in ResNet50 ResNet101; do timeout 600 python tools/holdout_report.py --arch $A --tables default --out $OUT/holdout_$A.json > $OUT/holdout_$A.log 2>&1; echo "holdout $A exit $?"; done
timeout 600 python tools/holdout_report.py --arch ResNet101 --tables g3_backbone --out $OUT/holdout_ResNet101_g3.json > $OUT/holdout_ResNet101_g3.log 2>&1
grep "^==" $OUT/holdout_*.log
timeout 600 python tools/train_bench.py --backbone resnet101 --grad-math bf16 --roi align --steps 20 --warmup 3 > $OUT/train_bench_r101_bf16.json 2>> $OUT/train_bench.err; cat $OUT/train_bench_r101_bf16.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train_r101 -o t -- python tools/train_bench.py --backbone resnet101 --grad-math bf16 --roi align --steps 8 --warmup 2 > $OUT/trace_train_r101.log 2>&1; echo "r101 train trace exit $?"
rm -f $OUT/trace_train_r101/t_kernel_trace.csv
python tools/summarize_profiles.py $OUT > $OUT/summary.md 2> $OUT/summary.err; echo "summary exit $?"
rm -f $OUT/pmc_*/*counter_collection.csv $OUT/pmc_*/*kernel_trace.csv $OUT/*/*.db $OUT/*/*/*.db
du -sh $OUT; ls $OUT
