#!/bin/bash
# tools/collect_profiles.sh <tag> -- runs on the GPU box (via gpurun); writes gpurun_out/<tag>/...
# 1. the default bench line; 2. rocprofv3 kernel trace + stats of a bench run; 3. PMC passes
# (counters in their own runs, kernel-trace only, as the pool requires).
set -u
TAG=${1:-r05}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-extra-legs"
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?"
tail -c 2500 $OUT/bench_default.json
# the driver's form of the same command (BENCH_rNN.json): --steps 20 --warmup 5
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err; echo "driver-form bench exit $?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B > $OUT/trace.log 2>&1; echo "trace exit $?"
P="python bench.py --steps 12 --warmup 4 --ramp-seconds 0 --no-cpu-baseline --no-secondary --no-extra-legs --inflight 1 --roofline-images 2 --map-images 0 --min-timed-seconds 0"
# single-stream kernel durations (what bench.py's roofline block times with HIP events): kernel trace + stats, no counters
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_single -o t -- python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary --inflight 1 --roofline-images 10 --map-images 0 --no-extra-legs > $OUT/trace_single.log 2>&1; echo "single-stream trace exit $?"
rm -f $OUT/trace_single/t_kernel_trace.csv
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc_mfma -o p -- $P > $OUT/pmc_mfma.log 2>&1; echo "pmc mfma exit $?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $P > $OUT/pmc_fetch.log 2>&1; echo "pmc fetch exit $?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $P > $OUT/pmc_write.log 2>&1; echo "pmc write exit $?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL --output-format csv -d $OUT/pmc_lds -o p -- $P > $OUT/pmc_lds.log 2>&1; echo "pmc lds exit $?"
# ResNet-50 (BASELINE configs[2]): kernel stats one image at a time and 8 in flight; PMC (MFMA busy, FETCH / WRITE) of its kernels
R="python bench.py --backbone resnet50 --no-cpu-baseline --no-secondary --no-extra-legs --map-images 0 --roofline-images 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_r50_single -o t -- $R --steps 30 --warmup 5 --inflight 1 > $OUT/trace_r50_single.log 2>&1; echo "r50 single trace exit $?"
rm -f $OUT/trace_r50_single/t_kernel_trace.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_r50 -o t -- $R --steps 60 --warmup 10 --inflight 8 > $OUT/trace_r50.log 2>&1; echo "r50 in-flight trace exit $?"
rm -f $OUT/trace_r50/t_kernel_trace.csv
RP="$R --steps 8 --warmup 2 --ramp-seconds 0 --inflight 1 --min-timed-seconds 0"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_r50_mfma -o p -- $RP > $OUT/pmc_r50_mfma.log 2>&1; echo "pmc r50 mfma exit $?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_r50_fetch -o p -- $RP > $OUT/pmc_r50_fetch.log 2>&1; echo "pmc r50 fetch exit $?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_r50_write -o p -- $RP > $OUT/pmc_r50_write.log 2>&1; echo "pmc r50 write exit $?"
# held-out parity sweep of the default tables (tests/test_holdout_gpu.py measures the same): per-case numbers for profiles/
for A in VGG16 ResNet50 ResNet101; do timeout 600 python tools/holdout_report.py --arch $A --tables default --out $OUT/holdout_$A.json > $OUT/holdout_$A.log 2>&1; echo "holdout $A exit $?"; done
# ... and of the in-flight slots' table (VGG-16: the 512-channel f32x3 layers in the one-launch form: what the headline runs on)
timeout 600 python tools/holdout_report.py --arch VGG16 --tables default --slot 1 --out $OUT/holdout_VGG16_inflight.json > $OUT/holdout_VGG16_inflight.log 2>&1; echo "holdout VGG16 in-flight exit $?"
grep "^==" $OUT/holdout_*.log
# train step: wall time per step + per-kernel stats
timeout 600 python tools/train_bench.py --steps 20 --warmup 3 > $OUT/train_bench.json 2> $OUT/train_bench.err; echo "train bench exit $?"; cat $OUT/train_bench.json
# BASELINE configs[4] on one GPU: ResNet-101, RoIAlign, grad_math bf16 (and its float32 counterpart)
timeout 600 python tools/train_bench.py --backbone resnet101 --grad-math bf16 --roi align --steps 20 --warmup 3 > $OUT/train_bench_r101_bf16.json 2>> $OUT/train_bench.err; cat $OUT/train_bench_r101_bf16.json
timeout 600 python tools/train_bench.py --backbone resnet101 --grad-math f32 --roi align --steps 20 --warmup 3 > $OUT/train_bench_r101_f32.json 2>> $OUT/train_bench.err; cat $OUT/train_bench_r101_f32.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train_r101 -o t -- python tools/train_bench.py --backbone resnet101 --grad-math bf16 --roi align --steps 8 --warmup 2 > $OUT/trace_train_r101.log 2>&1; echo "r101 train trace exit $?"
rm -f $OUT/trace_train_r101/t_kernel_trace.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -o t -- python tools/train_bench.py --steps 8 --warmup 2 > $OUT/trace_train.log 2>&1; echo "train trace exit $?"
rm -f $OUT/trace_train/t_kernel_trace.csv
ls -la $OUT $OUT/trace | head -40
# keep the merged output small: drop the raw per-dispatch traces beyond what the summary needs
python tools/summarize_profiles.py $OUT > $OUT/summary.md 2> $OUT/summary.err; echo "summary exit $?"; head -60 $OUT/summary.md
# gpurun merges at most 64 MiB back: the raw per-dispatch dumps are condensed above (summary.md, traffic.json, pmc_counters_by_kernel.csv,
# *_kernel_stats.csv) and dropped here
rm -f $OUT/pmc_*/*counter_collection.csv $OUT/pmc_*/*kernel_trace.csv $OUT/trace/*kernel_trace.csv $OUT/*/*.db $OUT/*/*/*.db
du -sh $OUT
