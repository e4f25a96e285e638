#!/bin/bash
# tools/collect_profiles_r06.sh -- round 6's evidence run on the GPU box (via gpurun): tools/collect_profiles.sh (bench lines with the parity block,
# kernel traces in flight / single stream, PMC incl. the LDS bank-conflict counters, FETCH / WRITE, held-out sweeps, train steps, summary.md from
# THESE traces) plus the stress sweeps, the full GPU test suite and smoke on the final tree, the clock tables of the one-launch kernel (four waves;
# the two-pass form from the EXPERIMENTS library), the layer bench and the batch-8 ResNet-50 kernel stats.
set -u
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_final.log 2>&1; tail -2 $OUT/pytest_gpu_final.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_final.log 2>&1; tail -1 $OUT/smoke_final.log
bash tools/collect_profiles.sh $TAG > $OUT/collect.log 2>&1; tail -5 $OUT/collect.log
for A in VGG16 ResNet50; do timeout 600 python tools/holdout_report.py --arch $A --stress --tables default,f32 --out $OUT/stress_$A.json > $OUT/stress_$A.log 2>&1; echo "stress $A exit $?"; done
timeout 600 python tools/holdout_report.py --arch VGG16 --stress --tables default --slot 1 --out $OUT/stress_VGG16_inflight.json > $OUT/stress_VGG16_inflight.log 2>&1
grep "^==" $OUT/stress_*.log | cut -c1-400
timeout 300 python tools/x3f_bench.py > $OUT/x3f_bench.txt 2>&1
FRCNN_LIB_PATH=build/libfrcnn_exp.so timeout 300 python tools/x3f_bench.py > $OUT/x3f_bench_exp.txt 2>&1
FRCNN_LIB_PATH=build/libfrcnn_xdclk.so timeout 300 python tools/xd_clocks.py four > $OUT/xd_clocks.txt 2>&1
[ -f build/libfrcnn_xpclk.so ] && FRCNN_LIB_PATH=build/libfrcnn_xpclk.so timeout 300 python tools/xd_clocks.py pair > $OUT/xp_clocks.txt 2>&1   # (the two-pass form's clock build: tools/build_experiments.sh -DXD_CLOCKS, renamed)
FRCNN_LIB_PATH=build/libfrcnn_chk.so timeout 300 python tools/xd_clocks.py four chunks > $OUT/xd_chunk_clocks.txt 2>&1
# the CU-time budget of one image (tools/cu_time_model.py: single-stream kernel trace -> duration x chip fill per dispatch)
for a in vgg16 resnet50; do rm -rf /tmp/tr_$a; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$a -o t -- python tools/cu_time_model.py run $a 40 > /dev/null 2>&1; python tools/cu_time_model.py report /tmp/tr_$a > $OUT/cu_time_$a.txt 2>&1; head -3 $OUT/cu_time_$a.txt | cut -c1-200; done
timeout 300 python tools/exp_r50_graphs.py resnet50 8 > $OUT/exp_r50_graphs.txt 2>&1; timeout 300 python tools/exp_r50_threads.py > $OUT/exp_r50_threads.txt 2>&1
timeout 600 python tools/exp_r50_presplit.py rev > $OUT/exp_r50_presplit.txt 2>&1; timeout 600 python tools/exp_r50_presplit.py all rev > $OUT/exp_r50_presplit_g3all.txt 2>&1
FRCNN_LIB_PATH=build/libfrcnn_exp.so timeout 600 python -m pytest tests/test_gemm_x3t_gpu.py -m gpu -q > $OUT/pytest_x3_exp.log 2>&1; tail -1 $OUT/pytest_x3_exp.log
rm -rf $OUT/tr_r50b8; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_r50b8 -o t -- python tools/r50_batch8_trace.py 4 > $OUT/tr_r50b8.log 2>&1
f=$(find $OUT/tr_r50b8 -name "*kernel_stats.csv" | head -1); cp $f $OUT/resnet50_batch8_kernel_stats.csv; rm -rf $OUT/tr_r50b8; grep images $OUT/tr_r50b8.log
du -sh $OUT; ls $OUT | head -80
