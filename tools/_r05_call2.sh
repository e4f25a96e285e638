set -u
OUT=gpurun_out/r05b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
FRCNN_RECORD_OBSERVED=1 timeout 900 python -m pytest tests/test_model_gpu.py tests/test_resnet_gpu.py -q -m gpu 2>&1 | tail -15 > $OUT/record.log; tail -5 $OUT/record.log
cat gpurun_out/observed_counts.json
timeout 1200 python -m pytest tests/test_stress_gpu.py tests/test_holdout_gpu.py tests/test_conv_x3g_gpu.py -q -m gpu -s 2>&1 > $OUT/stress_holdout.log; tail -30 $OUT/stress_holdout.log | cut -c1-400
grep -E "HELD-OUT|vs float64|gates missed|saturated operands [1-9]" $OUT/stress_holdout.log | cut -c1-330
