#!/bin/bash
set -u
OUT=gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FRCNN_LIB_PATH=build/libfrcnn_exp.so timeout 900 python -m pytest tests/test_gemm_x3t_gpu.py -m gpu -x -q > $OUT/pytest_x3_exp_fold.log 2>&1; tail -3 $OUT/pytest_x3_exp_fold.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_holdout_gpu.py -m gpu -x -q > $OUT/pytest_model_fold.log 2>&1; tail -3 $OUT/pytest_model_fold.log
FRCNN_LIB_PATH=build/libfrcnn_xdclk.so timeout 600 python tools/xd_clocks.py four > $OUT/xd_clocks_fold.txt 2>&1; grep -A1 "cycles / chunk" $OUT/xd_clocks_fold.txt | cut -c1-20,150-330
