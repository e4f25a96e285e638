#!/bin/bash
set -u
OUT=gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FRCNN_LIB_PATH=build/libfrcnn_exp.so timeout 900 python -m pytest tests/test_gemm_x3t_gpu.py -m gpu -x -q > $OUT/pytest_x3_exp_perm.log 2>&1; tail -3 $OUT/pytest_x3_exp_perm.log
FRCNN_LIB_PATH=build/libfrcnn_xdclk.so timeout 600 python tools/xd_clocks.py four > $OUT/xd_clocks_perm.txt 2>&1; grep "cycles / chunk" $OUT/xd_clocks_perm.txt | cut -c1-20,100-260
timeout 600 python tools/x3f_bench.py > $OUT/x3f_bench_perm.txt 2>&1; cut -c1-40,80-200 $OUT/x3f_bench_perm.txt
timeout 600 python tools/exp_pair.py none > $OUT/exp_none_perm.txt 2>&1; grep pair $OUT/exp_none_perm.txt
