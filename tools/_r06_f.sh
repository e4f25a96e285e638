#!/bin/bash
set -u
OUT=gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FRCNN_LIB_PATH=build/libfrcnn_xpclk.so timeout 600 python tools/xd_clocks.py pair > $OUT/xp_clocks_v1.txt 2>&1; cut -c1-420 $OUT/xp_clocks_v1.txt
timeout 600 python tools/x3f_bench.py > $OUT/x3f_bench_pair_v1.txt 2>&1; cut -c80-400 $OUT/x3f_bench_pair_v1.txt
