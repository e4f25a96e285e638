#!/bin/bash
set -u
OUT=gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FRCNN_LIB_PATH=build/libfrcnn_exp.so timeout 900 python -m pytest tests/test_gemm_x3t_gpu.py -m gpu -x -q > $OUT/pytest_x3_exp.log 2>&1; tail -3 $OUT/pytest_x3_exp.log
FRCNN_LIB_PATH=build/libfrcnn_xdclk.so timeout 600 python tools/xd_clocks.py four > $OUT/xd_clocks_D.txt 2>&1; grep -c cycles $OUT/xd_clocks_D.txt
FRCNN_LIB_PATH=build/libfrcnn_exp.so timeout 600 python tools/x3f_bench.py > $OUT/x3f_bench_D.txt 2>&1; tail -12 $OUT/x3f_bench_D.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-extra-legs > $OUT/bench_D.json 2> $OUT/bench_D.err; echo "bench exit $?"
tail -1 $OUT/bench_D.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
