set -u
OUT=gpurun_out/r06b; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gemm_x3t_gpu.py -m gpu -q -x > $OUT/pytest_x3.log 2>&1; tail -3 $OUT/pytest_x3.log
FRCNN_LIB_PATH=build/libfrcnn_exp.so timeout 600 python -m pytest tests/test_gemm_x3t_gpu.py -m gpu -q > $OUT/pytest_x3_exp.log 2>&1; tail -3 $OUT/pytest_x3_exp.log
FRCNN_LIB_PATH=build/libfrcnn_xdclk.so timeout 300 python tools/xd_clocks.py four > $OUT/xd_clocks.txt 2>&1; cut -c1-330 $OUT/xd_clocks.txt
timeout 300 python tools/x3f_bench.py > $OUT/x3f_bench.txt 2>&1; cut -c100-300 $OUT/x3f_bench.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-extra-legs --no-cpu-baseline > $OUT/bench_quick.json 2> $OUT/bench_quick.err; python -c "
import json;d=json.loads(open('$OUT/bench_quick.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['frac'],d.get('parity',{}).get('golden_600x1000'))"
