set -u
OUT=gpurun_out/r05c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gemm_x3t_gpu.py -q -m gpu -k "eight_wave or one_launch or chain" 2>&1 | tail -15 | cut -c1-300
timeout 600 python tools/x3f_bench.py > $OUT/x3f_bench.txt 2>&1; cat $OUT/x3f_bench.txt | cut -c1-330
FRCNN_LIB_PATH=build/libfrcnn_xdclk.so timeout 600 python tools/xd_clocks.py > $OUT/xd_clocks.txt 2>&1; cat $OUT/xd_clocks.txt | cut -c1-420
