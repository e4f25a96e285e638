cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gemm_x3t_gpu.py -m gpu -q -x 2>&1 | tail -2
FRCNN_LIB_PATH=build/libfrcnn_exp.so timeout 600 python -m pytest tests/test_gemm_x3t_gpu.py -m gpu -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x 2>&1 | tail -2
FRCNN_LIB_PATH=build/libfrcnn_chk.so python tools/xd_clocks.py four chunks 2>&1 | grep "cycles: chunk"
bash tools/run_ab_x3f.sh ab4 early1:build/libfrcnn_early1.so:build/libfrcnn_early1clk.so early2:fasterrcnn_amd/csrc/libfrcnn_hip.so:build/libfrcnn_xdclk.so
