cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab3
timeout 900 python -m pytest tests/test_gemm_x3t_gpu.py -m gpu -q -x 2>&1 | tail -2
FRCNN_LIB_PATH=build/libfrcnn_exp.so timeout 600 python -m pytest tests/test_gemm_x3t_gpu.py -m gpu -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x 2>&1 | tail -2
FRCNN_LIB_PATH=build/libfrcnn_chk.so python tools/xd_clocks.py four chunks 2>&1 | grep "cycles: chunk"
bash tools/run_ab_x3f.sh ab3 base:build/libfrcnn_base.so:build/libfrcnn_baseclk.so v3:fasterrcnn_amd/csrc/libfrcnn_hip.so:build/libfrcnn_xdclk.so
