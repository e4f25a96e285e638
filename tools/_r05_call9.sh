cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_harness_gpu.py -q -m gpu 2>&1 | tail -5 | cut -c1-250
FRCNN_LIB_PATH=build/libfrcnn_detclk.so timeout 200 python tools/exp_det_clocks.py 2>&1 | grep "det cls" | sort -t' ' -k5 -n | tail -4
timeout 200 python tools/exp_alone_tables.py 2>&1 | tail -3
