cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05e
timeout 900 python -m pytest -q -m gpu --tb=short tests/test_harness_gpu.py tests/test_model_gpu.py tests/test_winograd_gpu.py tests/test_wino_x6_gpu.py "tests/test_holdout_gpu.py::test_holdout_sweep" tests/test_stress_gpu.py tests/test_gemm_x3t_gpu.py 2>&1 | grep -v "^tests/golden\|^VGG16 \|^heavy\|^edges\|^outlier\|^ResNet" > gpurun_out/r05e/fail.log; grep -n "^E \|passed\|failed\|^FAILED" gpurun_out/r05e/fail.log | cut -c1-300 | head -40
