cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05g
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r05g/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05g/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05g/smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r05g/smoke.log
tail -3 gpurun_out/r05g/pytest_gpu.log | cut -c1-200; tail -2 gpurun_out/r05g/smoke.log
