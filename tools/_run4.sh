cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/gpu_suite.log
timeout 400 python bench.py > gpurun_out/bench_default.log 2>&1
tail -15 gpurun_out/gpu_suite.log; tail -1 gpurun_out/bench_default.log
