set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_winograd_gpu.py -x -q -m gpu -s 2>&1 | tail -40 > gpurun_out/wino_tests.log
timeout 200 python tools/layer_bench.py --winograd --only conv > gpurun_out/wino_layers.log 2>&1
timeout 200 python tools/layer_bench.py --only conv > gpurun_out/direct_layers.log 2>&1
timeout 300 python bench.py --math f32_winograd --no-cpu-baseline --no-secondary --steps 300 > gpurun_out/bench_wino.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 300 > gpurun_out/bench_direct.log 2>&1
tail -5 gpurun_out/wino_tests.log; cat gpurun_out/wino_layers.log; tail -1 gpurun_out/bench_wino.log | cut -c1-400; tail -1 gpurun_out/bench_direct.log | cut -c1-300
