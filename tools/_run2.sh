cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d gpurun_out/prof_wino -o t --output-format csv -- python tools/layer_bench.py --winograd --only conv --reps 10 > gpurun_out/prof_wino.log 2>&1
python tools/trace_by_grid.py gpurun_out/prof_wino wino linear_mfma conv3x3 | tee gpurun_out/prof_wino_by_grid.txt
