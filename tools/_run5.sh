cd $GRAFT_REPO_ROOT
sample() { for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 1; done; }
echo "== bench f32_winograd"
(timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 4000 > gpurun_out/b5.log 2>&1) &
sleep 14; sample; wait
tail -1 gpurun_out/b5.log | cut -c1-100
echo "== layer_bench winograd conv4_2 random data"
(timeout 100 python tools/layer_bench.py --winograd --only conv4_2 --reps 30000 > gpurun_out/l5.log 2>&1) &
sleep 12; sample; wait
grep conv4_2 gpurun_out/l5.log
