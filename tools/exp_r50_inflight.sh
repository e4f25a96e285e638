# tools/exp_r50_inflight.sh -- ResNet-50 (configs[2]): batch-1 images in flight, bursts of 20 (the driver's form) and steady state, HIP hardware queues 4 / 16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { python bench.py --backbone resnet50 --no-extra-legs --no-cpu-baseline --no-secondary --map-images 0 --roofline-images 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['timed_bursts']['min_ms'])"; }
for q in 16 4 8; do
for n in 4 5 6 8 10; do echo "queues $q inflight $n, bursts of 20: $(GPU_MAX_HW_QUEUES=$q run --steps 20 --warmup 5 --inflight $n)   steady: $(GPU_MAX_HW_QUEUES=$q run --inflight $n)"; done
done
