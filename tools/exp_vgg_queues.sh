# tools/exp_vgg_queues.sh -- VGG-16 headline: images in flight x HIP hardware queues (GPU_MAX_HW_QUEUES), steady state (200 images) and bursts of 20
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { python bench.py --no-extra-legs --no-cpu-baseline --no-secondary --map-images 0 --roofline-images 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
for q in 2 4 8 16 24; do
for n in 3 4 5 6 8; do echo "queues $q inflight $n steady: $(GPU_MAX_HW_QUEUES=$q run --inflight $n)  bursts of 20: $(GPU_MAX_HW_QUEUES=$q run --inflight $n --steps 20 --warmup 5)"; done
done
