# tools/exp_inflight_driver.sh -- images in flight in the DRIVER'S form (bursts of 20 images: 3 slots = 7 + 7 + 6, 4 slots = 5 x 4, 5 slots = 4 x 5) and in steady state
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { python bench.py --no-extra-legs --no-cpu-baseline --no-secondary --map-images 0 --roofline-images 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['timed_bursts']['min_ms'], d['timed_bursts'].get('max_ms'))"; }
for rep in 1 2; do
  for n in 3 4 5; do echo "driver form (20 / 5) inflight $n: $(run --steps 20 --warmup 5 --inflight $n)"; done
done
for n in 3 4 5; do echo "steady (200) inflight $n: $(run --inflight $n)"; done
for n in 3 4; do echo "bursts of 40 inflight $n: $(run --steps 40 --warmup 5 --inflight $n)"; done
