# tools/exp_burst_graphs.sh -- bursts of 20 images (the driver's form), 4 in flight: eager launches against one hipGraph replay per image
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { python bench.py --no-extra-legs --no-cpu-baseline --no-secondary --map-images 0 --roofline-images 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['timed_bursts']['min_ms'], d['parity']['golden_600x1000']['forward_rows_within_gate'], d['parity']['golden_600x1000']['predict_rows_within_gate'], d.get('host_cpu_per_image'))"; }
for rep in 1 2 3; do
  echo "eager  20/5: $(run --steps 20 --warmup 5)"
  echo "graphs 20/5: $(run --steps 20 --warmup 5 --hip-graphs)"
done
echo "eager  steady: $(run)"
echo "graphs steady: $(run --hip-graphs)"
