run() { python bench.py --no-extra-legs --no-cpu-baseline --no-secondary --map-images 0 --roofline-images 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
echo "auto if3: $(run)"
for n in 3 4 5 6 8; do echo "cfg1 if$n: $(FRCNN_GX_CFG=1 run --inflight $n)"; done
echo "auto if3 again: $(run)"
echo "f32-only if3: $(FRCNN_X6_OFF=1 run)"
