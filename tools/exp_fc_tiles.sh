run() { python bench.py --no-extra-legs --no-cpu-baseline --no-secondary --map-images 0 --roofline-images 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
echo "fc tiles 0: $(run)"
echo "fc tiles 2: $(FRCNN_FC_TILES=2 run)"
echo "fc tiles 0: $(run)"
echo "fc tiles 2: $(FRCNN_FC_TILES=2 run)"
