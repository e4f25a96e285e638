cd $GRAFT_REPO_ROOT
b() { python bench.py --no-cpu-baseline --no-secondary --steps 600 --roofline-images 4 --map-images 1 "$@" 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']['per_class_ms_per_image']; print(j['value'], 'gemm single-stream ms/img', r['winograd_gemm'])"; }
for t in 64 128 2568 256 1256 128; do echo "tile $t: $(FRCNN_WINO_TILE=$t b)"; done
