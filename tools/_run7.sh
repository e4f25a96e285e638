cd $GRAFT_REPO_ROOT
b() { python bench.py --no-cpu-baseline --no-secondary --steps 600 --roofline-images 1 --map-images 1 "$@" 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'])"; }
echo "default:           $(b)"
echo "inflight 16:       $(b --inflight 16)"
echo "inflight 32:       $(b --inflight 32)"
echo "BM128:             $(FRCNN_WINO_BM64_BELOW=0 b)"
echo "BM128 conv5 BM64:  $(FRCNN_WINO_BM64_BELOW=1000 b)"
echo "BM128 nsets2:      $(FRCNN_WINO_BM64_BELOW=0 FRCNN_WINO_NSETS=2 b)"
echo "blocks target 640: $(FRCNN_CONV_BLOCKS_TARGET=640 b)"
echo "blocks target 160: $(FRCNN_CONV_BLOCKS_TARGET=160 b)"
echo "default again:     $(b)"
