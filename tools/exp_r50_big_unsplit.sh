# tools/exp_r50_big_unsplit.sh -- ResNet-50 in flight: gemm_x3t on the 320 x 256 tiles wherever the cost model picks unsplit 160 x 128 tiles (FRCNN_HX_BIG_UNSPLIT, make KNOBS=1: build/libfrcnn_knobs.so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { python bench.py --backbone resnet50 --no-extra-legs --no-cpu-baseline --no-secondary --map-images 0 --roofline-images 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
export FRCNN_LIB_PATH=build/libfrcnn_knobs.so
for i in 1 2 3; do echo "auto: bursts $(run --steps 20 --warmup 5) steady $(run)"; echo "big unsplit: bursts $(FRCNN_HX_BIG_UNSPLIT=1 run --steps 20 --warmup 5) steady $(FRCNN_HX_BIG_UNSPLIT=1 run)"; done
