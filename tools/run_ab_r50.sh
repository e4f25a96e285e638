# tools/run_ab_r50.sh LIB_A LIB_B -- same-box A/B of two builds of the library on ResNet-50 (configs[2]): kernel tests on B, then `bench.py --backbone resnet50`
# alternating, bursts of 20 (the driver's form) and steady state, with the golden counts of the parity block
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A=$1; B=$2
[ -n "$SKIP_TESTS" ] || FRCNN_LIB_PATH=$B timeout 900 python -m pytest tests/test_gemm_x3t_gpu.py tests/test_kernels_gpu.py tests/test_resnet_gpu.py -m gpu -q -x 2>&1 | tail -2
run() { python bench.py --backbone resnet50 --inflight ${IF:-8} --no-extra-legs --no-cpu-baseline --no-secondary --map-images 0 --roofline-images 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], (d.get('parity') or {}).get('golden_600x1000'))"; }
for i in 1 2 3; do for l in $A $B; do echo "r50 $l bursts of 20: $(FRCNN_LIB_PATH=$l run --steps 20 --warmup 5)"; done; done
for i in 1 2; do for l in $A $B; do echo "r50 $l steady: $(FRCNN_LIB_PATH=$l run)"; done; done
