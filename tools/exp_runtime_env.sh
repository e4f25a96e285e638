# tools/exp_runtime_env.sh -- HIP / HSA runtime knobs under the headline loop (4 images in flight), bursts of 20 and steady state, and one image at a time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { python bench.py --no-extra-legs --no-cpu-baseline --no-secondary --map-images 0 --roofline-images 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
for rep in 1 2; do
for e in "X=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "HSA_ENABLE_INTERRUPT=0" "HSA_ENABLE_INTERRUPT=0 HIP_FORCE_DEV_KERNARG=1"; do
  echo "$e : bursts of 20 $(env $e bash -c "$(declare -f run); run --steps 20 --warmup 5")  steady $(env $e bash -c "$(declare -f run); run")  one at a time $(env $e bash -c "$(declare -f run); run --inflight 1")"
done; done
