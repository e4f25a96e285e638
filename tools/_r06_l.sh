#!/bin/bash
set -u
OUT=gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for N in 2 3 4 6; do
  timeout 300 python bench.py --inflight $N --steps 200 --warmup 20 --no-extra-legs --no-secondary --no-cpu-baseline --map-images 0 --roofline-images 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight $N', d['value'], d['ms_per_step'], (d.get('rocm_smi_under_load') or {}))"
done; done 2>&1 | tee $OUT/inflight_sweep.txt
