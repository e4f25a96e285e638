#!/bin/bash
set -u
OUT=gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 20 --warmup 5 --no-extra-legs > $OUT/bench_parity.json 2> $OUT/bench_parity.err; echo "bench exit $?"; tail -3 $OUT/bench_parity.err
tail -1 $OUT/bench_parity.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(json.dumps(d['parity'])); print(d['host_cpu_per_image']); print([(r['layer'], r['us'], r['frac']) for r in d['roofline']['per_layer']])"
timeout 600 python -m pytest tests/test_stress_gpu.py -m gpu -x -q 2>&1 | tail -3
