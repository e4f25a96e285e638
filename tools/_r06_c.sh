#!/bin/bash
set -u
OUT=gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
./build/split_fill > $OUT/micro_split_fill.txt 2>&1; tail -6 $OUT/micro_split_fill.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err; echo "driver-form exit $?"
tail -1 $OUT/bench_driver_form.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('single_stream_images_per_sec'), d['roofline']['frac'], d.get('resnet50_images_per_sec'), d.get('resnet50_batch8_images_per_sec'))"
