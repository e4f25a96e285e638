#!/bin/bash
set -u
OUT=gpurun_out/r05_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; grep -n "passed\|failed" $OUT/pytest_gpu.log | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err; echo "driver-form exit $?"
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default exit $?"
B="python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-extra-legs"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace -o t -- $B > $OUT/trace.log 2>&1; echo "trace exit $?"
cp $(ls /tmp/trace/*kernel_stats.csv /tmp/trace/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_single -o t -- python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary --inflight 1 --roofline-images 10 --map-images 0 --no-extra-legs > $OUT/trace_single.log 2>&1; echo "single trace exit $?"
cp $(ls /tmp/trace_single/*kernel_stats.csv /tmp/trace_single/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/single_stream_kernel_stats.csv
for f in bench_driver_form bench_default; do tail -1 $OUT/$f.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['single_stream_images_per_sec'], d['roofline']['frac'], d['cpu_baseline']['value'], d.get('resnet50_images_per_sec'))"; done
head -8 $OUT/kernel_stats.csv | cut -c1-60,160-260
