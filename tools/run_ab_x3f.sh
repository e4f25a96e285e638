#!/bin/bash
# tools/run_ab_x3f.sh TAG "name:lib:clklib" ... -- same-box A/B of builds of the one-launch f32x3 Winograd kernel: whole layers (tools/x3f_bench.py),
# in-kernel clocks (tools/xd_clocks.py four) and the headline in the driver's form, per library.  Runs on the GPU box (gpurun).
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for spec in "$@"; do
    name=${spec%%:*}; rest=${spec#*:}; lib=${rest%%:*}; clk=${rest#*:}
    FRCNN_LIB_PATH=$lib timeout 300 python tools/x3f_bench.py 2>&1 | grep -v amdgpu.ids | sed -e 's/.*one-launch, channel maxima given: four//' -e 's/ (.*//' | tr '\n' ' ' > $OUT/x3f_${name}_$rep.txt
    echo "x3f us $name #$rep: $(cat $OUT/x3f_${name}_$rep.txt)"
done; done
for spec in "$@"; do
    name=${spec%%:*}; rest=${spec#*:}; lib=${rest%%:*}; clk=${rest#*:}
    [ -n "$clk" ] && FRCNN_LIB_PATH=$clk timeout 300 python tools/xd_clocks.py four > $OUT/clk_$name.txt 2>&1
    [ -n "$clk" ] && echo "== $name" && grep -v "amdgpu.ids" $OUT/clk_$name.txt | sed -e 's/launch (with the channel-maximum pass) [0-9.]* us, //' -e 's/; 1536 = the MFMAs alone//' -e 's/| block starts.*//' | cut -c1-250
done
for rep in 1 2; do
for spec in "$@"; do
    name=${spec%%:*}; rest=${spec#*:}; lib=${rest%%:*}
    FRCNN_LIB_PATH=$lib timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-extra-legs --no-cpu-baseline > $OUT/bench_${name}_$rep.json 2> $OUT/bench_${name}_$rep.err
    python -c "
import json;d=json.loads(open('$OUT/bench_${name}_$rep.json').read().strip().splitlines()[-1]);print('bench $name #$rep', d['value'],d['ms_per_step'],d['roofline']['frac'],d.get('parity',{}).get('golden_600x1000',{}).get('forward_rows_within_gate'))"
done; done
