#!/bin/bash
set -u
OUT=gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for S in none conv4; do for N in 1 3; do
  rm -rf $OUT/tr_${S}_$N; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_${S}_$N -o t -- python tools/exp_pair_trace.py $S $N > $OUT/tr_${S}_$N.log 2>&1
  echo "== $S $N: $(grep images $OUT/tr_${S}_$N.log)"; python tools/kstats.py $OUT/tr_${S}_$N 12
  f=$(find $OUT/tr_${S}_$N -name "*kernel_stats.csv" | head -1); cp $f $OUT/pair_${S}_${N}_kernel_stats.csv; rm -rf $OUT/tr_${S}_$N
done; done
