#!/bin/bash
set -u
OUT=gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>&1 | head -30 > $OUT/hwmon_ls.txt
rocm-smi --showpower --showclocks --json > $OUT/smi.json 2>&1
for rep in 1 2; do
for L in conv4_2 conv2_2 conv1_2; do
  for lib in old exp; do FRCNN_LIB_PATH=build/libfrcnn_$lib.so timeout 120 python tools/power_probe.py $L 3 2>&1 | grep -v amdgpu.ids; done
done
done > $OUT/power_probe.txt 2>&1
cat $OUT/power_probe.txt
for v in oldclk xdclk xdclk_a16 xdclk_a3 xdclk_a1 xdclk_a2 xdclk_a4 xdclk_a8 oldclk xdclk; do
  echo "== $v"; FRCNN_LIB_PATH=build/libfrcnn_$v.so timeout 300 python tools/xd_clocks.py four 2>&1 | grep "cycles / chunk" | cut -c1-330
done > $OUT/xd_clocks_ablate.txt 2>&1
cat $OUT/xd_clocks_ablate.txt | cut -c1-20,100-330
