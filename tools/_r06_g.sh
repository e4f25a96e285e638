#!/bin/bash
set -u
OUT=gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python tools/exp_pair.py > $OUT/exp_pair_v1.txt 2>&1; cat $OUT/exp_pair_v1.txt | grep -v amdgpu.ids
