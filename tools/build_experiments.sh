#!/bin/bash
# Builds build/libfrcnn_exp.so = the library with `make EXPERIMENTS=1` kernels (csrc/wino_x3e.hip: FRCNN_X3F_WAVES8; csrc/wino_x3p.hip: FRCNN_X3F_PAIR); use with
# FRCNN_LIB_PATH=build/libfrcnn_exp.so.  The regular objects are reused; only wino_x3f / wino_x3e are compiled with -DFRCNN_EXPERIMENTS.
set -e
cd "$(dirname "$0")/.."
mkdir -p build
make -C fasterrcnn_amd/csrc >/dev/null
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-result -DFRCNN_EXPERIMENTS"
/opt/rocm/bin/hipcc $F "$@" -c fasterrcnn_amd/csrc/wino_x3f.hip -o build/wino_x3f_exp.o &
/opt/rocm/bin/hipcc $F "$@" -c fasterrcnn_amd/csrc/wino_x3e.hip -o build/wino_x3e_exp.o &
/opt/rocm/bin/hipcc $F "$@" -c fasterrcnn_amd/csrc/wino_x3p.hip -o build/wino_x3p_exp.o &
wait
objs=$(ls fasterrcnn_amd/csrc/*.o | grep -v "/wino_x3f.o" | grep -v "/wino_x3e.o" | grep -v "/wino_x3p.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/wino_x3f_exp.o build/wino_x3e_exp.o build/wino_x3p_exp.o -o build/libfrcnn_exp.so
echo build/libfrcnn_exp.so
