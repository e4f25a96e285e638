#!/bin/bash
# Builds timing-experiment variants of libfrcnn_hip.so into build/ (git-ignored, shipped by gpurun):
#   tools/build_ablate.sh NAME "-DWF_ABLATE=3 ..."   ->  build/libfrcnn_NAME.so   (use with FRCNN_LIB_PATH=build/libfrcnn_NAME.so)
#   SRC=gemm_x3t tools/build_ablate.sh NAME -DHX_ABLATE=3
#   SRC="wino_x3f wino_x3e" tools/build_ablate.sh xdclk -DXD_CLOCKS
# Only csrc/$SRC.hip (default winofused; several names allowed) are recompiled with the extra flags; the other objects come from the regular build.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
srcs=${SRC:-winofused}
mkdir -p build
make -C fasterrcnn_amd/csrc >/dev/null
objs=$(ls fasterrcnn_amd/csrc/*.o)
for src in $srcs; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-result "$@" \
        -c fasterrcnn_amd/csrc/$src.hip -o build/${src}_$name.o
    objs=$(echo "$objs" | grep -v "/$src.o")
    objs="$objs build/${src}_$name.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o build/libfrcnn_$name.so
echo build/libfrcnn_$name.so
