#!/bin/bash
# Builds timing-experiment variants of libfrcnn_hip.so into build/ (git-ignored, shipped by gpurun):
#   tools/build_ablate.sh NAME "-DWF_ABLATE=3 ..."   ->  build/libfrcnn_NAME.so   (use with FRCNN_LIB_PATH=build/libfrcnn_NAME.so)
#   SRC=linear_x6 tools/build_ablate.sh NAME -DLX_ABLATE=3
# Only csrc/$SRC.hip (default winofused) is recompiled with the extra flags; the other objects come from the regular build.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=${SRC:-winofused}
mkdir -p build
make -C fasterrcnn_amd/csrc >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result "$@" \
    -c fasterrcnn_amd/csrc/$src.hip -o build/${src}_$name.o
objs=$(ls fasterrcnn_amd/csrc/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/${src}_$name.o -o build/libfrcnn_$name.so
echo build/libfrcnn_$name.so
