#!/bin/bash
# Builds timing-experiment variants of libfrcnn_hip.so into build/ (git-ignored, shipped by gpurun):
#   tools/build_ablate.sh NAME "-DWF_ABLATE=3 ..."   ->  build/libfrcnn_NAME.so   (use with FRCNN_LIB_PATH=build/libfrcnn_NAME.so)
# Only csrc/winofused.hip is recompiled with the extra flags; the other objects come from the regular build.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build
make -C fasterrcnn_amd/csrc >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result "$@" \
    -c fasterrcnn_amd/csrc/winofused.hip -o build/winofused_$name.o
objs=$(ls fasterrcnn_amd/csrc/*.o | grep -v winofused.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/winofused_$name.o -o build/libfrcnn_$name.so
echo build/libfrcnn_$name.so
