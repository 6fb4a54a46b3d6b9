#!/bin/bash
# tools/ab_aggregate.sh NAME [-DMACRO ...]: build build_ab/lib_NAME.so = the library with csrc/aggregate.hip compiled under the given macros
# (the other translation units from build_ab/obj/*.o: run tools/ab_objects.sh and tools/ab_variant.sh base first); use with IHGNN_HIP_LIBRARY=...
set -e
cd "$(dirname "$0")/.."
name=$1; shift
# the cached objects must be of the current ABI (include/ihgnn_hip.h newer than an object = stale: rebuild them all)
mkdir -p build_ab/obj
if [ ! -f build_ab/obj/split_base.o ] || [ -n "$(find include/ihgnn_hip.h ihgnn_amd/csrc -newer build_ab/obj/host.o \( -name '*.h' -o -name '*.hpp' -o -name '*.hip' \) | head -1)" ]; then
  bash tools/ab_objects.sh > /dev/null
  bash tools/ab_variant.sh base > /dev/null
fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wno-unused-function -I include -I ihgnn_amd/csrc "$@" -o build_ab/obj/aggregate_$name.o ihgnn_amd/csrc/aggregate.hip
hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/lib_$name.so build_ab/obj/host.o build_ab/obj/aggregate_$name.o build_ab/obj/interact.o build_ab/obj/dense.o build_ab/obj/tail.o build_ab/obj/eval.o build_ab/obj/narrow.o build_ab/obj/split_base.o build_ab/obj/splitnode_base.o
echo build_ab/lib_$name.so
