#!/bin/bash
# tools/ab_variant.sh NAME [-DMACRO ...]: build build_ab/lib_NAME.so = the library with csrc/split_arith.hip and csrc/split_node.hip compiled under the given
# macros (the other translation units come from build_ab/obj/*.o, see DESIGN.md "A/B variants"); use with IHGNN_HIP_LIBRARY=...
set -e
cd "$(dirname "$0")/.."
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wno-unused-function -I include -I ihgnn_amd/csrc "$@" -o build_ab/obj/split_$name.o ihgnn_amd/csrc/split_arith.hip &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wno-unused-function -I include -I ihgnn_amd/csrc "$@" -o build_ab/obj/splitnode_$name.o ihgnn_amd/csrc/split_node.hip &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/lib_$name.so build_ab/obj/host.o build_ab/obj/aggregate.o build_ab/obj/interact.o build_ab/obj/dense.o build_ab/obj/tail.o build_ab/obj/eval.o build_ab/obj/narrow.o build_ab/obj/split_$name.o build_ab/obj/splitnode_$name.o
echo build_ab/lib_$name.so
