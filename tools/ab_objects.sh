#!/bin/bash
# tools/ab_objects.sh: compile every translation unit but split_arith.hip / split_node.hip into build_ab/obj/*.o (what tools/ab_variant.sh links its variants against)
set -e
cd "$(dirname "$0")/.."
mkdir -p build_ab/obj
for f in host aggregate interact dense tail eval narrow; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wno-unused-function -I include -I ihgnn_amd/csrc -o build_ab/obj/$f.o ihgnn_amd/csrc/$f.hip &
done
wait
ls -la build_ab/obj
