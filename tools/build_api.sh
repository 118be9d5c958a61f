#!/bin/bash
# Experiment helper: rebuild pgtt_api.hip (observe / task / reset kernels) with extra flags and link it with the product objects of csrc/build/.
#   tools/build_api.sh NAME [extra hipcc flags...]  ->  alt_build/libpgtt_NAME.so
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd); cd $root/phase_guided_terrain_traversal_amd/csrc
mkdir -p $root/alt_build/$name
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fno-slp-vectorize -fno-hip-fp32-correctly-rounded-divide-sqrt "$@" -c pgtt_api.hip -o $root/alt_build/$name/api.o
objs=$(ls build/*.o | grep -v "build/api.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $root/alt_build/libpgtt_$name.so $objs $root/alt_build/$name/api.o && echo built alt_build/libpgtt_$name.so
