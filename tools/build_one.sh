#!/bin/bash
# Experiment helper: rebuild ONE physics_kernel variant with extra flags and link it with the product objects of csrc/build/.
#   tools/build_one.sh NAME SUBS_MODE_DR_TERRAIN [extra hipcc flags...]  ->  alt_build/libpgtt_NAME.so   (use with PGTT_LIB=...)
# e.g. tools/build_one.sh e1 4_0_0_1 -DPGTT_EXP_E1      (product flags of csrc/Makefile implied; `make` first)
set -e
name=$1; v=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd); cd $root/phase_guided_terrain_traversal_amd/csrc
IFS=_ read s m d t <<< "$v"
mkdir -p $root/alt_build/$name
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp -mllvm -amdgpu-load-store-vectorizer=0 "$@" \
  -DPG_SUBS=$s -DPG_MODE=$m -DPG_DR=$d -DPG_TERRAIN=$t -Rpass-analysis=kernel-resource-usage -c pgtt_physics_inst.hip -o $root/alt_build/$name/physics_$v.o 2>&1 | grep -E "VGPRs:|AGPRs|ScratchSize|LDS Size" | sed 's/^.*remark: //' | tr '\n' ' '; echo
objs=$(ls build/*.o | grep -v "physics_$v.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $root/alt_build/libpgtt_$name.so $objs $root/alt_build/$name/physics_$v.o && echo built alt_build/libpgtt_$name.so
