#!/bin/bash
# Debug helper: build an alternate libpgtt under alt_build/ with extra compiler flags.
#   tools/build_variant.sh NAME [extra hipcc flags...]   (the product flags -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp are NOT implied)  ->  alt_build/libpgtt_NAME.so (use with PGTT_LIB=...)
set +m
out=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/phase_guided_terrain_traversal_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $*"
mkdir -p $root/alt_build/$out
for v in 1_0_0_0 1_0_0_1 1_0_1_0 1_0_1_1 1_1_0_0 1_1_0_1 1_1_1_0 1_1_1_1 4_0_0_0 4_0_0_1 4_0_1_0 4_0_1_1 4_1_0_0 4_1_0_1 4_1_1_0 4_1_1_1 2_0_0_0 2_0_0_1 2_0_1_0 2_0_1_1 2_1_0_0 2_1_0_1 2_1_1_0 2_1_1_1; do
  IFS=_ read s m d t <<< "$v"
  hipcc $F -DPG_SUBS=$s -DPG_MODE=$m -DPG_DR=$d -DPG_TERRAIN=$t -c pgtt_physics_inst.hip -o $root/alt_build/$out/p_$v.o 2>$root/alt_build/$out/p_$v.log &
  if [ "$v" = "1_1_1_1" ] || [ "$v" = "4_1_1_1" ]; then wait; fi
done
hipcc ${F//-mllvm -amdgpu-sched-strategy=iterative-ilp/} -c pgtt_api.hip -o $root/alt_build/$out/api.o 2>/dev/null &     # the api kernels keep the default scheduler (Makefile)
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $root/alt_build/libpgtt_$out.so $root/alt_build/$out/*.o && echo built alt_build/libpgtt_$out.so
