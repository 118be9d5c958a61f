#!/bin/bash
# Experiment helper: rebuild SEVERAL physics_kernel variants with extra flags and link them with the product objects of csrc/build/ (`make` first).
#   tools/build_multi.sh NAME "4_0_0_1 4_1_0_1 ..." [extra hipcc flags...]  ->  alt_build/libpgtt_NAME.so   (use with PGTT_LIB=...)
set -e
name=$1; vars=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd); cd $root/phase_guided_terrain_traversal_amd/csrc
mkdir -p $root/alt_build/$name
objs=$(ls build/*.o)
for v in $vars; do
  IFS=_ read s m d t <<< "$v"
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp -mllvm -amdgpu-load-store-vectorizer=0 "$@" \
      -DPG_SUBS=$s -DPG_MODE=$m -DPG_DR=$d -DPG_TERRAIN=$t -Rpass-analysis=kernel-resource-usage -c pgtt_physics_inst.hip -o $root/alt_build/$name/physics_$v.o 2>&1 | grep -E "VGPRs:|AGPRs|ScratchSize" | sed 's/^.*remark: //' | tr '\n' ' '; echo " <- $v" ) &
  objs=$(echo "$objs" | grep -v "physics_$v.o")
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $root/alt_build/libpgtt_$name.so $objs $root/alt_build/$name/physics_*.o && echo built alt_build/libpgtt_$name.so
