#!/bin/bash
# Register / scratch / LDS usage of one physics_kernel variant as the compiler reports it:  tools/kernel_resources.sh SUBS MODE DR TERRAIN [extra flags]
root=$(cd "$(dirname "$0")/.." && pwd); cd $root/phase_guided_terrain_traversal_amd/csrc
s=$1; m=$2; d=$3; t=$4; shift 4
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fno-slp-vectorize -fno-hip-fp32-correctly-rounded-divide-sqrt -mllvm -amdgpu-sched-strategy=iterative-ilp -mllvm -amdgpu-load-store-vectorizer=0 \
  -DPG_SUBS=$s -DPG_MODE=$m -DPG_DR=$d -DPG_TERRAIN=$t "$@" -Rpass-analysis=kernel-resource-usage -c pgtt_physics_inst.hip -o /tmp/kr_$$.o 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size|SGPRs:" | sed 's/^.*remark: //'
rm -f /tmp/kr_$$.o
