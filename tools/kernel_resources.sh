#!/bin/bash
# Register / scratch / LDS usage of physics_kernel variants as the compiler reports it, with the product's flags (csrc/Makefile):
#   tools/kernel_resources.sh SUBS MODE DR TERRAIN [extra flags]      one variant, full report
#   tools/kernel_resources.sh all [extra flags]                       one line per variant (24)
root=$(cd "$(dirname "$0")/.." && pwd); cd $root/phase_guided_terrain_traversal_amd/csrc
one() {
  s=$1; m=$2; d=$3; t=$4; shift 4
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp -mllvm -amdgpu-load-store-vectorizer=0 \
    -DPG_SUBS=$s -DPG_MODE=$m -DPG_DR=$d -DPG_TERRAIN=$t "$@" -Rpass-analysis=kernel-resource-usage -c pgtt_physics_inst.hip -o /tmp/kr_$$.o 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size|SGPRs:" | sed 's/^.*remark: //; s/ \[-Rpass.*$//'
  rm -f /tmp/kr_$$.o
}
if [ "$1" = all ]; then
  shift
  for s in 4 2 1; do for m in 0 1; do for d in 0 1; do for t in 0 1; do
    echo "subs=$s mode=$m dr=$d terrain=$t: $(one $s $m $d $t "$@" | grep -E "VGPRs:|AGPRs|ScratchSize|LDS Size" | sed 's/^ *//' | tr '\n' ';')"
  done; done; done; done
else
  one "$@"
fi
