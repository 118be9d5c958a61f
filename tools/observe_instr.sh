#!/bin/bash
# Dynamic instruction counts of observe_kernel per phase (see tools/gpu_observe_instr.py); run on the GPU box from the repo root:
#   tools/build_variant.sh obsstop -DPGTT_OBS_STOP -fno-slp-vectorize    (here)      then      bash tools/observe_instr.sh > gpurun_out/observe_instr.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for k in 0 1 2 3 4 5 6 7 -1; do
  rm -rf /tmp/oi_$k
  PGTT_LIB=$R/alt_build/libpgtt_obsstop.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES -d /tmp/oi_$k -o p --output-format csv -- python $R/tools/gpu_observe_instr.py $k > /dev/null 2>&1
  echo "== stop $k"; python $R/tools/pmc_summary.py /tmp/oi_$k | grep "observe_kernel<0" 
done
