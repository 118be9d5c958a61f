# GPU side: A/B of two complete builds (bit-for-bit incl. oct layout) + kernel times on the single-GPU workloads:  bash tools/ab_full.sh REF NEW
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
PGTT_AB_OCT=1 python tools/gpu_ab_bitwise.py alt_build/libpgtt_$1.so alt_build/libpgtt_$2.so 40 2>&1 | grep -v amdgpu.ids | tail -9
for n in $1 $2; do
  for A in "" "--workload flat" "--workload wfc_dr --envs 8192" "--envs 32768"; do
    PGTT_LIB=$PWD/alt_build/libpgtt_$n.so python bench.py $A --steps 300 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('%-10s %-32s %.3f M  physics %.2f us  observe %.2f us' % ('$n', '$A', d['value']/1e6, 1e3*k['physics_kernel'], 1e3*k['observe_kernel']))"
  done
done
