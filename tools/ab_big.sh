# GPU side: bit-for-bit A/B of two complete builds + kernel times incl. the big batches:  bash tools/ab_big.sh REF NEW
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
PGTT_AB_OCT=1 python tools/gpu_ab_bitwise.py alt_build/libpgtt_$1.so alt_build/libpgtt_$2.so 40 2>&1 | grep -v amdgpu.ids | cut -c1-140 | tail -9
for n in $1 $2; do
  for A in "" "--workload wfc_dr --envs 8192" "--envs 16384" "--envs 32768" "--workload wfc_dr --envs 16384"; do
    PGTT_LIB=$PWD/alt_build/libpgtt_$n.so python bench.py $A --steps 300 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('%-6s %-36s %.3f M  physics %.2f us  observe %.2f us' % ('$n', '$A', d['value']/1e6, 1e3*k['physics_kernel'], 1e3*k['observe_kernel']))"
  done
done
