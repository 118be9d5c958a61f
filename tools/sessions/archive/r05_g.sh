# round 5: per-variant allocator flag for the oct DR terrain kernel - bit-for-bit against the build before it (all layouts incl. the DR kernels), then configs[3] timing
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05g; mkdir -p $O
PGTT_AB_OCT=1 python tools/gpu_ab_bitwise.py alt_build/libpgtt_prod.so phase_guided_terrain_traversal_amd/libpgtt.so 40 2>&1 | grep -v amdgpu.ids | tee $O/ab_bitwise.txt | tail -12
for n in alt_build/libpgtt_prod.so phase_guided_terrain_traversal_amd/libpgtt.so alt_build/libpgtt_prod.so phase_guided_terrain_traversal_amd/libpgtt.so; do
  PGTT_LIB=$PWD/$n python bench.py --workload wfc_dr --envs 8192 --steps 300 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('%-50s %.3f M  physics %.2f us  observe %.2f us' % ('$n', d['value']/1e6, 1e3*k['physics_kernel'], 1e3*k['observe_kernel']))"
done | tee $O/ab_oct_dr.txt
