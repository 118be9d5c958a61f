# round 5: the 12 B of scratch of physics_kernel<0,true,true,2> (oct, DR, terrain = configs[3]) under correct rounding - two allocator settings that bring it to 0
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05f; mkdir -p $O
for n in prod ra1 ra2 prod ra1 ra2; do
  PGTT_LIB=$PWD/alt_build/libpgtt_$n.so python bench.py --workload wfc_dr --envs 8192 --steps 300 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('%-5s %.3f M  physics %.2f us  observe %.2f us' % ('$n', d['value']/1e6, 1e3*k['physics_kernel'], 1e3*k['observe_kernel']))"
done | tee $O/ab_oct_dr.txt
PGTT_AB_OCT=1 python tools/gpu_ab_bitwise.py alt_build/libpgtt_prod.so alt_build/libpgtt_ra1.so 40 2>&1 | tail -3
