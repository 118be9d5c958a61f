# round 5: the two allocator settings that freed the oct DR kernel, tried on the other kernels that carry a BASELINE row (hex terrain = headline, hex flat, oct terrain, quad terrain at 32768)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05h; mkdir -p $O
run() { PGTT_LIB=$PWD/alt_build/libpgtt_$1.so python bench.py $2 --steps 300 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('%-12s %-28s %.3f M  physics %.2f us' % ('$1', '$2', d['value']/1e6, 1e3*k['physics_kernel']))"; }
for rep in 1 2; do
  for n in prod ss_4_0_0_1 tg_4_0_0_1; do run $n ""; done
  for n in prod ss_4_0_0_0 tg_4_0_0_0; do run $n "--workload flat"; done
  for n in prod ss_2_0_0_1 tg_2_0_0_1; do run $n "--envs 8192"; done
  for n in prod ss_1_0_0_1 tg_1_0_0_1; do run $n "--envs 32768"; done
done | tee $O/ab_alloc.txt
