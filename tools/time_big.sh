# GPU side: big-batch timings (automatic layout = quad beyond 8192 envs) of alternate libraries:  bash tools/time_big.sh NAME [NAME...]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for n in "$@"; do
  for A in "--envs 16384" "--envs 32768" "--workload wfc_dr --envs 16384" "--workload flat --envs 16384"; do
    for i in 1 2; do PGTT_LIB=$PWD/alt_build/libpgtt_$n.so python bench.py $A --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('%-6s %-40s %.3f M  physics %.2f us  observe %.2f us' % ('$n', '$A', d['value']/1e6, 1e3*k['physics_kernel'], 1e3*k['observe_kernel']))"; done
  done
done
