# GPU side of a flag / variant sweep:  [BENCH_ARGS="--workload flat"] bash tools/time_libs.sh NAME [NAME...]  -> physics_kernel time of alt_build/libpgtt_NAME.so (300-step bench, twice)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for n in "$@"; do
  L=alt_build/libpgtt_$n.so
  for i in 1 2; do PGTT_LIB=$PWD/$L python bench.py $BENCH_ARGS --steps 300 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('%-16s %.3f M  physics %.2f us  observe %.2f us' % ('$n', d['value']/1e6, 1e3*k['physics_kernel'], 1e3*k['observe_kernel']))"; done
done
