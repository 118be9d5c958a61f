# GPU side: the quad layout against the oct layout on box terrain at large batches:  bash tools/time_quad.sh NAME [NAME...]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for n in "$@"; do
  for A in "--envs 4096 --layout quad" "--envs 16384 --layout quad" "--envs 16384 --layout oct" "--envs 32768 --layout quad" "--envs 32768 --layout oct" "--workload wfc_dr --envs 16384 --layout quad" "--workload wfc_dr --envs 16384 --layout oct" "--envs 12288 --layout quad" "--envs 12288 --layout oct"; do
    PGTT_LIB=$PWD/alt_build/libpgtt_$n.so python bench.py $A --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('%-6s %-46s %.3f M  physics %.2f us  observe %.2f us' % ('$n', '$A', d['value']/1e6, 1e3*k['physics_kernel'], 1e3*k['observe_kernel']))"
  done
done
