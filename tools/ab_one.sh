# GPU side of an experiment round:  bash tools/ab_one.sh NAME [NAME...]   (alt_build/libpgtt_NAME.so against alt_build/libpgtt_ref.so)
# per library: bit-for-bit A/B of the seeded roll-outs (tools/gpu_ab_bitwise.py) and the kernel times of the driver's bench command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/ab
for n in ref "$@"; do
  L=alt_build/libpgtt_$n.so
  if [ $n != ref ]; then python tools/gpu_ab_bitwise.py alt_build/libpgtt_ref.so $L 40 2>&1 | tail -8; fi
  for i in 1 2; do PGTT_LIB=$PWD/$L python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', round(d['value']/1e6,3), 'M', d['kernels_ms'])"; done
done
