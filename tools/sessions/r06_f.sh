# round 6, VERDICT item 4: the v_rsq_f32 inverse pivots against correctly rounded 1 / sqrt ones (alt_build/libpgtt_exactpivot.so, hex kernels): parity statistics and time
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06f; mkdir -p $O
export PGTT_STATS_ENVS=2048
for L in product exactpivot; do
  if [ $L = product ]; then unset PGTT_LIB; else export PGTT_LIB=$GRAFT_REPO_ROOT/alt_build/libpgtt_$L.so; fi
  echo "=== $L"
  python tools/gpu_parity_stats.py --ratios hex 2>&1 | grep -v amdgpu.ids
  for i in 1 2; do python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['kernels_ms'])"; done
done 2>&1 | tee $O/rsq_ab.txt
