# round 5: the scheduler settings of rounds 1 - 3 once more on the correctly rounded hex terrain kernel (headline): default / max-ilp / iterative-minreg / max-memory-clause scheduling, load-store vectoriser on
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05w; mkdir -p $O
run() { PGTT_LIB=$PWD/alt_build/libpgtt_$1.so python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('%-12s %.3f M  physics %.2f us' % ('$1', d['value']/1e6, 1e3*k['physics_kernel']))"; }
for rep in 1 2; do for n in prod f_default f_maxilp f_lsv f_maxmem f_minreg; do run $n; done; done | tee $O/ab_sched.txt
