# round 6: spread of the driver's 20-step window over repeated runs on one box (after the collector was switched off round the window)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06q; mkdir -p $O
for i in $(seq 1 12); do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run', d['value'], d['ms_per_step'], d['wall_over_kernels'], d['kernels_ms']['physics_kernel'])"; done | tee $O/window_spread.txt
