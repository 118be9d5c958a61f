# round 6, first contact: the reference-pinned stochastic branches through the C ABI, then the whole -m gpu suite and the driver's command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06a; mkdir -p $O
python -m pytest tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -5 | tee $O/golden.txt
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $O/gpu_suite.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_bench.json 2>$O/driver_cmd_bench.err; echo "driver rc=$?"; cut -c1-600 $O/driver_cmd_bench.json
