cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06i; mkdir -p $O
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "wrapper_contract or episode_and_autoreset" 2>&1 | tail -25 | cut -c1-300 | tee $O/wrapper.txt
python -m pytest tests/test_gpu_bench.py -q -m gpu 2>&1 | tail -15 | cut -c1-300 | tee $O/bench_test.txt
