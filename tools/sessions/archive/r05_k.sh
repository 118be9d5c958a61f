# round 5: observe_kernel against the reference's closed-loop roll-out records + the driver-command contract test (counters must not be stale now)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05k; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_golden.py tests/test_gpu_bench.py -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?"; tail -15 $O/pytest.log | cut -c1-400
