# round 5: the whole -m gpu suite without -x (after the first call stopped at one failure)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05b; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu --durations=10 -k "not test_driver_command_line_contract" > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | tail -30
