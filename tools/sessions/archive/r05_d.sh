# round 5: whole -m gpu suite after the cap / terrain fixes, scale_check on the one-GPU box, driver command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05d; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | tail -30
OUT=$O/scale_check bash tools/scale_check.sh 8 100 2>&1 | tail -25
