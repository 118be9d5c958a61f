# round 5: whole -m gpu suite + smoke on the build with the per-variant allocator flag, then the measurement round (r05b)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05j; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu.log | tail -10
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
ROUND_TAG=r05b bash tools/profile_round.sh > $O/profile_round.log 2>&1; tail -16 $O/profile_round.log | cut -c1-260
