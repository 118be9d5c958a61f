# round 4, GPU call C: the whole -m gpu suite, then one measurement round (tools/profile_round.sh)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04c; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
ROUND_TAG=r04a bash tools/profile_round.sh
