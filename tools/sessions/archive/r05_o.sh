cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05o; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "flat_dr" > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  |env-steps in W" $O/pytest.log | cut -c1-300 | tail -12
