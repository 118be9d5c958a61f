cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05u; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "extreme_states" > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  |extreme states" $O/pytest.log | cut -c1-400 | tail -14
