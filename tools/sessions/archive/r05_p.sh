cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05p; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_abi.py -q -m gpu -s -k "config_values or refuses_what" > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  |env-steps in W" $O/pytest.log | cut -c1-300 | tail -12
python -m pytest tests/test_abi.py -q 2>&1 | tail -2
