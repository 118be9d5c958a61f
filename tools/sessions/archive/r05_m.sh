# round 5: new parity cases - fewer than 100 boxes per variant (B = 37 / 6 / 1), tilted boxes with DR / AutoReset / baseline / split observe
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05m; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fewer_than or tilted_boxes_with" > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/pytest.log | cut -c1-300 | tail -20
