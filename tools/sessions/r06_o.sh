cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06o; mkdir -p $O
python -m pytest tests/test_gpu_policy.py -q -m gpu -s -k "policy_family" 2>&1 | grep -E "^policy|passed|failed|Error|assert" | cut -c1-400 | tee $O/family_test.txt
