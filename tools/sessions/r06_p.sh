cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06p; mkdir -p $O
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s -k "task_layer_is_the_oracles" 2>&1 | grep -E "^\[|passed|failed|Error|assert|errs" | cut -c1-500 | tee $O/task_audit.txt
