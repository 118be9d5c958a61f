cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06l; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "every_device_substep" 2>&1 | grep -E "^\[|cut on both|passed|failed|Error|assert|unexplained" | cut -c1-600 | tee $O/audit.txt
