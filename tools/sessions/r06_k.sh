cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06k; mkdir -p $O
python tools/gpu_explain_big.py 2>&1 | grep -E "^==|^TOTAL|^reset:|Error|error|assert" | tee $O/explain_big.txt
