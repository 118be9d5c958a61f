# round 6: the whole parity file with the post-mortem on (no -x: collect every unexplained case in one call)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06e; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -s 2>&1 | grep -E "^\[|post-mortem|^reset:|UNEXPLAINED|ALL env-steps|^FAILED|passed|failed|Error" | cut -c1-600 | tee $O/explain_all.txt
ls gpurun_out/unexplained 2>/dev/null
