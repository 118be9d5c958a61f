cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06c; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "test_level4_parity and hex" 2>&1 | grep -v "^$" | grep -v "^ \{4,\}" | tail -30 | cut -c1-400 | tee $O/explain.txt
ls -la gpurun_out/unexplained
