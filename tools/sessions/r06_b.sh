# round 6: the post-mortem of W-violations (tests/parity_explain.py) on the device for the first time
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06b; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "one_substep_launches or test_flat_parity or test_level4_parity or single_mjx_step or level13_dr or full_size" 2>&1 | grep -v "^$" | grep -v "^ \{4,\}" | tail -200 | tee $O/explain.txt
