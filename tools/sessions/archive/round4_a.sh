# round 4, GPU call A2: the many-box pass, full-size parity
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -s -k "more_penetrating or wfc_dr_parity_full_size" > $O/pytest_sel2.log 2>&1; echo "pytest sel rc=$?"; tail -5 $O/pytest_sel2.log
grep -E "^(graded|equal_tops) |^\[stairs n=8192|env-steps in W" $O/pytest_sel2.log | cut -c1-600
