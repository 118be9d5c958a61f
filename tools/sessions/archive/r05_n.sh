# round 5: more parity cases - no variant buffer, 2 / 8 substeps per control step, Episode(7) + AutoReset
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05n; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "single_variant or other_substep or short_episodes" > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  |well_frac|env-steps in W" $O/pytest.log | cut -c1-420 | tail -30
