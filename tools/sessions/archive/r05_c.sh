# round 5: diagnostics of the parity caps that the correctly rounded build trips + the MJX pin kit's GPU tests on the dry-run file + parity statistics
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05c; mkdir -p $O
for lay in hex oct quad; do python tools/gpu_reset_warm_diag.py level13 128 $lay 2>&1 | grep -v amdgpu.ids; done | tee $O/reset_warm_diag.txt
timeout 1200 python -m pytest tests/test_mjx_pin.py tests/test_gpu_parity.py -q -m gpu -s -k "mjx or tilted or curriculum or equal_depth or variant_label" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|^\[|env-steps in W|many-box|HIP" $O/pytest_sel.log | cut -c1-600 | tail -60
timeout 1500 python tools/gpu_parity_stats.py $O/parity_stats.json > $O/parity_stats.txt 2>&1; grep -E "^(flat|level)|filter r64<1e-6&agree" $O/parity_stats.txt | cut -c1-400
