cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05q; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "model_values" > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  |env-steps in W|well_frac" $O/pytest.log | cut -c1-420 | tail -20
python - <<'PY' 2>&1 | grep -v amdgpu | tail -8
# exploratory: the same with tilted joint axes (does the kernel take jnt_axis from the model?)
import sys, os; sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
import test_gpu_parity as T
T.EXEC["layout"] = "hex"
try:
    st = T.run_parity("flat_terrain", 128, None, steps=16, model=T.perturbed_model("flat_terrain", axes=True), w_floor=0.3, cap_scale=4.0, med_tol=1e-4)
    print("tilted joint axes: parity holds", {k: st[k] for k in ("well_frac", "med_gpu", "frac_gpu_1e4", "frac_fp_1e4")})
except AssertionError as e:
    print("tilted joint axes: parity FAILS", str(e)[:300])
PY
