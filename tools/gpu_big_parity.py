"""Large-sample run of the parity harness (tests/test_gpu_parity.run_parity): 1024 envs x 100 control steps on level4 per layout and
512 x 80 on level13 with full DR.  Its output is kept as profiles/archive/r02d_parity_big.txt / r02e_parity_big_oct.txt.   usage: python tools/gpu_big_parity.py [layout ...]"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import tests.test_gpu_parity as T
terrain = np.load(os.path.join(T.ASSETS, "terrains", "level4.npy"))
for lay in (sys.argv[1:] or ["hex", "quad", "oct"]):
    T.EXEC["layout"] = lay
    st = T.run_parity("stairs", 1024, terrain, steps=100)
    print(lay, "OK", {k: st[k] for k in ("frac_gpu_1e4", "frac_fp_1e4", "well_frac", "well_flag_mismatch", "well_set_mismatch")}, st["well_violations"])
t13 = np.load(os.path.join(T.ASSETS, "terrains", "level13.npy"))
for lay in (sys.argv[1:] or ["hex", "oct"]):
    T.EXEC["layout"] = lay
    st = T.run_parity("stairs", 512, t13, steps=80, dr=True, autoreset=True)
    print(lay, "level13 dr OK", {k: st[k] for k in ("frac_gpu_1e4", "frac_fp_1e4", "well_frac", "well_flag_mismatch", "well_set_mismatch")}, st["well_violations"])
