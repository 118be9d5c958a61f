"""The post-mortem of tests/parity_explain.py on a LARGE sample: run_parity (the suite's own function, every assertion on) with 2048 envs x 50 control steps
per workload and lane layout - ~100 k env-steps each - and the ledger of the env-steps of W that miss a bar.     python tools/gpu_explain_big.py    (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_parity as P
A = os.path.join(ROOT, "phase_guided_terrain_traversal_amd", "assets", "terrains")
tot = {}
for lay in ("hex", "oct", "quad"):
    P.EXEC["layout"] = lay
    for name, kw in (("flat", dict(task="flat_terrain", terrain=None)), ("level4", dict(task="stairs", terrain=np.load(os.path.join(A, "level4.npy")))),
                     ("level13+DR+AutoReset", dict(task="stairs", terrain=np.load(os.path.join(A, "level13.npy")), dr=True, autoreset=True)),
                     ("level4, one mjx.step", dict(task="stairs", terrain=np.load(os.path.join(A, "level4.npy")), ctrl_dt=0.005))):
        t0 = time.time()
        try:
            st = P.run_parity(kw.pop("task"), 2048, kw.pop("terrain"), steps=50, **kw)
        except AssertionError as ex:
            import traceback
            print(f"== {lay} {name}: ASSERTION {ex!r} at {traceback.extract_tb(ex.__traceback__)[-1].line}", flush=True)
            continue
        ex = st["explained"]
        print(f"== {lay} {name}: W = {st['well_frac']:.1%} of 102400 env-steps, {ex['violations']} miss a bar: {ex}   ({time.time() - t0:.0f} s)", flush=True)
        for k, v in ex.items():
            if isinstance(v, int):
                tot[k] = tot.get(k, 0) + v
print("TOTAL over 12 runs x 102400 env-steps:", tot)
