"""Which envs of a reset differ from the oracle in qacc_warmstart by more than the bar of tests/test_gpu_parity.py::run_parity (2e-2 relative), and why:
per offending env the Newton iteration counts of both sides, the contact distances closest to zero and the largest |qacc| entries.   (GPU box)
    python tools/gpu_reset_warm_diag.py [level13|level4] [n] [layout]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as P
from oracle import oracle
lvl = sys.argv[1] if len(sys.argv) > 1 else "level13"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
P.EXEC["layout"] = sys.argv[3] if len(sys.argv) > 3 else "hex"
terrain = np.load(os.path.join(P.ASSETS, "terrains", lvl + ".npy"))
env, hb, cs, ms = P.make_pair("stairs", n, terrain, dr=True, autoreset=True)
h64 = oracle.HostBuffers(n, with_params=True, with_variant=True, with_box_friction=True)
for k in ("params", "variant", "box_friction"):
    h64[k][...] = hb[k]
env.reset(3); oracle.reset(cs, ms, terrain, hb, seed=3, nthreads=8); oracle.reset(cs, ms, terrain, h64, seed=3, nthreads=8, fp64=True)
torch.cuda.synchronize()
g = {k: v.cpu().numpy() for k, v in env.buffers.items()}
rel = lambda a, b: (np.abs(a - b) / (1 + np.abs(b))).max(0)
e_go, e_g64, e_o64 = rel(g["state"][37:55], hb["state"][37:55]), rel(g["state"][37:55], h64["state"][37:55]), rel(hb["state"][37:55], h64["state"][37:55])
ni_g, ni_o, ni_64 = g["dbg_niter"] & 0xFFFF, hb["dbg_niter"], h64["dbg_niter"]
print(f"{lvl} n={n} layout={P.EXEC['layout']}: envs with warm error > 2e-2: GPU vs oracle-f32 {(e_go > 2e-2).sum()}, GPU vs oracle-f64 {(e_g64 > 2e-2).sum()}, oracle-f32 vs oracle-f64 {(e_o64 > 2e-2).sum()}")
for e in np.argsort(-e_go)[:5]:
    print(f" env {e}: warm rel err GPU|o32 {e_go[e]:.2e} GPU|o64 {e_g64[e]:.2e} o32|o64 {e_o64[e]:.2e}; newton iterations GPU {ni_g[e]} o32 {ni_o[e]} o64 {ni_64[e]}")
    print("    contact dist GPU", np.round(g["dbg_dist"][e], 6), "\n                 o32", np.round(hb["dbg_dist"][e], 6), "\n                 o64", np.round(h64["dbg_dist"][e], 6))
    print("    max |warm| GPU %.3f o32 %.3f o64 %.3f" % (np.abs(g["state"][37:55, e]).max(), np.abs(hb["state"][37:55, e]).max(), np.abs(h64["state"][37:55, e]).max()))
