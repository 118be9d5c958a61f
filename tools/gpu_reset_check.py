"""Flat reset on the GPU vs the oracle: Newton iteration histogram and qacc error (debug helper)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_parity import make_pair
from oracle import oracle
n = 64
env, hb, cs, ms = make_pair("flat_terrain", n, None)
env.reset(3); oracle.reset(cs, ms, None, hb, seed=3, nthreads=8); torch.cuda.synchronize()
g = {k: v.cpu().numpy() for k, v in env.buffers.items()}
err = np.abs(g["state"][37:55] - hb["state"][37:55]).max()
print(os.environ.get("PGTT_LIB", "default"), "niter gpu", np.bincount(g["dbg_niter"] & 0xFFFF, minlength=6).tolist(), "cpu", np.bincount(hb["dbg_niter"], minlength=6).tolist(), "qacc err", err)
