import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_parity import make_pair, sync_to_host, ASSETS
from oracle import oracle
from phase_guided_terrain_traversal_amd import abi
task = sys.argv[1] if len(sys.argv) > 1 else "flat_terrain"
terrain = None if task == "flat_terrain" else np.load(os.path.join(ASSETS, "terrains", "level4.npy"))
n = 64
env, hb, cs, ms = make_pair(task, n, terrain)
env.reset(3); oracle.reset(cs, ms, terrain, hb, seed=3, nthreads=8); torch.cuda.synchronize()
def rows(name, a, b, axis=1):
    d = np.abs(a - b).max(axis=axis)
    bad = np.nonzero(d > 1e-5)[0]
    print(name, "max", d.max(), "bad rows", [(int(i), float(d[i])) for i in bad[:40]])
g = {k: v.cpu().numpy() for k, v in env.buffers.items()}
rows("reset state", g["state"], hb["state"]); rows("reset frame", g["frame"], hb["frame"])
rows("reset obs", g["obs_state"], hb["obs_state"], 0); rows("reset priv", g["obs_priv"], hb["obs_priv"], 0)
print("istate eq", np.array_equal(g["istate"], hb["istate"]), g["istate"][:, :4], hb["istate"][:, :4])
rng = np.random.default_rng(1)
for k in range(3):
    sync_to_host(env, hb)
    act = np.tanh(rng.normal(size=(n, 12)) * 0.6).astype(np.float32)
    env.step(torch.from_numpy(act).cuda()); oracle.step(cs, ms, terrain, hb, act, seed=3, nthreads=8); torch.cuda.synchronize()
    g = {k: v.cpu().numpy() for k, v in env.buffers.items()}
    print("---- step", k)
    rows("state", g["state"], hb["state"]); rows("frame", g["frame"], hb["frame"]); rows("scan", g["scan_z"], hb["scan_z"], 0)
    rows("obs", g["obs_state"], hb["obs_state"], 0); rows("priv", g["obs_priv"], hb["obs_priv"], 0)
    rows("metrics", g["metrics"], hb["metrics"]); print("reward", np.abs(g["reward"] - hb["reward"]).max())
    print("istate eq", np.array_equal(g["istate"], hb["istate"]))
    e = int(np.abs(g["state"][:19] - hb["state"][:19]).max(axis=0).argmax())
    print("worst env", e, "gpu qpos", g["state"][:7, e], "cpu", hb["state"][:7, e])
    print("gpu con", g["dbg_contact"][e], g["dbg_dist"][e]); print("cpu con", hb["dbg_contact"][e], hb["dbg_dist"][e])
