import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_parity import sync_to_host, dense_terrain, active_sets
from oracle import oracle
from phase_guided_terrain_traversal_amd import abi, configs, mjcf
from phase_guided_terrain_traversal_amd.env import Joystick
np.set_printoptions(precision=5, suppress=True, linewidth=220)
terrain = dense_terrain(); n = 64
cfg = configs.with_overrides(configs.training_config(), ctrl_dt=0.005)       # ONE substep per step: dbg contacts belong to the synced state
variant = np.random.default_rng(2).integers(0, terrain.shape[0], n).astype(np.int32)
env = Joystick("stairs", cfg, num_envs=n, terrain=terrain, device="cuda:0", debug_contacts=True, variant=torch.from_numpy(variant))
cs, ms = abi.config_struct(env.config), abi.model_struct(mjcf.load_model("stairs"))
hb = oracle.HostBuffers(n, with_variant=True); hb["variant"][:] = variant
env.reset(3); oracle.reset(cs, ms, terrain, hb, seed=3, nthreads=8); torch.cuda.synchronize()
rng = np.random.default_rng(1); shown = 0; total = 0
for k in range(120):
    sync_to_host(env, hb)
    prev = hb["state"].copy()
    act = np.tanh(rng.normal(size=(n, 12)) * 0.6).astype(np.float32)
    env.step(torch.from_numpy(act).cuda()); oracle.step(cs, ms, terrain, hb, act, seed=3, nthreads=8); torch.cuda.synchronize()
    g = {kk: v.cpu().numpy() for kk, v in env.buffers.items()}
    ga, ha = active_sets(g["dbg_contact"], g["dbg_dist"]), active_sets(hb["dbg_contact"], hb["dbg_dist"])
    for e in range(n):
        if ga[e] != ha[e]:
            total += 1
            if shown < 5:
                shown += 1
                print("step", k, "env", e, "variant", variant[e]); print(" gpu", ga[e], g["dbg_dist"][e]); print(" cpu", ha[e], hb["dbg_dist"][e])
                d = oracle.forward(ms, prev[:19, e].astype(np.float64), prev[19:37, e].astype(np.float64), np.zeros(12), boxes=terrain[variant[e]], fp64=False)
                fx = d["foot_xpos"].astype(np.float32); boxes = terrain[variant[e]]
                keys = np.linalg.norm(boxes[None, :, :3] - fx[:, None, :], axis=2).reshape(-1)
                order = np.argsort(keys, kind="stable")
                rank = {int(i): r for r, i in enumerate(order)}
                print(" oracle.forward slots", list(zip(d["con_foot"][4:], d["con_box"][4:], np.round(d["con_dist"][4:], 5))))
                for (f, b) in sorted(set(ga[e]) | set(ha[e])):
                    if b >= 0: print("   pair", (f, b), "rank", rank[f * 100 + b], "key", keys[f * 100 + b])
print("total mismatching env-steps", total)
