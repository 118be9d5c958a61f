"""-DPGTT_EFFORT builds: how much of a physics wave's solver work is the company it keeps?  Every env records the line-search rounds it
NEEDED in each Newton trip of each substep (3 bits per trip); the tool replays the wave's cost under the env -> wave groupings
  fixed    : envs 4 w .. 4 w + 3 (what the kernel does),
  sorted   : envs ordered by the work they needed in the PREVIOUS control step (a grouping a launch could be given),
  oracle   : envs ordered by the work they need in THIS step (the bound of any grouping by a scalar key),
with cost = trips x C_TRIP + rounds x C_ROUND (shader ticks of one Newton trip without its bracketing rounds / of one round, from
profiles/*_stage_time.txt).   usage: PGTT_LIB=alt_build/libpgtt_effort.so python tools/gpu_effort.py [level4|flat] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phase_guided_terrain_traversal_amd import configs
from phase_guided_terrain_traversal_amd.env import Joystick
wl = sys.argv[1] if len(sys.argv) > 1 else "level4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
C_TRIP, C_ROUND, n, G = 7160.0, 940.0, 4096, 4
assets = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "phase_guided_terrain_traversal_amd", "assets")
terrain = None if wl == "flat" else np.load(os.path.join(assets, "terrains", "level4.npy"))
variant = None if terrain is None else torch.from_numpy(np.sort(np.random.default_rng(0).integers(0, terrain.shape[0], n)).astype(np.int32))
env = Joystick("flat_terrain" if wl == "flat" else "stairs", configs.training_config(), num_envs=n, terrain=terrain, device="cuda:0", variant=variant, autoreset=True, debug_contacts=True)
env.reset(seed=1)
g = torch.Generator(device="cuda").manual_seed(0)
pool = [torch.tanh(torch.randn(n, 12, generator=g, device="cuda") * 0.6) for _ in range(32)]
rec = []; hess = []
for k in range(steps):
    env.step(pool[k % 32])
    dc = env.buffers["dbg_contact"].cpu().numpy().reshape(n, 16)
    eff = dc[:, 14].astype(np.uint32).astype(np.uint64) | (dc[:, 15].astype(np.uint32).astype(np.uint64) << np.uint64(32))
    need = np.stack([(eff >> np.uint64(3 * i)) & np.uint64(7) for i in range(20)], 1).astype(np.int64)     # [env][substep * 5 + trip]: 0 = not in the trip, else 1 + rounds
    rec.append(need)
    hess.append(dc[::G, 12:14].astype(np.int64))          # per wave: Hessian builds, of which with no row of any lane having changed sides
rec = np.stack(rec)[50:]                                   # past the reset transient
def cost(need, order):
    w = need[order].reshape(n // G, G, 20)
    trips = (w > 0).any(1).sum(1)                         # trips some env of the wave is in
    rounds = np.maximum(w - 1, 0).max(1).sum(1)           # rounds of a trip = the most any of its envs needs
    return trips * C_TRIP + rounds * C_ROUND, trips, rounds
own = lambda need: ((need > 0).sum(1) * C_TRIP + np.maximum(need - 1, 0).sum(1) * C_ROUND)     # an env on its own
tot = {k: [] for k in ("fixed", "sorted", "oracle", "alone")}; tr = []; rd = []
for t in range(1, len(rec)):
    c, trips, rounds = cost(rec[t], np.arange(n)); tot["fixed"].append(c.max()); tr.append(trips.mean()); rd.append(rounds.mean())
    tot["fixed_mean"] = tot.get("fixed_mean", []) + [c.mean()]
    for name, key in (("sorted", own(rec[t - 1])), ("oracle", own(rec[t]))):
        c, _, _ = cost(rec[t], np.argsort(key, kind="stable")); tot[name].append(c.max()); tot[name + "_mean"] = tot.get(name + "_mean", []) + [c.mean()]
    tot["alone"].append(own(rec[t]).mean())
print(f"{wl}: {len(rec) - 1} control steps of {n} envs; wave of {G} envs: Newton trips {np.mean(tr):.2f}, line-search rounds {np.mean(rd):.1f} per control step (fixed grouping)")
# histogram of the Newton trips an env needs in ONE substep (5 = the cap of go2_mjx_feetonly.xml:17): what a "main pass of k trips + compacted tail pass" would leave for the tail
per_sub = (rec.reshape(-1, n, 4, 5) > 0).sum(3).ravel()
hist = np.bincount(per_sub, minlength=6) / per_sub.size
print("  Newton trips of an env in one substep: " + "  ".join(f"{k}: {100 * hist[k]:.1f} %" for k in range(6)) + f"   -> needs trips 4 - 5: {100 * hist[4:].sum():.1f} % of env-substeps")
print(f"  an env on its own: trips {(rec > 0).sum(2).mean():.2f}, rounds {np.maximum(rec - 1, 0).sum(2).mean():.1f}; solver ticks {np.mean(tot['alone']):.0f}")
for k in ("fixed", "sorted", "oracle"):
    print(f"  {k:7s}: solver ticks of the mean wave {np.mean(tot[k + '_mean']):9.0f}   of the slowest wave {np.mean(tot[k]):9.0f}")
# round 6: the CEILING of any de-synchronised bracketing - a wave whose envs each follow their own (trip, round) cursor can at best
# last as long as its slowest env's own path (exec-masked paths of different stages add up, they do not overlap): mean / slowest wave of max_env own ticks
own_w = np.stack([own(rec[t]).reshape(n // G, G).max(1) for t in range(1, len(rec))])
print(f"  de-synchronised ceiling (max over a wave's envs of the env's OWN solver ticks): mean wave {own_w.mean():9.0f}   slowest wave {own_w.max(1).mean():9.0f}"
      f"   -> the launch lasts as long as its slowest wave: {np.mean(tot['fixed']):.0f} today, {own_w.max(1).mean():.0f} at the ceiling ({100 * (1 - own_w.max(1).mean() / np.mean(tot['fixed'])):.1f} % of the solver ticks)")
a, b = own(rec[1:].reshape(-1, 20)).reshape(len(rec) - 1, n), own(rec[:-1].reshape(-1, 20)).reshape(len(rec) - 1, n)
print(f"  correlation of an env's solver work with its previous step's: {np.corrcoef(a.ravel(), b.ravel())[0, 1]:.3f}")
hess = np.stack(hess)[51:]
wcost = np.stack([cost(rec[t], np.arange(n))[0] for t in range(1, len(rec))])
slow = wcost >= np.percentile(wcost, 99, axis=1, keepdims=True)
print(f"  Hessian builds per wave and control step {hess[..., 0].mean():.2f}, of which with NO row of the wave on a new side since the last trip: "
      f"{hess[..., 1].mean():.2f} (all waves), {hess[..., 1][slow].mean():.2f} of {hess[..., 0][slow].mean():.2f} (the slowest 1 % of each step)")
