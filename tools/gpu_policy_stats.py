"""Distribution-level comparison with the reference's simulator: policy177's normaliser holds mean / std of the 215 privileged-observation
rows over the 443 M samples of its training run (MJX).  Roll the same policy out here under the training conditions of its last
curriculum stage (level13, full randomize.py DR, observation noise, the task's own command / gait-frequency sampling, AutoReset) and
print both, block by block.      python tools/gpu_policy_stats.py [level13]      (GPU box)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phase_guided_terrain_traversal_amd import abi, configs, mjcf
from phase_guided_terrain_traversal_amd.env import Joystick
from phase_guided_terrain_traversal_amd.policy import load_policy
from phase_guided_terrain_traversal_amd.randomize import domain_randomize

BLOCKS = [("gyro", 0, 3), ("gravity", 3, 6), ("joint pos - default", 6, 18), ("joint vel", 18, 30), ("cos phase", 30, 34), ("sin phase", 34, 38),
          ("scan - min", 38, 155), ("gait freq", 155, 156), ("last action", 156, 168), ("command", 168, 171), ("local linvel", 171, 174),
          ("accelerometer", 174, 177), ("global angvel", 177, 180), ("actuator force", 180, 192), ("last contact", 192, 196),
          ("feet linvel", 196, 208), ("feet air time", 208, 212)]


BLOCKS_BASELINE = [("gyro", 0, 3), ("gravity", 3, 6), ("joint pos - default", 6, 18), ("joint vel", 18, 30), ("scan - min", 30, 147),
                   ("last action", 147, 159), ("command", 159, 162), ("local linvel", 162, 165), ("accelerometer", 165, 168),
                   ("global angvel", 168, 171), ("actuator force", 171, 183), ("last contact", 183, 187), ("feet linvel", 187, 199),
                   ("feet air time", 199, 203)]


def rollout_stats(level="level13", n=2048, steps=700, seed=0, dr=True, stochastic=False, kv=None, policy="policy177", method="pgtt", layout=None, hist=None):
    """hist: optional dict that receives the histogram of max(0, -dist) over the box contacts met during the rollout (see __main__)"""
    assets = os.path.join(os.path.dirname(mjcf.__file__), "assets")
    flat = level == "flat"
    terrain = None if flat else np.load(os.path.join(assets, "terrains", level + ".npy"))
    model = mjcf.load_model("flat_terrain" if flat else "stairs")
    if kv is not None:
        model = mjcf.with_bias_velocity(model, kv)
    kw = {"model": model, "layout": layout, "debug_contacts": hist is not None}
    if dr and flat:
        kw["params"] = torch.from_numpy(domain_randomize(model, n, seed=5)["params"])
    elif flat:
        pass
    elif dr:
        out = domain_randomize(model, n, seed=5, terrain=terrain)
        kw.update(variant=torch.from_numpy(out["variant"]), params=torch.from_numpy(out["params"]), box_friction=torch.from_numpy(out["box_friction"]))
    else:
        kw["variant"] = torch.from_numpy(np.random.default_rng(2).integers(0, terrain.shape[0], n).astype(np.int32))
    env = Joystick("flat_terrain" if flat else "stairs", configs.training_config(method), num_envs=n, terrain=terrain, device="cuda:0", autoreset=True, **kw)
    pi = load_policy(policy)
    env.reset(seed)
    s1 = torch.zeros(env.observation_size["privileged_state"], device="cuda:0", dtype=torch.float64); s2 = torch.zeros_like(s1); cnt = 0
    for k in range(steps):
        env.step(pi.sample(env.buffers["obs_state"]) if stochastic else pi(env.buffers["obs_state"]))
        if hist is not None and k >= 100:
            # box-contact slots 4..7 of the debug record (foot, geom >= 0 = box index, dist): penetration depth of every box contact
            dist = env.buffers["dbg_dist"][:, 4:8]; geom = env.buffers["dbg_contact"].view(n, abi.NCON, 2)[:, 4:8, 1]
            pen = (-dist[geom >= 0]).clamp(min=0)
            hist["counts"] = hist.get("counts", 0) + torch.histc(pen, bins=len(hist["edges"]) - 1, min=float(hist["edges"][0]), max=float(hist["edges"][-1])).cpu().numpy()
            hist["n"] = hist.get("n", 0) + int(pen.numel()); hist["over_radius"] = hist.get("over_radius", 0) + int((pen > 0.0175).sum()); hist["max"] = max(hist.get("max", 0.0), float(pen.max()) if pen.numel() else 0.0)
            pd0 = env.buffers["dbg_dist"][:, 0:4]
            hist["plane_n"] = hist.get("plane_n", 0) + int((pd0 < 0).sum()); hist["plane_over_radius"] = hist.get("plane_over_radius", 0) + int((pd0 < -0.0175).sum())
        if k >= 100:
            p = env.buffers["obs_priv"].double()
            s1 += p.mean(0); s2 += (p * p).mean(0); cnt += 1
    env.close()
    mean = (s1 / cnt).cpu().numpy()
    return mean, np.sqrt(np.maximum((s2 / cnt).cpu().numpy() - mean ** 2, 0))


def compare(mean, std, ref_mean, ref_std, blocks=None):
    rows = []
    for name, a, b in (blocks or BLOCKS):
        rows.append(dict(block=name, mean_here=float(mean[a:b].mean()), mean_ref=float(ref_mean[a:b].mean()),
                         std_here=float(std[a:b].mean()), std_ref=float(ref_std[a:b].mean()),
                         mean_dev_sigma=float((np.abs(mean[a:b] - ref_mean[a:b]) / np.maximum(ref_std[a:b], 1e-6)).mean()),
                         std_ratio=float((std[a:b] / np.maximum(ref_std[a:b], 1e-6)).mean())))
    return rows


if __name__ == "__main__":
    level = sys.argv[1] if len(sys.argv) > 1 else "level13"
    use_dr = not (len(sys.argv) > 2 and sys.argv[2] == "nodr")
    stoch = len(sys.argv) > 3 and sys.argv[3] == "stochastic"
    kv = float(sys.argv[4]) if len(sys.argv) > 4 else None
    d = np.load(os.path.join(os.path.dirname(mjcf.__file__), "assets", "policies", "policy177.npz"))
    mean, std = rollout_stats(level, dr=use_dr, stochastic=stoch, kv=kv)
    print(f"{'block':22s} {'mean here':>10s} {'mean ref':>10s} {'|dmean|/sigma':>13s} {'std here':>10s} {'std ref':>10s} {'std ratio':>10s}")
    for r in compare(mean, std, d["mean_priv"], d["std_priv"]):
        print(f"{r['block']:22s} {r['mean_here']:10.4f} {r['mean_ref']:10.4f} {r['mean_dev_sigma']:13.3f} {r['std_here']:10.4f} {r['std_ref']:10.4f} {r['std_ratio']:10.3f}")
