"""When can each of the three RECALLED MJX decisions (no output of the reference's own mjx.step exists) matter on the data the reference ships?
For every terrains/level*.npy under closed-loop rollouts of the reference's trained policies (policy177 = PGTT task, policy175 = baseline task; GPU
rollout, >= 100 k env-steps each after the landing), count the env-steps on which
  (a) the max_geom_pairs cut (go2_mjx_feetonly.xml:14-15) ranked with the STALE compiled rbound (the reading used) and with every box's own bounding
      radius pick different 25-pair sets in some substep - and on how many of those the ACTIVE contact set or the state after the step differs at all
      (the fp32 oracle stepped twice from the identical state, switch off / on);
  (b) an ACTIVE foot-box contact has the sphere centre INSIDE the box (dist < -radius) in some substep: where the recalled frame flip of
      _sphere_convex and the frame used here differ;
  (d) [not a model switch, a property of the reference] an ACTIVE foot-box contact has its sphere centre within 1e-6 of the box SURFACE in some substep:
      mjx's normal is normalize(closest point - centre) of a zero-length vector there - rounding noise in fp32 ("the seam attractor");
  (c) the velocity term of the actuator bias (biasprm[2] = -0.5, go2_mjx_feetonly.xml:27), 0.5 |qdot_j|, exceeds 1 % of |actuator force_j| on some joint.
    python tools/gpu_model_switch_relevance.py [n_envs] [steps]          (GPU box; the oracle runs on its host cores)"""
import os, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import oracle
from phase_guided_terrain_traversal_amd import abi, configs, mjcf
from phase_guided_terrain_traversal_amd.env import Joystick
from phase_guided_terrain_traversal_amd.policy import load_policy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
NT = os.cpu_count() or 8
assets = os.path.join(os.path.dirname(mjcf.__file__), "assets", "terrains")
levels = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(assets, "level*.npy")))
print(f"# {n} envs x {steps} control steps (after 10 steps of landing) per (policy, level); oracle on {NT} host threads; levels: {levels}")
print("# policy level | env-steps | cut active | (a) pair sets differ | of those: ACTIVE set differs, state differs (> 1e-6) | (b) centre inside a box | (c) 0.5|qd| > 1 % of |force| ; median over joints of 0.5|qd|/|force| | (d) env-steps with an ACTIVE foot-box contact whose sphere centre is ON the box surface to 1e-6 (the seam attractor)")
tot = np.zeros(8, np.int64)
for policy, method in (("policy177", "pgtt"), ("policy175", "baseline")):
    pi = load_policy(policy)
    for level in levels:
        terrain = np.load(os.path.join(assets, level + ".npy"))
        cfg = configs.training_config(method)
        model = mjcf.load_model("stairs")
        variant = np.random.default_rng(2).integers(0, terrain.shape[0], n).astype(np.int32)
        env = Joystick("stairs", cfg, num_envs=n, terrain=terrain, device="cuda:0", autoreset=True, variant=torch.from_numpy(variant), debug_contacts=True)
        cfg2 = dict(env.config); cs, ms = abi.config_struct(cfg2), abi.model_struct(model)
        hs, hf = (oracle.HostBuffers(n, with_variant=True, method=method) for _ in range(2))
        hs["variant"][:] = variant; hf["variant"][:] = variant
        env.reset(7)
        c = np.zeros(8, np.int64); ratios = []
        for k in range(steps + 10):
            a = pi(env.buffers["obs_state"])
            if k >= 10:
                for hb in (hs, hf):
                    for key in ("state", "istate", "scan_z", "done", "first_state", "first_obs", "ep_metrics"):
                        hb[key][...] = env.buffers[key].cpu().numpy()
            env.step(a)
            if k < 10:
                continue
            act = a.cpu().numpy()
            fl = np.zeros(n, np.int32)
            oracle.step(cs, ms, terrain, hs, act, seed=7, nthreads=NT, flags=fl)
            oracle.step(cs, ms, terrain, hf, act, seed=7, nthreads=NT, fresh_rbound=True)
            sets = lambda hb: [sorted((int(f), int(b)) for (f, b), d in zip(cc, dd) if d < 0 and b != -2) for cc, dd in zip(hb["dbg_contact"].reshape(n, 8, 2), hb["dbg_dist"])]
            sd = np.array([x != y for x, y in zip(sets(hs), sets(hf))])
            qd = np.abs(hs["state"][:37] - hf["state"][:37]).max(0) > 1e-6
            pair = (fl & 1) != 0
            force = np.abs(hs["frame"][abi.F_ACT_FORCE:abi.F_ACT_FORCE + 12]); bias = 0.5 * np.abs(hs["state"][abi.S_QVEL + 6:abi.S_QVEL + 18])
            c += np.array([n, int(((fl & 4) != 0).sum()), int(pair.sum()), int((pair & sd).sum()), int((pair & qd).sum()), int(((fl & 2) != 0).sum()), int((bias > 0.01 * force).any(0).sum()), int(((fl & 8) != 0).sum())])
            assert not (sd & ~pair).any() and not (qd & ~pair).any()          # the switch can only act through the pair set
            ratios.append(np.median(bias / np.maximum(force, 1e-6)))
        env.close()
        tot += c
        print(f"{policy} {level:8s} | {c[0]:7d} | {c[1] / c[0]:7.2%} | {c[2]:6d} ({c[2] / c[0]:.3%}) | {c[3]:5d} {c[4]:5d} | {c[5]:6d} ({c[5] / c[0]:.3%}) | {c[6] / c[0]:7.2%} ; {np.median(ratios):.3f} | seam {c[7]:5d} ({c[7] / c[0]:.3%})", flush=True)
print(f"TOTAL | {tot[0]} env-steps | cut active {tot[1] / tot[0]:.2%} | (a) pair sets differ {tot[2]} ({tot[2] / tot[0]:.4%}), ACTIVE set differs {tot[3]}, state differs {tot[4]} | (b) centre inside {tot[5]} ({tot[5] / tot[0]:.4%}) | (c) {tot[6] / tot[0]:.2%} | (d) a foot centre ON a box surface to 1e-6 (contact normal = normalize(~0)): {tot[7]} ({tot[7] / tot[0]:.4%})")
