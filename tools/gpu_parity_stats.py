"""Error statistics of the HIP step against the oracle (fp32, and the oracle's fp32 against its fp64 build) over the env-steps of a
rollout, per workload: the numbers the tolerances of tests/test_gpu_parity.py are set from.
    python tools/gpu_parity_stats.py [out.json]          (GPU box)"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as P
from oracle import oracle
from phase_guided_terrain_traversal_amd import abi

def collect(task, n, terrain, steps, dr=False, autoreset=False, method="pgtt", ctrl_dt=None):
    env, hb, cs, ms = P.make_pair(task, n, terrain, dr=dr, autoreset=autoreset, method=method, ctrl_dt=ctrl_dt)
    h64 = oracle.HostBuffers(n, with_params=dr, with_variant="variant" in hb.arrays, with_box_friction="box_friction" in hb.arrays, method=method)
    for k in ("params", "variant", "box_friction"):
        if k in hb.arrays:
            h64[k][...] = hb[k]
    env.reset(3); oracle.reset(cs, ms, terrain, hb, seed=3, nthreads=16)
    rng = np.random.default_rng(1)
    EG, EF, NI, FM, SM, RS = [], [], [], [], [], []
    for k in range(steps):
        P.sync_to_host(env, hb, h64)
        act = np.tanh(rng.normal(size=(n, 12)) * 0.6).astype(np.float32)
        env.step(torch.from_numpy(act).cuda())
        r32, r64 = np.zeros(n), np.zeros(n)
        oracle.step(cs, ms, terrain, hb, act, seed=3, nthreads=16, resid=r32)
        oracle.step(cs, ms, terrain, h64, act, seed=3, nthreads=16, fp64=True, resid=r64)
        RS.append(np.stack([r32, r64]))
        torch.cuda.synchronize()
        g = {kk: v.cpu().numpy() for kk, v in env.buffers.items()}
        EG.append(P.per_env_errors(g, hb)); EF.append(P.per_env_errors(hb.arrays, h64))
        NI.append(np.stack([g["dbg_niter"] & 0xFFFF, hb["dbg_niter"], h64["dbg_niter"]]))
        FM.append((g["frame"][abi.F_CONTACT:abi.F_CONTACT + 4] != hb["frame"][abi.F_CONTACT:abi.F_CONTACT + 4]).any(0))
        ga, ha = P.active_sets(g["dbg_contact"], g["dbg_dist"]), P.active_sets(hb["dbg_contact"], hb["dbg_dist"])
        SM.append(np.array([a != b for a, b in zip(ga, ha)]))
    env.close()
    cat = lambda L, key: np.concatenate([d[key] for d in L])
    eg = {k: cat(EG, k) for k in EG[0]}; ef = {k: cat(EF, k) for k in EF[0]}
    ni = np.concatenate(NI, 1); fm = np.concatenate(FM); sm = np.concatenate(SM)
    return eg, ef, ni, fm, sm, np.concatenate(RS, 1)

def pct(a):
    return {p: float(np.percentile(a, q)) for p, q in (("p50", 50), ("p90", 90), ("p99", 99), ("p999", 99.9), ("max", 100))} if a.size else {}

RAW = {}
def summarise(name, eg, ef, ni, fm, sm, rs):
    for k in eg: RAW[f'{name}/eg_{k}'] = eg[k].astype(np.float32); RAW[f'{name}/ef_{k}'] = ef[k].astype(np.float32)
    RAW[f'{name}/ni'] = ni; RAW[f'{name}/fm'] = fm; RAW[f'{name}/sm'] = sm; RAW[f'{name}/rs'] = rs
    agree = (ef["qpos"] < 1e-5) & (ef["qvel"] < 1e-3)
    conv = (ni < 5).all(0)
    out = {"env_steps": int(agree.size), "agree_frac": float(agree.mean()), "conv_frac": float(conv.mean()), "agree_and_conv_frac": float((agree & conv).mean()),
           "flag_mismatch_all": int(fm.sum()), "set_mismatch_all": int(sm.sum()), "flag_mismatch_agree": int((fm & agree).sum()), "set_mismatch_agree": int((sm & agree).sum())}
    for key in eg:
        out[f"gpu_{key}_agree"] = pct(eg[key][agree]); out[f"gpu_{key}_all"] = pct(eg[key]); out[f"fp_{key}_all"] = pct(ef[key]); out[f"fp_{key}_agree"] = pct(ef[key][agree])
    for key, tols in (("qpos", (1e-5, 1e-4, 1e-3)), ("qvel", (1e-3, 5e-3, 2e-2)), ("warm", (1e-3, 1e-2, 1e-1)), ("obs", (1e-3, 2e-2)), ("priv", (1e-3, 2e-2)), ("frame", (1e-3, 2e-2))):
        out[f"viol_{key}_agree"] = {str(t): float((eg[key][agree] > t).mean()) for t in tols}
    for fname, filt in (("agree", agree), ("r64<1e-6", rs[1] < 1e-6), ("r64<1e-6&agree", (rs[1] < 1e-6) & agree), ("r64<1e-6&r32<1e-3", (rs[1] < 1e-6) & (rs[0] < 1e-3)),
                        ("ni64<5", ni[2] < 5), ("old", agree & conv)):
        if filt.sum() == 0: continue
        out["filter_" + fname] = dict(frac=float(filt.mean()), qpos_viol_1e4=float((eg["qpos"][filt] > 1e-4).mean()), qvel_viol_1e3=float((eg["qvel"][filt] > 1e-3).mean()),
                                      qvel_viol_5e3=float((eg["qvel"][filt] > 5e-3).mean()), warm_viol_1e2=float((eg["warm"][filt] > 1e-2).mean()),
                                      flag=int((fm & filt).sum()), sets=int((sm & filt).sum()))
        print("    filter", fname, out["filter_" + fname])
    print(name, json.dumps({k: v for k, v in out.items() if not isinstance(v, dict)}))
    for key in ("qpos", "qvel", "warm", "obs", "frame"):
        print("   ", key, "gpu|agree", out[f"gpu_{key}_agree"], "viol", out[f"viol_{key}_agree"])
        print("   ", key, "gpu|all  ", out[f"gpu_{key}_all"], " fp|all", out[f"fp_{key}_all"])
    return out

def ratios(lay="hex"):
    """p50 / p90 / p99 of the GPU-vs-oracle error next to the oracle's own fp32-vs-fp64 error, all env-steps and W, for whichever build PGTT_LIB names:
    the column of profiles/archive/r03_parity_p90.txt (product, and `make PRECISE_DIV=1`)"""
    from phase_guided_terrain_traversal_amd import native
    A = os.path.join(ROOT, "phase_guided_terrain_traversal_amd", "assets", "terrains")
    P.EXEC["layout"] = lay
    print("build:", os.path.basename(native.LIB_PATH), "layout", lay)
    nenv = int(os.environ.get("PGTT_STATS_ENVS", "512"))
    for name, args in (("flat", ("flat_terrain", nenv, None, 60)), ("level4", ("stairs", nenv, np.load(os.path.join(A, "level4.npy")), 60))):
        eg, ef, ni, fm, sm, rs = collect(*args)
        W = (rs[1] < 1e-6) & (ef["qpos"] < 1e-5) & (ef["qvel"] < 1e-3)
        print(f"  {name:7s} W = {W.mean():.3%} of {W.size} env-steps; on W: qpos > 1e-4 on {(eg['qpos'][W] > 1e-4).mean():.4%}, qvel > 5e-3 on {(eg['qvel'][W] > 5e-3).mean():.4%}, warm > 1e-2 on {(eg['warm'][W] > 1e-2).mean():.4%}, flags {int((fm & W).sum())}, sets {int((sm & W).sum())}")
        for key in ("qpos", "qvel", "obs", "frame"):
            for tag, f in (("all", np.ones_like(W)), ("W", W)):
                g, o = np.percentile(eg[key][f], [50, 90, 99, 99.9]), np.percentile(ef[key][f], [50, 90, 99, 99.9])
                print(f"  {name:7s} {key:5s} {tag:3s} gpu p50/p90/p99/p99.9 {g[0]:.2e} {g[1]:.2e} {g[2]:.2e} {g[3]:.2e} | oracle f32-vs-f64 {o[0]:.2e} {o[1]:.2e} {o[2]:.2e} {o[3]:.2e} | ratio p50 {g[0] / max(o[0], 1e-30):.2f} p90 {g[1] / max(o[1], 1e-30):.2f} p99 {g[2] / max(o[2], 1e-30):.2f}")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--ratios":
    ratios(sys.argv[2] if len(sys.argv) > 2 else "hex")
    sys.exit(0)
if __name__ == "__main__":
    A = os.path.join(ROOT, "phase_guided_terrain_traversal_amd", "assets", "terrains")
    res = {}
    for lay in ("hex", "quad"):
        P.EXEC["layout"] = lay
        res[f"flat_{lay}"] = summarise(f"flat_{lay}", *collect("flat_terrain", 512, None, 60))
        res[f"level4_{lay}"] = summarise(f"level4_{lay}", *collect("stairs", 512, np.load(os.path.join(A, "level4.npy")), 60))
        res[f"level13_dr_{lay}"] = summarise(f"level13_dr_{lay}", *collect("stairs", 256, np.load(os.path.join(A, "level13.npy")), 60, dr=True, autoreset=True))
    P.EXEC["layout"] = "hex"
    res["flat_1substep"] = summarise("flat_1substep", *collect("flat_terrain", 512, None, 60, ctrl_dt=0.005))
    res["level4_1substep"] = summarise("level4_1substep", *collect("stairs", 512, np.load(os.path.join(A, "level4.npy")), 60, ctrl_dt=0.005))
    if len(sys.argv) > 1:
        np.savez_compressed(sys.argv[1].replace(".json", "_raw.npz"), **RAW)
        json.dump(res, open(sys.argv[1], "w"), indent=1)
