"""A/B check of two builds of libpgtt.so: the same seeded rollouts, EVERY buffer the step writes compared bit for bit at the end
(state, counters, sensor frame, scan, both observation blocks, reward, done, metrics, episode / interval sums).
   usage: python tools/gpu_ab_bitwise.py alt_build/libpgtt_ref.so phase_guided_terrain_traversal_amd/libpgtt.so [steps]
   (each build runs in its own process: PGTT_LIB is read when the library is first loaded; PGTT_AB_OCT=1 adds the oct layout)
Workloads: level4 and flat ground at 1024 envs per lane layout, and level13 + full DR + AutoReset at a RAGGED env count (1000: the last
workgroup of the tiled observe kernel is partly empty) with a short episode length so that AutoReset-to-first-state is exercised."""
import os, subprocess, sys
import numpy as np
KEYS = ("state", "istate", "frame", "scan_z", "obs_state", "obs_priv", "reward", "done", "metrics", "ep_metrics", "interval_sums")
if sys.argv[1] == "--child":
    sys.path.insert(0, os.getcwd())
    import torch
    from phase_guided_terrain_traversal_amd import configs, mjcf
    from phase_guided_terrain_traversal_amd.env import Joystick
    from phase_guided_terrain_traversal_amd.randomize import domain_randomize
    out, steps = sys.argv[2], int(sys.argv[3])
    res = {}
    A = "phase_guided_terrain_traversal_amd/assets/terrains/"

    def run(tag, task, n, terrain, lay, cfg=None, **kw):
        env = Joystick(task, cfg or configs.training_config(), num_envs=n, terrain=terrain, device="cuda:0", autoreset=True, layout=lay, interval_sums=True, **kw)
        env.reset(seed=4)
        g = torch.Generator(device="cuda").manual_seed(7)
        for k in range(steps):
            env.step(torch.tanh(torch.randn(n, 12, device="cuda", generator=g) * 0.6))
        torch.cuda.synchronize()
        for key in KEYS:
            res[f"{tag}/{key}"] = env.buffers[key].cpu().numpy()
        env.close()
    for wl in ("level4", "flat"):
        for lay in ("hex", "quad") + (("oct",) if os.environ.get("PGTT_AB_OCT") else ()):
            n = 1024
            terrain = None if wl == "flat" else np.load(A + "level4.npy")
            kw = {} if terrain is None else {"variant": torch.from_numpy(np.random.default_rng(0).integers(0, terrain.shape[0], n).astype(np.int32))}
            run(f"{wl}_{lay}", "flat_terrain" if wl == "flat" else "stairs", n, terrain, lay, **kw)
    t13 = np.load(A + "level13.npy"); n = 1000
    dr = domain_randomize(mjcf.load_model("stairs"), n, seed=3, terrain=t13)
    for lay in ("hex",) + (("oct", "quad") if os.environ.get("PGTT_AB_OCT") else ()):          # the DR + terrain kernels of every layout
        run("level13_dr_ragged" + ("" if lay == "hex" else "_" + lay), "stairs", n, t13, lay, cfg=configs.with_overrides(configs.training_config(), episode_length=17),
            variant=torch.from_numpy(dr["variant"]), params=torch.from_numpy(dr["params"]), box_friction=torch.from_numpy(dr["box_friction"]))
    np.savez(out, **res)
    sys.exit(0)
a, b = sys.argv[1], sys.argv[2]
steps = sys.argv[3] if len(sys.argv) > 3 else "40"
for lib, out in ((a, "/tmp/ab_a.npz"), (b, "/tmp/ab_b.npz")):
    subprocess.run([sys.executable, __file__, "--child", out, steps], check=True, env=dict(os.environ, PGTT_LIB=os.path.abspath(lib)))
A, B = np.load("/tmp/ab_a.npz"), np.load("/tmp/ab_b.npz")
allsame = True
for tag in sorted({k.split("/")[0] for k in A.files}):
    diff = []
    for key in KEYS:
        x, y = A[f"{tag}/{key}"], B[f"{tag}/{key}"]
        same = x.shape == y.shape and np.array_equal(x.view(np.uint32), y.view(np.uint32))
        if not same:
            d = np.abs(x.astype(np.float64) - y.astype(np.float64)) if x.shape == y.shape else np.array([np.inf])
            diff.append(f"{key} (max |diff| {np.nanmax(d):.3g}, {int((d > 0).sum())} entries)")
    allsame &= not diff
    print(f"{tag:20s} bit-identical: {not diff}" + ("" if not diff else "   DIFFERENT: " + "; ".join(diff)))
    if diff and f"{tag}/state" in A.files:
        d = np.abs(A[f"{tag}/state"] - B[f"{tag}/state"])
        dq = d[:19].max(0); dq = dq[dq > 0]
        if dq.size:
            print("             qpos |diff| of the differing envs: quantiles 10/50/90/100 % =", " ".join(f"{q:.2g}" for q in np.percentile(dq, [10, 50, 90, 100])))
print("ALL BIT-IDENTICAL" if allsame else "DIFFERENCES FOUND")
