"""A/B check of two builds of libpgtt.so: the same seeded rollout, state compared bit for bit at the end.
   usage: python tools/gpu_ab_bitwise.py alt_build/libpgtt_ref.so phase_guided_terrain_traversal_amd/libpgtt.so [steps]
   (each build runs in its own process: PGTT_LIB is read when the library is first loaded)"""
import os, subprocess, sys
import numpy as np
if sys.argv[1] == "--child":
    sys.path.insert(0, os.getcwd())
    import torch
    from phase_guided_terrain_traversal_amd import configs
    from phase_guided_terrain_traversal_amd.env import Joystick
    out, steps = sys.argv[2], int(sys.argv[3])
    res = {}
    for wl in ("level4", "flat"):
        for lay in ("hex", "quad") + (("oct",) if os.environ.get("PGTT_AB_OCT") else ()):
            n = 1024
            terrain = None if wl == "flat" else np.load("phase_guided_terrain_traversal_amd/assets/terrains/level4.npy")
            kw = {} if terrain is None else {"variant": torch.from_numpy(np.random.default_rng(0).integers(0, terrain.shape[0], n).astype(np.int32))}
            env = Joystick("flat_terrain" if wl == "flat" else "stairs", configs.training_config(), num_envs=n, terrain=terrain, device="cuda:0", autoreset=True, layout=lay, **kw)
            env.reset(seed=4)
            g = torch.Generator(device="cuda").manual_seed(7)
            for k in range(steps):
                env.step(torch.tanh(torch.randn(n, 12, device="cuda", generator=g) * 0.6))
            torch.cuda.synchronize()
            res[f"{wl}_{lay}"] = env.buffers["state"].cpu().numpy()
            env.close()
    np.savez(out, **res)
    sys.exit(0)
a, b = sys.argv[1], sys.argv[2]
steps = sys.argv[3] if len(sys.argv) > 3 else "40"
for lib, out in ((a, "/tmp/ab_a.npz"), (b, "/tmp/ab_b.npz")):
    subprocess.run([sys.executable, __file__, "--child", out, steps], check=True, env=dict(os.environ, PGTT_LIB=lib))
A, B = np.load("/tmp/ab_a.npz"), np.load("/tmp/ab_b.npz")
for k in A.files:
    same = np.array_equal(A[k].view(np.uint32), B[k].view(np.uint32))
    d = np.abs(A[k] - B[k])
    print(f"{k:12s} bit-identical: {same}   max |diff| {np.nanmax(d):.3g}   envs differing {int((d.max(0) > 0).sum())} / {A[k].shape[1]}")
    if not same:
        dq = d[:19].max(0); dq = dq[dq > 0]
        if dq.size:
            print("             qpos |diff| of the differing envs: quantiles 10/50/90/100 % =", " ".join(f"{q:.2g}" for q in np.percentile(dq, [10, 50, 90, 100])))
