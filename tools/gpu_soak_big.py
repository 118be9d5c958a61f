import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from phase_guided_terrain_traversal_amd import configs, abi, mjcf
from phase_guided_terrain_traversal_amd.env import Joystick
from phase_guided_terrain_traversal_amd.randomize import domain_randomize
assets = "phase_guided_terrain_traversal_amd/assets/terrains"
for level, n in (("level13", 16384), ("level10", 32768)):
    terrain = np.load(f"{assets}/{level}.npy")
    out = domain_randomize(mjcf.load_model("stairs"), n, seed=5, terrain=terrain)
    env = Joystick("stairs", configs.training_config(), num_envs=n, terrain=terrain, device="cuda:0", autoreset=True,
                   variant=torch.from_numpy(out["variant"]), params=torch.from_numpy(out["params"]), box_friction=torch.from_numpy(out["box_friction"]))
    env.reset(seed=9)
    g = torch.Generator(device="cuda").manual_seed(3)
    bad = 0; dones = 0.0
    for k in range(2000):
        obs, r, d, info = env.step(torch.tanh(torch.randn(n, 12, device="cuda", generator=g) * 0.8))
        if k % 100 == 0:
            dones += float(d.sum())
            bad += sum(0 if torch.isfinite(env.buffers[key]).all() else 1 for key in ("state", "obs_state", "obs_priv", "reward", "metrics", "frame"))
    S = env.buffers["state"]
    print(f"auto (quad) {level} dr n={n}: nonfinite checks failed: {bad}  |qpos z| max {float(S[2].abs().max()):.3f}  |qvel| max {float(S[19:37].abs().max()):.2f}  dones/100-step sample {dones}")
    env.close()
