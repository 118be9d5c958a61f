"""Is the 13 x 9 scan fed to the policy in the orientation the reference trained it with?  A reference-trained policy must do best with
the scan as built (rows front -> back, cols left -> right, go2/heightmap.py:34-65) and worse with it mirrored or blanked.
    python tools/gpu_scan_orientation.py [level]      (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phase_guided_terrain_traversal_amd import abi, configs, mjcf
from phase_guided_terrain_traversal_amd.env import Joystick
from phase_guided_terrain_traversal_amd.policy import load_policy


def run(level, mode, n=1024, steps=400, cmd=(0.5, 0.0, 0.0)):
    assets = os.path.join(os.path.dirname(mjcf.__file__), "assets")
    terrain = np.load(os.path.join(assets, "terrains", level + ".npy"))
    kw = {"variant": torch.from_numpy(np.random.default_rng(2).integers(0, terrain.shape[0], n).astype(np.int32))}
    env = Joystick("stairs", configs.training_config(), num_envs=n, terrain=terrain, device="cuda:0", autoreset=False, **kw)
    pi = load_policy("policy177")
    env.reset(0)
    S, I = env.buffers["state"], env.buffers["istate"]
    c = torch.tensor(cmd, device="cuda:0")
    alive = torch.ones(n, dtype=torch.bool, device="cuda:0")
    vx = tilt = duty = 0.0
    for k in range(steps):
        S[abi.S_CMD:abi.S_CMD + 3] = c[:, None]; I[abi.I_STEPS_UNTIL_CMD] = 1000000
        obs = env.buffers["obs_state"].clone()
        obs[:, -3:] = c
        sc = obs[:, 38:38 + 117].reshape(n, 13, 9)
        if mode == "flip_rows": sc = sc.flip(1)
        elif mode == "flip_cols": sc = sc.flip(2)
        elif mode == "flip_both": sc = sc.flip(1).flip(2)
        elif mode == "blank": sc = torch.zeros_like(sc)
        obs[:, 38:38 + 117] = sc.reshape(n, 117)
        o, r, d, info = env.step(pi(obs))
        alive &= d == 0
        fr = env.buffers["frame"]; w = alive.float(); nw = w.sum().clamp(min=1)
        vx += float((fr[abi.F_LOCAL_LINVEL] * w).sum() / nw); tilt += float(((1 - fr[abi.F_UPVECTOR + 2]) * w).sum() / nw)
        duty += float((fr[abi.F_CONTACT:abi.F_CONTACT + 4].mean(0) * w).sum() / nw)
    x = float(S[0][alive].mean()) if alive.any() else 0.0
    env.close()
    return dict(mode=mode, survival=float(alive.float().mean()), vx=vx / steps, tilt=tilt / steps, duty=duty / steps)


if __name__ == "__main__":
    level = sys.argv[1] if len(sys.argv) > 1 else "level4"
    for mode in ("as_built", "flip_rows", "flip_cols", "flip_both", "blank"):
        print(level, run(level, mode))
