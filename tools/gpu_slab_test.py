"""Does a foot stand ON a box?  policy177 walks onto one wide slab of a given height; prints the steady trunk / foot heights over the slab
against the flat-ground values (the experiment behind the sphere-box fix)."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tools"))
import numpy as np, torch
from phase_guided_terrain_traversal_amd import abi, configs, mjcf
from phase_guided_terrain_traversal_amd.env import Joystick
from phase_guided_terrain_traversal_amd.policy import load_policy
def run(terrain, task="stairs", n=512, steps=400, cmd=(0.5,0,0)):
    kw = {} if terrain is None else {"variant": torch.zeros(n, dtype=torch.int32)}
    env = Joystick(task, configs.training_config(), num_envs=n, terrain=terrain, device="cuda:0", autoreset=False, **kw)
    pi = load_policy("policy177"); env.reset(0)
    S, I = env.buffers["state"], env.buffers["istate"]; c = torch.tensor(cmd, device="cuda:0")
    alive = torch.ones(n, dtype=torch.bool, device="cuda:0"); vx = duty = 0.0
    for k in range(steps):
        S[abi.S_CMD:abi.S_CMD + 3] = c[:, None]; I[abi.I_STEPS_UNTIL_CMD] = 1000000
        obs = env.buffers["obs_state"].clone(); obs[:, -3:] = c
        o, r, d, info = env.step(pi(obs)); alive &= d == 0
        fr = env.buffers["frame"]; w = alive.float(); nw = w.sum().clamp(min=1)
        vx += float((fr[abi.F_LOCAL_LINVEL] * w).sum() / nw); duty += float((fr[abi.F_CONTACT:abi.F_CONTACT + 4].mean(0) * w).sum() / nw)
    z = float(S[2][alive].mean()); env.close()
    return dict(survival=float(alive.float().mean()), vx=vx/steps, duty=duty/steps, base_z=z)
def slab(h, half=6.0, n=1):
    t = np.zeros((1, 100, 10), np.float32); t[0, :, 3] = 1; t[0, :, :3] = [[100+k,100+k,100+k] for k in range(100)]; t[0, :, 7:] = 0.5
    t[0, 0] = [0, 0, h/2, 1, 0, 0, 0, half, half, h/2]
    return t
def tiles(h, size=0.5):   # the same flat surface at height h, but made of many small square boxes (seams everywhere)
    t = np.zeros((1, 100, 10), np.float32); t[0, :, 3] = 1
    k = 0
    for i in range(10):
        for j in range(10):
            t[0, k] = [(i - 4.5) * size, (j - 4.5) * size, h/2, 1, 0, 0, 0, size/2, size/2, h/2]; k += 1
    return t
if __name__ == "__main__":
    print("flat plane      ", run(None, "flat_terrain"))
    print("one slab 6 cm   ", run(slab(0.06)))
    print("tiles 0.5 m 6 cm", run(tiles(0.06)))
    print("tiles 0.3 m 6 cm", run(tiles(0.06, 0.3), steps=250))
