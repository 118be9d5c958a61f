import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np, torch
from phase_guided_terrain_traversal_amd import abi, configs, mjcf
from phase_guided_terrain_traversal_amd.env import Joystick
from phase_guided_terrain_traversal_amd.policy import load_policy
from gpu_slab_test import slab
n = 256; h = 0.06
env = Joystick("stairs", configs.training_config(), num_envs=n, terrain=slab(h), device="cuda:0", autoreset=False, variant=torch.zeros(n, dtype=torch.int32), debug_contacts=True)
pi = load_policy("policy177"); env.reset(0)
S, I = env.buffers["state"], env.buffers["istate"]; c = torch.tensor((0.5, 0.0, 0.0), device="cuda:0")
for k in range(200):
    S[abi.S_CMD:abi.S_CMD + 3] = c[:, None]; I[abi.I_STEPS_UNTIL_CMD] = 1000000
    obs = env.buffers["obs_state"].clone(); obs[:, -3:] = c
    env.step(pi(obs))
    if k % 10 == 0 or k < 12:
        fr = env.buffers["frame"]; fz = fr[abi.F_FOOT_SITE_Z:abi.F_FOOT_SITE_Z + 4]
        dd = env.buffers["dbg_dist"]; dc = env.buffers["dbg_contact"].reshape(n, 8, 2)
        below = (fz < h + 0.0175 - 0.0175).float().mean()          # foot centre below the slab surface
        print(k, "base z %.3f" % float(S[2].mean()), "foot z min %.3f mean %.3f" % (float(fz.min()), float(fz.mean())), "frac feet with centre below the surface %.3f" % float(below),
              "box contact dist min %.4f" % float(dd[:, 4:].min()), "n box contacts/env %.2f" % float((dc[:, 4:, 1] >= 0).float().sum(1).mean()), "plane dist min %.3f" % float(dd[:, :4].min()))
