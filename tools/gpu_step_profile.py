import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from phase_guided_terrain_traversal_amd import configs
from phase_guided_terrain_traversal_amd.env import Joystick
n = 4096
terrain = np.load("phase_guided_terrain_traversal_amd/assets/terrains/level4.npy")
variant = torch.from_numpy(np.sort(np.random.default_rng(0).integers(0, terrain.shape[0], n)).astype(np.int32))
env = Joystick("stairs", configs.training_config(), num_envs=n, terrain=terrain, device="cuda:0", variant=variant, autoreset=True, interval_sums=True)
env.reset(seed=0)
g = torch.Generator(device="cuda").manual_seed(1)
pool = [torch.tanh(torch.randn(n, 12, generator=g, device="cuda") * 0.6) for _ in range(32)]
import time
for rep in range(2):
    env.reset(seed=0)
    out = []
    for c in range(40):
        env.enable_timing(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(25):
            env.step(pool[(c * 25 + k) % 32])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        p, o, cnt = env.kernel_ms_mean()
        out.append((c * 25, round(p * 1000, 1), round(o * 1000, 1), round(dt / 25 * 1e6, 1), round(float(env.buffers["done"].mean()), 4)))
    print("rep", rep, "(first step, physics us, observe us, wall us/step, done frac):")
    print(" ".join(str(x) for x in out))
