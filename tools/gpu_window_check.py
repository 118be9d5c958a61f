import os, sys, time
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
T0 = time.perf_counter()
k = 0
def run(m):
    global k
    for _ in range(m):
        env.step(pool[k % 32]); k += 1
for rep in range(12):
    run(int(sys.argv[1]))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(20)
    torch.cuda.synchronize(); d20 = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    run(300)
    torch.cuda.synchronize(); d300 = (time.perf_counter() - t0) / 300
    print(f"t={time.perf_counter() - T0:6.2f}s  20-step window {d20 * 1e6:6.1f} us/step   300-step window {d300 * 1e6:6.1f} us/step", flush=True)
