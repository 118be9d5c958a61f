"""What the exact many-box pass of the collision stage costs: physics_kernel time on terrains where every standing foot penetrates more boxes than the
per-foot table holds (tests/test_gpu_parity.py: ten graded slabs; seven slabs with one common top face) next to level4, 4096 envs, hex layout.
   usage (GPU box): python tools/gpu_manybox_time.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from phase_guided_terrain_traversal_amd import configs
from phase_guided_terrain_traversal_amd.env import Joystick
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import importlib.util
spec = importlib.util.spec_from_file_location("tp", os.path.join(os.getcwd(), "tests", "test_gpu_parity.py")); tp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tp)
n = 4096
level4 = np.load("phase_guided_terrain_traversal_amd/assets/terrains/level4.npy")
for name, terrain in (("level4", level4), ("graded slabs", tp.stacked_slabs_terrain()), ("equal tops", tp.equal_top_slabs_terrain())):
    variant = torch.from_numpy(np.sort(np.random.default_rng(0).integers(0, terrain.shape[0], n)).astype(np.int32))
    env = Joystick("stairs", configs.training_config(), num_envs=n, terrain=terrain, device="cuda:0", autoreset=True, variant=variant, debug_contacts=True)
    env.reset(seed=1)
    g = torch.Generator(device="cuda").manual_seed(3)
    for k in range(300):
        env.step(torch.tanh(torch.randn(n, 12, device="cuda", generator=g) * 0.3))
    env.enable_timing(4)
    flagged = 0
    for k in range(200):
        env.step(torch.tanh(torch.randn(n, 12, device="cuda", generator=g) * 0.3))
        flagged += int(((env.buffers["dbg_niter"] & 0x10000) != 0).sum())
    torch.cuda.synchronize()
    p, o, m = env.kernel_ms_mean()
    print(f"{name:14s} physics_kernel {1e3 * p:7.1f} us  observe {1e3 * o:5.1f} us  ({m} samples); env-steps that took the many-box pass: {flagged / (200 * n):.2f}")
    env.close()
