"""-DPGTT_TIME builds: distribution of per-wave kernel ticks of physics_kernel (tail imbalance) on level4."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phase_guided_terrain_traversal_amd import native, configs
from phase_guided_terrain_traversal_amd.env import Joystick
n = 4096
assets = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "phase_guided_terrain_traversal_amd", "assets")
terrain = np.load(os.path.join(assets, "terrains", "level4.npy"))
variant = torch.from_numpy(np.random.default_rng(0).integers(0, terrain.shape[0], n).astype(np.int32))
env = Joystick("stairs", configs.training_config(), num_envs=n, terrain=terrain, device="cuda:0", variant=variant, autoreset=True)
env.reset(seed=1)
L = native.lib(); L.pgtt_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
g = torch.Generator(device="cuda").manual_seed(0)
for k in range(60):
    env.step(torch.tanh(torch.randn(n, 12, device="cuda", generator=g) * 0.6))
    if k >= 50:
        buf = np.zeros(262144, np.float32); L.pgtt_trace_read(buf.ctypes.data, buf.size)
        seg = buf.reshape(4, -1)[(k + 2) % 4]
        t = seg[32:32 + n // 16]; ns = seg[32 + 4096:32 + 4096 + n // 16]
        q = np.percentile(t, [0, 10, 50, 90, 99, 100])
        print(f"step {k}: ticks/wave min {q[0]:.0f} p10 {q[1]:.0f} median {q[2]:.0f} p90 {q[3]:.0f} p99 {q[4]:.0f} max {q[5]:.0f}  mean {t.mean():.0f}  max/mean {t.max() / t.mean():.2f}")
        for v in np.unique(ns):
            sel = ns == v
            print(f"     sum nslots over 4 substeps = {v:.0f}: {sel.sum():4d} waves, mean ticks {t[sel].mean():.0f}")
