"""-DPGTT_TIME=<env> builds: shader-clock ticks at the phase boundaries of observe_kernel for the wave of that env.
   usage: PGTT_LIB=alt_build/libpgtt_time.so python tools/gpu_observe_time.py [level4|flat]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phase_guided_terrain_traversal_amd import native, configs
from phase_guided_terrain_traversal_amd.env import Joystick
wl = sys.argv[1] if len(sys.argv) > 1 else "level4"
n = 4096
assets = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "phase_guided_terrain_traversal_amd", "assets")
terrain = None if wl == "flat" else np.load(os.path.join(assets, "terrains", "level4.npy"))
variant = None if terrain is None else torch.from_numpy(np.random.default_rng(0).integers(0, terrain.shape[0], n).astype(np.int32))
env = Joystick("flat_terrain" if wl == "flat" else "stairs", configs.training_config(), num_envs=n, terrain=terrain, device="cuda:0", variant=variant, autoreset=True)
env.reset(seed=1)
L = native.lib(); L.pgtt_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
g = torch.Generator(device="cuda").manual_seed(0)
names = ["state / frame rows in LDS", "scan grid, cull, rays", "scan stores, quadrant statistics", "per-env scalars", "observation rows (noise)", "rewards / bookkeeping", "episode metrics", "state stores", ]
acc = np.zeros(8); cnt = 0
for k in range(60):
    env.step(torch.tanh(torch.randn(n, 12, device="cuda", generator=g) * 0.6))
    if k >= 20:
        buf = np.zeros(262144, np.float32); L.pgtt_trace_read(buf.ctypes.data, buf.size)
        t = buf[60000:60008].astype(np.float64)
        acc += np.diff(np.concatenate([[0.0], t])); cnt += 1
print(f"{wl}: observe wave of env 0, mean over {cnt} steps, total {acc.sum() / cnt:.0f} ticks")
for nme, v in zip(names, acc / cnt):
    print(f"  {nme:34s} {v:9.0f} ticks  {100 * v / acc.sum() * cnt:5.1f} %")
allw = np.diff(np.concatenate([np.zeros((512, 1)), buf[61000:61000 + 8 * 512].reshape(512, 8).astype(np.float64)], 1), axis=1)
print("last launch, the waves of blocks 0..511: ticks per phase, quantiles 10 / 50 / 90 / 100 %")
for i, nme in enumerate(names):
    print(f"  {nme:34s}", " ".join(f"{q:7.0f}" for q in np.percentile(allw[:, i], [10, 50, 90, 100])))
print(f"  {'whole wave':34s}", " ".join(f"{q:7.0f}" for q in np.percentile(allw.sum(1), [10, 50, 90, 100])))
slow = np.argsort(allw.sum(1))[-8:]
print("  the 8 slowest of them, ticks per phase:"); 
for w in slow[::-1]:
    print(f"    block {w:4d}: " + " ".join(f"{v:6.0f}" for v in allw[w]) + f"   sum {allw[w].sum():6.0f}")
tw = buf[40960:40960 + 4 * 4096].view(np.uint32).reshape(-1, 4)
tw = tw[tw[:, 3] != 0]
hw, xcc, t0, t1 = tw[:, 0], tw[:, 1] & 0xF, tw[:, 2].astype(np.int64), tw[:, 3].astype(np.int64)
key = (xcc.astype(np.int64) << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | (((hw >> 8) & 15) << 4) | ((hw >> 4) & 3)
import collections
per_simd = collections.Counter(key.tolist())
base = t0.min(); st = (t0 - base) * 10; en = (t1 - base) * 10
print(f"last launch: {len(tw)} waves on {len(set((key >> 4).tolist()))} CUs / {len(per_simd)} SIMDs, waves per SIMD {sorted(collections.Counter(per_simd.values()).items())}")
print("  start ns quantiles 10/50/90/100 %:", " ".join(f"{q:.0f}" for q in np.percentile(st, [10, 50, 90, 100])), "  end ns:", " ".join(f"{q:.0f}" for q in np.percentile(en, [10, 50, 90, 100])), "  duration ns:", " ".join(f"{q:.0f}" for q in np.percentile(en - st, [10, 50, 90, 100])))
