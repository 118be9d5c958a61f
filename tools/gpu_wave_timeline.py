"""-DPGTT_TIME builds: where and when the waves of one physics_kernel launch ran (HW_ID / XCC_ID placement, start and end on the
   constant 100 MHz clock).  usage: PGTT_LIB=alt_build/libpgtt_time.so python tools/gpu_wave_timeline.py [num_envs]"""
import os, sys, ctypes, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phase_guided_terrain_traversal_amd import native, configs
from phase_guided_terrain_traversal_amd.env import Joystick
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
assets = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "phase_guided_terrain_traversal_amd", "assets")
terrain = np.load(os.path.join(assets, "terrains", "level4.npy"))
variant = torch.from_numpy(np.random.default_rng(0).integers(0, terrain.shape[0], n).astype(np.int32))
env = Joystick("stairs", configs.training_config(), num_envs=n, terrain=terrain, device="cuda:0", variant=variant, autoreset=True)
env.reset(seed=1)
L = native.lib(); L.pgtt_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
g = torch.Generator(device="cuda").manual_seed(0)
show = {52, 53, 100, 150, 200, 250, 300, 329}
for k in range(330):
    env.step(torch.tanh(torch.randn(n, 12, device="cuda", generator=g) * 0.6))
    if k in show:
        buf = np.zeros(262144, np.float32); L.pgtt_trace_read(buf.ctypes.data, buf.size)
        seg = buf.reshape(4, -1)[(k + 2) % 4]
        nw = min(1024, env.num_blocks if hasattr(env, "num_blocks") else 1024)
        tw = seg[32 + 8192:32 + 8192 + 4 * nw].view(np.uint32).reshape(-1, 4)
        tw = tw[tw[:, 3] != 0]
        hw, xcc, t0, t1 = tw[:, 0], tw[:, 1] & 0xF, tw[:, 2].astype(np.int64), tw[:, 3].astype(np.int64)
        simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
        base = t0.min(); s = (t0 - base) * 10; e = (t1 - base) * 10
        key = (xcc.astype(np.int64) << 16) | (se << 12) | (sh << 8) | (cu << 4) | simd
        per_simd = collections.Counter(key.tolist()); per_cu = collections.Counter((key >> 4).tolist())
        print(f"step {k}: {len(tw)} waves on {len(per_cu)} CUs / {len(per_simd)} SIMDs; waves per SIMD histogram {sorted(collections.Counter(per_simd.values()).items())}; per CU {sorted(collections.Counter(per_cu.values()).items())}")
        print(f"   start ns: p50 {np.percentile(s, 50):.0f} p90 {np.percentile(s, 90):.0f} max {s.max():.0f};  end ns: p10 {np.percentile(e, 10):.0f} p50 {np.percentile(e, 50):.0f} p90 {np.percentile(e, 90):.0f} max {e.max():.0f};  duration ns: p50 {np.percentile(e - s, 50):.0f} max {(e - s).max():.0f}")
        multi = np.array([per_simd[x] for x in key.tolist()])
        for c in sorted(set(multi.tolist())):
            print(f"   waves sharing their SIMD with {c - 1} others: {np.sum(multi == c):4d}, mean duration {np.mean((e - s)[multi == c]):.0f} ns")
        st = seg[16384:16384 + 24 * 1024].reshape(1024, 24)[:len(tw)]
        names = ["position", "velocity", "constraint", "sensors", "solver init", "first gradient", "line search", "update_constraint", "update_gradient", "rest", "newton trips", "c:limits+plane", "c:AABB", "c:narrow", "c:table+count", "prologue", "c:records", "sum nslots", "ls rounds", "ls needed"]
        order = np.argsort(-(e - s))
        med = np.median(st, axis=0)
        print("   stage ticks: " + "  ".join(f"{names[i]}={med[i]:.0f}" for i in range(20)) + "   <- median wave")
        for w in order[:3]:
            print(f"   wave {w} duration {(e - s)[w]} ns: " + "  ".join(f"{names[i]}={st[w, i]:.0f}" for i in range(20)))
