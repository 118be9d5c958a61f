"""-DPGTT_TIME=<env> builds only: shader-clock ticks per stage of physics_kernel for the wave that owns that env.
   usage: PGTT_LIB=alt_build/libpgtt_time.so python tools/gpu_stage_time.py [level4|flat] [num_envs]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phase_guided_terrain_traversal_amd import native, configs
from phase_guided_terrain_traversal_amd.env import Joystick
wl = sys.argv[1] if len(sys.argv) > 1 else "level4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
assets = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "phase_guided_terrain_traversal_amd", "assets")
terrain = None if wl == "flat" else np.load(os.path.join(assets, "terrains", "level4.npy"))
variant = None if terrain is None else torch.from_numpy(np.random.default_rng(0).integers(0, terrain.shape[0], n).astype(np.int32))
env = Joystick("flat_terrain" if wl == "flat" else "stairs", configs.training_config(), num_envs=n, terrain=terrain, device="cuda:0", variant=variant, autoreset=True)
env.reset(seed=1)
L = native.lib(); L.pgtt_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
g = torch.Generator(device="cuda").manual_seed(0)
names = ["position", "velocity", "constraint", "sensors", "solver init x2-3", "first gradient", "line search", "update_constraint", "update_gradient", "rest/integrate", "iterations", "  c: limits+plane", "  c: AABB pass 1a", "  c: narrow 1b", "  c: table+count 2a/2b", "  launch prologue", "  c: contact records", "-"]
acc = np.zeros(32)
for k in range(60):
    act = torch.tanh(torch.randn(n, 12, device="cuda", generator=g) * 0.6)
    env.step(act)
    if k >= 20:
        buf = np.zeros(262144, np.float32); L.pgtt_trace_read(buf.ctypes.data, buf.size)
        seg = buf.reshape(4, -1)[:, :32]
        acc += seg[(k + 2) % 4]           # reset issued launches 0 and 1; step k is launch k + 2, segment = launch % 4
tot = acc[:10].sum()
print(f"{wl} n={n}: mean Newton iterations per control step {acc[10] / 40:.2f} (4 substeps); line-search rounds per step executed by the wave {acc[18] / 40:.1f}, needed by env 0 {acc[19] / 40:.1f}")
print(f"  whole kernel, this wave: {acc[20] / 40:.0f} shader ticks = {acc[21] / 40 * 10:.0f} ns -> shader clock {acc[20] / max(acc[21], 1) * 0.1:.2f} GHz")
for i in list(range(10)) + list(range(11, 17)):
    print(f"  {names[i]:22s} {acc[i] / 40:12.0f} ticks/step  {100 * acc[i] / tot:5.1f} %")
for i, nm in zip(range(24, 29), ("  ls: set-up (M s, J s, rows, Gauss terms)", "  ls: two initial points", "  ls: bracketing rounds", "  ls: final costs", "  ls: update")):
    print(f"  {nm:42s} {acc[i] / 40:12.0f} ticks/step  {100 * acc[i] / tot:5.1f} %")
print(f"  broad-phase count of pass 2a, env 0 (pairs at least as close as the farthest candidate; exact ranks are needed above max_geom_pairs = 25): sum over the passes of a control step {acc[29] / 40:.1f}; substeps per control step in which the wave carries the proof over instead of counting: {acc[30] / 40:.2f} of 4")
