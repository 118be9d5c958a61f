"""-DPGTT_OBS_STOP builds: dynamic instruction counts of observe_kernel per phase.  The observe wave leaves at phase boundary k when the
test-hook integer scan_preset is 100 + k (a RUN-TIME value, so nothing before the boundary is optimised away); run under
   rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES
for k = 0..7 and without a stop, the differences of consecutive per-wave means are the phases' counts (tools/observe_instr.sh).
   usage: PGTT_LIB=alt_build/libpgtt_obsstop.so python tools/gpu_observe_instr.py <k | -1>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phase_guided_terrain_traversal_amd import configs, native
from phase_guided_terrain_traversal_amd.env import Joystick
k = int(sys.argv[1]); n = 4096
assets = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "phase_guided_terrain_traversal_amd", "assets")
terrain = np.load(os.path.join(assets, "terrains", "level4.npy"))
variant = torch.from_numpy(np.sort(np.random.default_rng(0).integers(0, terrain.shape[0], n)).astype(np.int32))
env = Joystick("stairs", configs.training_config(), num_envs=n, terrain=terrain, device="cuda:0", variant=variant, autoreset=True, test_hooks=True)
env.reset(seed=1)
g = torch.Generator(device="cuda").manual_seed(0)
pool = [torch.tanh(torch.randn(n, 12, generator=g, device="cuda") * 0.6) for _ in range(8)]
for i in range(20): env.step(pool[i % 8])          # a rollout state with feet on the stairs
if k >= 0:
    native.check(env._lib.pgtt_set_test_overrides(env._h, float("nan"), 100 + k))
for i in range(10): env.step(pool[i % 8])
torch.cuda.synchronize()
