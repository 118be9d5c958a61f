"""Do an env's bits depend on the company it keeps in its wave?  One control step of the same envs from the same state, once as the whole batch and once
as a batch that holds every third env only (other wave mates), per lane layout: number of envs whose state rows differ.
    python tools/gpu_wave_company.py          (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from phase_guided_terrain_traversal_amd import abi, configs, mjcf
from phase_guided_terrain_traversal_amd.env import Joystick
terrain = np.load(os.path.join(os.path.dirname(mjcf.__file__), "assets", "terrains", "level4.npy"))
n = 1536
variant = np.random.default_rng(2).integers(0, terrain.shape[0], n).astype(np.int32)
sub = np.arange(0, n, 3)
for task, ter in (("stairs", terrain), ("flat_terrain", None)):
    for layout in ("quad", "oct", "hex"):
        mk = lambda idx: Joystick(task, configs.training_config(), num_envs=len(idx), terrain=ter, device="cuda:0", layout=layout,
                                  **({"variant": torch.from_numpy(variant[idx])} if ter is not None else {}))
        A, B = mk(np.arange(n)), mk(sub)
        A.reset(3)
        g = torch.Generator(device="cuda").manual_seed(0)
        diff = 0; steps = 40
        for k in range(steps):
            a = torch.tanh(torch.randn(n, 12, generator=g, device="cuda") * 0.6)
            for key in ("state", "istate", "scan_z"):
                B.buffers[key].copy_(A.buffers[key][..., torch.from_numpy(sub).cuda()] if key != "scan_z" else A.buffers[key][torch.from_numpy(sub).cuda()])
            A.step(a); B.step(a[torch.from_numpy(sub).cuda()].contiguous())
            torch.cuda.synchronize()
            diff += int((A.buffers["state"][:55, torch.from_numpy(sub).cuda()] != B.buffers["state"][:55]).any(0).sum())
        print(f"{task:13s} {layout:4s}: {diff} of {steps * len(sub)} env-steps differ in qpos / qvel / warm start between the whole batch and the every-third-env batch")
        A.close(); B.close()
