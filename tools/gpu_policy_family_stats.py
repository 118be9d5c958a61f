"""The distribution-level pin of the un-vendored physics (tests/test_gpu_policy.py) on MORE of the reference's trained policies: every pickle
of policy_folder/ carries the observation normaliser of its training run - mean / std of all 215 (206) privileged-observation rows over 10^8 - 10^9 samples
of the reference's own simulator, accumulated over its curriculum stages.  Each policy is rolled out here (full randomize.py DR, observation noise, the
task's own command sampling, AutoReset, actions SAMPLED from its head) on several level files - which level files a run saw is not recorded, the `scan`
block tells - and the block statistics are printed next to the normaliser's, with the actuator-bias velocity term kept (-0.5) and cleared (0).
    python tools/gpu_policy_family_stats.py [policyNNN ...]          (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from gpu_policy_stats import BLOCKS, BLOCKS_BASELINE, compare, rollout_stats
from phase_guided_terrain_traversal_amd import mjcf
names = sys.argv[1:] or ["policy162", "policy172", "policy182", "policy185", "policy174", "policy177", "policy175"]
KEYS = ("gyro", "gravity", "joint pos - default", "joint vel", "scan - min", "last action", "local linvel", "accelerometer", "actuator force", "last contact", "feet linvel", "feet air time")
print("policy level kv | " + " | ".join(f"{k[:12]:>12s}" for k in KEYS) + "   (std here / std of the normaliser; 'last contact' and 'scan - min': mean here / mean ref)")
for name in names:
    d = np.load(os.path.join(os.path.dirname(mjcf.__file__), "assets", "policies", name + ".npz"))
    baseline = d["mean"].shape[0] == 162
    for level in ("level4", "level7", "level10", "level13"):
        for kv in (-0.5, 0.0):
            mean, std = rollout_stats(level, n=1024, steps=500, stochastic=True, kv=kv, policy=name, method="baseline" if baseline else "pgtt")
            rows = {r["block"]: r for r in compare(mean, std, d["mean_priv"], d["std_priv"], BLOCKS_BASELINE if baseline else BLOCKS)}
            cell = lambda k: (rows[k]["mean_here"] / rows[k]["mean_ref"]) if k in ("last contact", "scan - min") else rows[k]["std_ratio"]
            print(f"{name} {level:7s} {kv:4.1f} | " + " | ".join(f"{cell(k):12.3f}" for k in KEYS), flush=True)
