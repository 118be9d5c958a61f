"""policy177 privileged-observation statistics (stochastic closed-loop rollout on level13 with DR) per lane layout, against the policy's own
normaliser: the distribution-level pin of DESIGN.md 2 holds in every layout."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import gpu_policy_stats as G
from phase_guided_terrain_traversal_amd.policy import load_policy
pi = load_policy("policy177")
ref_m = np.load("phase_guided_terrain_traversal_amd/assets/policies/policy177.npz")
for lay, n in (("hex", 4096), ("oct", 8192)):
    out = G.rollout_stats("level13", n=n, steps=600, stochastic=True, layout=lay)
    mean, std = out[0], out[1]
    rows = G.compare(mean, std, ref_m["mean_priv"], ref_m["std_priv"])
    print(lay, n, "survival/extra:", out[2:] if len(out) > 2 else "")
    for r in rows: print("   %-18s mean %.3f (ref %.3f)  std ratio %.2f  dev %.2f sigma" % (r["block"], r["mean_here"], r["mean_ref"], r["std_ratio"], r["mean_dev_sigma"]))
