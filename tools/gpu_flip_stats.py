"""The sphere-box decision as numbers: policy177 rolled out stochastically on a level file with full DR in WHICHEVER
libpgtt build PGTT_LIB points at (default: the product), and on one 6 cm slab -
  * contact duty, mean air time and the tilt / joint-velocity spreads against the policy's own normaliser (443 M samples of the reference's simulator),
  * base height gained on the slab (a solid top gives + 6 cm),
  * the histogram of the penetration depth max(0, -dist) of every box contact met (what fraction exceeds the foot radius 17.5 mm, the depth at
    which the literal recalled `_sphere_convex` flips the contact frame).
Prints one JSON object.  tests/test_gpu_policy.py runs it once per build and asserts that the product passes and the -DPGTT_SPHERE_CONVEX_FLIP build fails.
    python tools/gpu_flip_stats.py [level13] [n] [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np


def main():
    from gpu_policy_stats import compare, rollout_stats
    from gpu_slab_test import run, slab
    from phase_guided_terrain_traversal_amd import mjcf, native
    level = sys.argv[1] if len(sys.argv) > 1 else "level13"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 500
    d = np.load(os.path.join(os.path.dirname(mjcf.__file__), "assets", "policies", "policy177.npz"))
    edges = np.linspace(0.0, 0.04, 33)          # 1.25 mm bins up to 40 mm
    hist = {"edges": edges}
    mean, std = rollout_stats(level, n=n, steps=steps, stochastic=True, hist=hist)
    rows = {r["block"]: r for r in compare(mean, std, d["mean_priv"], d["std_priv"])}
    flat, one = run(None, "flat_terrain", n=256, steps=300), run(slab(0.06), n=256, steps=300)
    counts = np.asarray(hist["counts"], dtype=np.float64)
    out = {"lib": os.path.basename(native.LIB_PATH), "level": level, "envs": n, "steps": steps,
           "contact_duty": rows["last contact"]["mean_here"], "contact_duty_ref": rows["last contact"]["mean_ref"],
           "air_time_mean": rows["feet air time"]["mean_here"], "air_time_mean_ref": rows["feet air time"]["mean_ref"],
           "std_ratio": {k: rows[k]["std_ratio"] for k in ("gravity", "gyro", "joint vel", "actuator force", "feet linvel", "accelerometer")},
           "slab_base_gain_m": one["base_z"] - flat["base_z"], "slab_vx": one["vx"], "flat_vx": flat["vx"], "slab_survival": one["survival"],
           "box_contacts": int(hist["n"]), "box_contacts_deeper_than_radius": int(hist["over_radius"]),
           "frac_deeper_than_radius": hist["over_radius"] / max(hist["n"], 1), "max_penetration_m": hist["max"],
           "plane_contacts": int(hist["plane_n"]), "plane_contacts_deeper_than_radius": int(hist["plane_over_radius"]),
           "hist_edges_mm": [round(1e3 * e, 3) for e in edges], "hist_counts": counts.tolist(),
           "hist_cum_frac": (np.cumsum(counts) / max(counts.sum(), 1)).round(5).tolist()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
