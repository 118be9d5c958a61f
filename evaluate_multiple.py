"""Level sweep of trained policies - what the reference's training/evaluate_multiple.py:9-45 does with `evaluate.run_training`:
one policy per method, a list of terrain files, per file the number of evaluation envs (of 1000) that get through the episode
without falling; the two result vectors are saved as plots/pgtt_results.npy and plots/baseline_results.npy.

    python evaluate_multiple.py                                   (the shipped reference-trained policies on the shipped levels)
    python evaluate_multiple.py --pgtt checks_stairs/checkpoint_122 --baseline checks_stairs/checkpoint_125 --levels terrains/level09.npy ...

The reference sweeps `terrains/level09.npy .. level11.npy` (files its terrain/generator.py:430-437 writes; they are not in its
repository) with its checkpoints 122 (pgtt) and 125 (baseline).  The defaults here are what ships: levels 4, 7, 10, 13 and the
exported policy177 / policy175.  A policy argument is a checkpoint folder of train.py, an .npz path, or a shipped policy's name.
"""
import argparse
import os

import numpy as np

import evaluate

HERE = os.path.dirname(os.path.abspath(__file__))


def survivors_per_level(method, policy, levels):
    """[number of the 1000 evaluation envs that did not fall, for every terrain file]"""
    source = ["--checkpoint_folder", policy] if os.path.isdir(policy) else ["--policy", policy]
    counts = []
    for level in levels:
        args = evaluate.make_parser().parse_args(["--method", method, "--task_name", "stairs", "--terrain_file", level] + source)
        res = evaluate.run_evaluation(args, verbose=False)
        print(f"[{method:8s}] {level:>24s}: {res['survivors']:4d} of {res['num_eval_envs']}   reward / episode {res['episode_reward']:7.2f}   "
              f"tracking lin {res['tracking_lin_vel']:.2f} ang {res['tracking_ang_vel']:.2f}", flush=True)
        counts.append(res["survivors"])
    return counts


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="sweep a pgtt and a baseline policy over terrain levels")
    ap.add_argument("--pgtt", default="policy177")
    ap.add_argument("--baseline", default="policy175")
    ap.add_argument("--levels", nargs="*", default=["level4", "level7", "level10", "level13"])
    opt = ap.parse_args()
    table = {"pgtt": survivors_per_level("pgtt", opt.pgtt, opt.levels), "baseline": survivors_per_level("baseline", opt.baseline, opt.levels)}
    out = os.path.join(HERE, "plots")
    os.makedirs(out, exist_ok=True)
    for name, counts in table.items():
        np.save(os.path.join(out, f"{name}_results.npy"), np.array(counts))
    print("\nlevels  ", opt.levels)
    for name, counts in table.items():
        print(f"{name:8s}", counts)
    print(f"saved: {out}/pgtt_results.npy, {out}/baseline_results.npy")
