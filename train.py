"""CLI mirror of the reference's training/train.py:285-301 (same flags, same PPO settings, same early-stop rule),
driving the HIP environment through the reset()/step()/obs API.

    python train.py --method pgtt --task_name stairs --terrain_file level4 --num_timesteps 20000000 --index 1

    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py --num_envs 32768 ...      (data parallel)

`--terrain_file` accepts a shipped level name (level1,2,3,4,7,10,13) or a path to a (T,100,10) .npy.
With a torchrun environment (WORLD_SIZE > 1) the job is data parallel, one process per GPU over RCCL, the way the reference's
Brax PPO spreads over its local devices: `--num_envs` and `--batch_size` stay the job's totals, rank r owns the contiguous
global env ids of `distributed.shard_range` (DR and reset draws are keyed by global id, so a shard is the same whatever the
world size), gradients and observation statistics are all-reduced inside `ppo.train`, rank 0 logs and writes checkpoints.
Checkpoints: checks_stairs/checkpoint_<index>/<env_steps>.pt (torch.save of policy/value weights + normalisers),
resumed with --checkpoint_folder like the reference (training/train.py:249-257).
"""
import argparse
import json
import os
import time

import numpy as np
import torch

from phase_guided_terrain_traversal_amd import configs, mjcf, ppo
from phase_guided_terrain_traversal_amd.distributed import init_from_env, shard_range
from phase_guided_terrain_traversal_amd.env import Joystick
from phase_guided_terrain_traversal_amd.randomize import domain_randomize

ROOT = os.path.dirname(os.path.abspath(__file__))
DEVICE = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))      # one process per GPU; this script trains on the GPU of its rank


def load_terrain(spec):
    p = spec if os.path.exists(spec) else os.path.join(ROOT, "phase_guided_terrain_traversal_amd", "assets", "terrains",
                                                       os.path.basename(spec).replace(".npy", "") + ".npy")
    return np.load(p)


def run_training(args):
    if args.method not in ("pgtt", "baseline"):
        raise SystemExit("--method must be pgtt (go2/joystick_pgtt.py) or baseline (go2/joystick.py)")
    cfg = configs.training_config(args.method)                    # train.py:120-129: config by method + overrides
    model = mjcf.load_model(args.task_name)
    terrain = load_terrain(args.terrain_file) if args.task_name == "stairs" else None
    rank, _, world = init_from_env()
    lo, hi = shard_range(args.num_envs, rank, world)
    if (hi - lo) * world != args.num_envs:
        raise SystemExit(f"--num_envs {args.num_envs} must be a multiple of the world size {world}")
    dr = domain_randomize(model, hi - lo, seed=args.index, terrain=terrain, env_id_offset=lo, total_envs=args.num_envs)      # once per env index
    kw = {"params": torch.from_numpy(dr["params"])}
    if terrain is not None:
        kw.update(variant=torch.from_numpy(dr["variant"]), box_friction=torch.from_numpy(dr["box_friction"]))
    env = Joystick(args.task_name, cfg, num_envs=hi - lo, terrain=terrain, device=DEVICE, autoreset=True, env_id_offset=lo, **kw)
    ckpt = os.path.join(ROOT, "checks_stairs", f"checkpoint_{args.index}")
    if rank == 0:
        os.makedirs(ckpt, exist_ok=True)
        json.dump({k: v for k, v in cfg.items()}, open(os.path.join(ckpt, "config.json"), "w"), indent=4, default=str)
    restore = None
    if args.checkpoint_folder:
        steps = [int(f[:-3]) for f in os.listdir(args.checkpoint_folder) if f.endswith(".pt") and f[:-3].isdigit()]
        restore = torch.load(os.path.join(args.checkpoint_folder, f"{max(steps)}.pt"), map_location=DEVICE)
        if rank == 0:
            print("restoring", args.checkpoint_folder, max(steps))
    y, lin, ang, times = [], [], [], [time.time()]

    def progress(num_steps, m):                                   # training/train.py:198-229
        times.append(time.time())
        y.append(m["eval/episode_reward"])
        L = cfg["episode_length"]
        vel = m["eval/episode_reward/tracking_lin_vel"] / (cfg["reward_config"]["scales"]["tracking_lin_vel"] * L)
        av = m["eval/episode_reward/tracking_ang_vel"] / (cfg["reward_config"]["scales"]["tracking_ang_vel"] * L)
        lin.append(vel); ang.append(av)
        if rank == 0:
            print(f"steps {num_steps:>12d}  reward/episode {y[-1]:9.3f}  len {m['eval/avg_episode_length']:7.1f}  lin {vel:.3f}  ang {av:.3f}  "
                  f"rollout {m['env_steps_per_s_rollout'] / 1e6:.2f} M steps/s  total {m['env_steps_per_s_total'] / 1e6:.2f} M steps/s", flush=True)
        if len(y) >= 2 and y[-1] != 0:
            rel = abs((y[-1] - y[-2]) / y[-1])
            if (vel > cfg["vel_percentage"] and av > cfg["vel_percentage"] and rel <= 0.005) or rel <= 0.001:
                return True
        return False

    def save_params(num_steps, params):
        """training/train.py:189-195: a checkpoint per evaluation (resume) plus a `policy{index}` file a deployment script can read
        (here the npz layout of policy.PolicyMLP / tools/rollout_policy.py instead of a pickled Brax pytree)"""
        torch.save(params, os.path.join(ckpt, f"{num_steps}.pt"))
        ppo.export_policy_npz(params, os.path.join(ckpt, f"policy{args.index}.npz"))

    pcfg = ppo.PPOConfig(num_timesteps=args.num_timesteps, num_evals=args.num_evals, num_minibatches=args.num_minibatches,
                         batch_size=args.batch_size, discounting=args.discount, learning_rate=args.learning_rate, seed=args.index)
    model_, norms, hist = ppo.train(env, pcfg, progress_fn=progress,
                                    policy_params_fn=save_params, restore=restore)
    if rank != 0:
        return hist
    print(f"time to train: {times[-1] - times[0]:.1f} s")
    os.makedirs(os.path.join(ROOT, "plots", args.method), exist_ok=True)
    np.save(os.path.join(ROOT, "plots", args.method, f"mean{args.index}"), np.array(y))
    np.save(os.path.join(ROOT, "plots", args.method, f"lin_vel{args.index}"), np.array(lin))
    np.save(os.path.join(ROOT, "plots", args.method, f"anf_vel{args.index}"), np.array(ang))
    return hist


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Train PPO on the MI355X-native PGTT env")
    ap.add_argument("--method", type=str, default="pgtt")
    ap.add_argument("--task_name", type=str, default="stairs")
    ap.add_argument("--terrain_file", type=str, default="terrains/level1.npy")
    ap.add_argument("--checkpoint_folder", type=str, default=None)
    ap.add_argument("--num_envs", type=int, default=4096)
    ap.add_argument("--batch_size", type=int, default=256)
    ap.add_argument("--discount", type=float, default=0.97)
    ap.add_argument("--learning_rate", type=float, default=3e-4)
    ap.add_argument("--num_minibatches", type=int, default=32)
    ap.add_argument("--num_timesteps", type=int, default=1)
    ap.add_argument("--num_evals", type=int, default=31)
    ap.add_argument("--index", type=int, default=32)
    run_training(ap.parse_args())
