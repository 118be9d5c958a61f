"""Benchmark of the hot path: env-steps/s of the vectorised Go2 joystick_pgtt step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs 4096] [--workload level4|flat|level13_dr|wfc_dr|curriculum]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one control step (4 physics substeps + 13x9 scan + obs + 21 rewards + bookkeeping, with the
Episode/AutoReset wrapper fused) over the per-GPU batch of synthetic actions.  Workload = BASELINE.json
configs[2]: 4096 Go2 envs per GPU on terrains/level4.npy with the height scan, no DR (the configuration the
metric "env-steps/sec at 4096 envs (Go2, level4 hfield)" is quoted on).  Weak scaling: every GPU owns 4096
envs; the only collective is the 25-float metric all-reduce every 20 steps.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (physics_kernel); `cpu_baseline` is the
build's own CPU restatement (oracle/, kind "port") timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic per-env-step figures (restated in DESIGN.md)
ALGO_BYTES_PER_ENV_STEP = 3456          # no-DR workload; 4216 with per-env DR params
ALGO_BYTES_PER_ENV_STEP_DR = 4216
ALGO_FLOP_PER_ENV_STEP = 1.6e6          # fp32, dense MJX formulation (the reference's arithmetic)
PEAK_FP32_TFLOPS = 157.3                # MI355X fp32: matrix peak == vector peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
CURRICULUM = [1, 2, 3, 4, 7, 10, 13]    # the level files the reference ships (terrains/level*.npy), easiest first


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--workload", default="level4", choices=["level4", "flat", "level13_dr", "wfc_dr", "curriculum"])
    ap.add_argument("--stage", type=int, default=None, help="curriculum workload: index into the level list (default: the rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-steps", type=int, default=100)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from phase_guided_terrain_traversal_amd import abi, configs, mjcf
    from phase_guided_terrain_traversal_amd.distributed import MetricReducer, init_from_env
    from phase_guided_terrain_traversal_amd.env import Joystick
    from phase_guided_terrain_traversal_amd.randomize import domain_randomize

    rank, local, world = init_from_env("nccl")
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    n = args.envs
    off = rank * n
    assets = os.path.join(ROOT, "phase_guided_terrain_traversal_amd", "assets", "terrains")
    cfg = configs.training_config()
    kw, terrain, task, dr = {}, None, "stairs", False
    if args.workload == "flat":
        task = "flat_terrain"
    elif args.workload in ("level4", "curriculum"):
        # curriculum (BASELINE configs[4]): GPU r trains on stage r of the reference's level files (terrains/level*.npy)
        level = "level4" if args.workload == "level4" else "level%d" % CURRICULUM[(rank if args.stage is None else args.stage) % len(CURRICULUM)]
        terrain = np.load(os.path.join(assets, level + ".npy"))
        variant = np.random.Generator(np.random.Philox(key=[2, 0])).integers(0, terrain.shape[0], n * max(world, 1))[off:off + n]
        kw["variant"] = torch.from_numpy(variant.astype(np.int32))
    else:
        if args.workload == "wfc_dr":           # BASELINE configs[3]: WFC-generated terrain (host, once) + full randomize.py DR
            from phase_guided_terrain_traversal_amd.terrain_gen import create_random_matrix
            terrain = create_random_matrix(100, 100, 5, 0.05, 0.13, seed=3)
        else:
            terrain = np.load(os.path.join(assets, "level13.npy"))
        dr = True
        out = domain_randomize(mjcf.load_model("stairs"), n, seed=3, terrain=terrain, env_id_offset=off)
        kw = {"variant": torch.from_numpy(out["variant"]), "params": torch.from_numpy(out["params"]),
              "box_friction": torch.from_numpy(out["box_friction"])}
    env = Joystick(task, cfg, num_envs=n, terrain=terrain, device=f"cuda:{local}", autoreset=True, env_id_offset=off, **kw)
    env.reset(seed=0)
    g = torch.Generator(device=dev); g.manual_seed(1 + rank)
    pool = [torch.tanh(torch.randn(n, 12, generator=g, device=dev) * 0.6) for _ in range(32)]   # tanh(N(0,0.6)), SURVEY 8d
    reducer = MetricReducer(dev)

    def run(k0, k1):
        for k in range(k0, k1):
            obs, reward, done, info = env.step(pool[k % len(pool)])
            reducer.accumulate_block(env.step_block)
            if (k + 1) % 20 == 0:
                reducer.reduce()

    run(0, args.warmup)
    torch.cuda.synchronize()
    # per-kernel durations over the timed steps themselves: HIP events recorded by libpgtt around its own launches on
    # the launch stream (every 8th step), kept in a ring and read back later (no stall in the timed loop)
    env.enable_timing(8)          # every 8th step: an event record costs a few us of GPU idle
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.warmup, args.warmup + args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    phys_ms, obs_ms, ntimed = env.kernel_ms_mean()
    env.enable_timing(False)
    done_frac = float(env.buffers["done"].mean().item())

    if rank == 0:
        total_envs = n * world
        value = total_envs * args.steps / dt
        algo_bytes = (ALGO_BYTES_PER_ENV_STEP_DR if dr else ALGO_BYTES_PER_ENV_STEP) * n
        algo_flop = ALGO_FLOP_PER_ENV_STEP * n
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"{args.workload}_{n}", {}).get("physics_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec at 4096 envs (Go2, level4 hfield), 1/2/4/8 MI355X",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": {"level4": "4096 Go2 envs/GPU, terrains/level4.npy (100 variants x 100 boxes) + 13x9 height scan, no DR (BASELINE configs[2])",
                                    "flat": "4096 Go2 envs/GPU, plane only, no DR (BASELINE configs[1])",
                                    "level13_dr": "Go2 envs/GPU, level13 + full randomize.py DR (BASELINE configs[3] shape)",
                                    "curriculum": "Go2 envs/GPU, rank r on stage r of terrains/level{1,2,3,4,7,10,13}.npy + height scan, no DR (BASELINE configs[4])",
                                    "wfc_dr": "Go2 envs/GPU, WFC-generated stairs (terrain_gen.py, 100 variants) + full randomize.py DR (BASELINE configs[3])"}[args.workload],
                       "envs_per_gpu": n, "substeps": 4, "autoreset": True, "actions": "tanh(N(0,0.6)) iid", "parallelism": f"env-shard x{world}"},
            "kernels_ms": {"physics_kernel": phys_ms, "observe_kernel": obs_ms, "launches": ntimed},
            "done_fraction_last_step": done_frac,
            "roofline": {"bound": "mfma", "achieved": algo_flop / (phys_ms * 1e-3) / 1e12, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": algo_flop / (phys_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, "traffic": traffic,
                         "kernel": "physics_kernel", "note": "fp32: matrix peak == vector peak (157.3 TF); kernel is FP32-VALU/latency bound, no MFMA issued"},
            "roofline_hbm": {"bound": "hbm", "achieved": algo_bytes / (phys_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": algo_bytes / (phys_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": traffic},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, cfg, terrain, task, n)
        print(json.dumps(out))
    env.close()
    if world > 1:
        dist.destroy_process_group()


def host_cores():
    """Usable host cores: the smaller of the affinity mask and the cgroup CPU quota (the GPU box shows 256 CPUs
    but grants a 16-CPU quota)."""
    c = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            c = min(c, max(1, int(round(int(q) / int(p)))))
    except Exception:
        pass
    return c


def cpu_baseline(args, cfg, terrain, task, n):
    """The build's CPU restatement (oracle/, fp32, OpenMP over envs) on a bounded sample of the same workload."""
    from oracle import oracle
    from phase_guided_terrain_traversal_amd import abi, configs, mjcf
    cores = host_cores()
    cfg2 = dict(cfg); cfg2["autoreset"] = 1
    cs, ms = abi.config_struct(cfg2), abi.model_struct(mjcf.load_model(task))
    hb = oracle.HostBuffers(n, with_variant=terrain is not None, debug=False)
    if terrain is not None:
        hb["variant"][:] = np.random.Generator(np.random.Philox(key=[2, 0])).integers(0, terrain.shape[0], n).astype(np.int32)
    oracle.reset(cs, ms, terrain, hb, seed=0, nthreads=cores)
    rng = np.random.default_rng(1)
    acts = [np.tanh(rng.normal(size=(n, 12)) * 0.6).astype(np.float32) for _ in range(8)]
    for k in range(3):
        oracle.step(cs, ms, terrain, hb, acts[k % 8], seed=0, nthreads=cores)
    t0 = time.perf_counter()
    for k in range(args.cpu_sample_steps):
        oracle.step(cs, ms, terrain, hb, acts[k % 8], seed=0, nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": n * args.cpu_sample_steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} envs x {args.cpu_sample_steps} control steps of the same workload ({args.workload}), fp32 oracle, OpenMP over envs, {dt:.1f} s"}


if __name__ == "__main__":
    main()
