"""Two ranks over RCCL on ONE node (`-m gpu`, self-skipping on a box with fewer than two GPUs).

The driver's 8-GPU scaling run must not be the first execution of the N > 1 code: on any box with two or more devices these tests run the
spawned and the torchrun launch forms of bench.py with backend "nccl" (= RCCL), check that the env shards of two ranks hold, bit for bit, what
a single process computes for the same global env ids (pinned lane layout: results are bit-identical across batch sizes within one layout),
and take train.py through two data-parallel PPO iterations.  The path's only collective is the fused 25-float metric all-reduce
(training/train.py:242-263 spreads its envs over the local devices the same way, through Brax's pmap)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need_two_gpus():
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < 2:
        pytest.skip(f"needs two GPUs on the node, this box has {have}")


def _clean_env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(kw)
    return env


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_bench_two_ranks_spawned_rccl():
    """`python bench.py --gpus 2` spawns one process per GPU; both ranks contribute every interval of the all-reduce"""
    _need_two_gpus()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"],
                       env=_clean_env(), capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1                                   # rank 0 only
    d = lines[0]
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "error" not in d
    assert d["env_steps_allreduced"] == d["env_steps_expected"] == 2 * 4096 * 20
    assert "nccl" in d["config"]["collective"] and d["config"]["parallelism"] == "env-shard x2"
    assert abs(d["value"] - d["env_steps_allreduced"] / (d["ms_per_step"] * 1e-3 * 20)) < 1e-6 * d["value"]


def test_bench_two_ranks_torchrun_rccl():
    """the driver's launch form: python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2"""
    _need_two_gpus()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29655",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _json_lines(p.stdout)
    assert len(d) == 1 and d[0]["n_gpus"] == 2 and d[0]["env_steps_allreduced"] == 2 * 4096 * 20


SHARD_WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from phase_guided_terrain_traversal_amd import abi, configs, mjcf
from phase_guided_terrain_traversal_amd.distributed import MetricReducer, init_from_env, shard_range
from phase_guided_terrain_traversal_amd.env import Joystick
from phase_guided_terrain_traversal_amd.randomize import domain_randomize
out_dir, total, steps, backend = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
rank, local, world = init_from_env(backend, force=True)
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
lo, hi = shard_range(total, rank, world)
n = hi - lo
terrain = np.load(os.path.join(os.path.dirname(mjcf.__file__), "assets", "terrains", "level13.npy"))
dr = domain_randomize(mjcf.load_model("stairs"), n, seed=3, terrain=terrain, env_id_offset=lo, total_envs=total)
env = Joystick("stairs", configs.with_overrides(configs.training_config(), episode_length=13), num_envs=n, terrain=terrain, device=f"cuda:{local}", autoreset=True,
               env_id_offset=lo, layout="hex", interval_sums=True, variant=torch.from_numpy(dr["variant"]), params=torch.from_numpy(dr["params"]),
               box_friction=torch.from_numpy(dr["box_friction"]))
env.reset(seed=4)
acts = np.tanh(np.random.Generator(np.random.Philox(key=[9, 0])).normal(size=(steps, total, 12)) * 0.6).astype(np.float32)
red = MetricReducer(dev)
for k in range(steps):
    env.step(torch.from_numpy(acts[k, lo:hi]).to(dev))
res = red.reduce_env(env, float(steps) * n)
torch.cuda.synchronize()
np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=lo, hi=hi, **{k: env.buffers[k].cpu().numpy() for k in ("state", "istate", "obs_state", "obs_priv", "reward", "done", "ep_metrics")})
print(json.dumps({"rank": rank, "world": world, "env_steps": float(res["env_steps"]), "reward_mean": float(res["reward_mean"]), "done_count": float(res["done_count"])}))
dist.destroy_process_group()
'''


def _run_shards(tmp_path, world, total, steps, port, backend="nccl", one_gpu=False):
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER)
    out = tmp_path / f"w{world}"; out.mkdir()
    env = _clean_env(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(out), str(total), str(steps), backend],
                              env=dict(env, RANK=str(r), LOCAL_RANK="0" if one_gpu else str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    lines = []
    for p in procs:
        o, e = p.communicate(timeout=900)
        assert p.returncode == 0, e[-3000:]
        lines.append(_json_lines(o)[-1])
    return out, lines


def test_two_rank_shards_hold_the_single_process_bits(tmp_path):
    """level13 + full DR + AutoReset (short episodes), 2048 envs: ranks 0 / 1 own global env ids [0, 1024) / [1024, 2048) on their own GPUs; every
    buffer equals, bit for bit, the slice of ONE process stepping all 2048 (DR, variants, reset and command draws are keyed by the global id),
    and the all-reduce of the interval sums returns the single process's totals in both ranks.  With two or more GPUs the ranks sit on their own
    devices and talk over RCCL; on a one-GPU box the two rank PROCESSES share cuda:0 and the 25 floats travel over gloo (RCCL refuses two ranks on
    one device) - the shard arithmetic, the global-id keyed draws and the collective's call sequence are the same code either way."""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    assert have >= 1
    two_gpus = have >= 2
    total, steps = 2048, 30
    one, l1 = _run_shards(tmp_path, 1, total, steps, 29661)
    two, l2 = _run_shards(tmp_path, 2, total, steps, 29662, backend="nccl" if two_gpus else "gloo", one_gpu=not two_gpus)
    full = np.load(one / "rank0.npz")
    for r in range(2):
        part = np.load(two / f"rank{r}.npz")
        lo, hi = int(part["lo"]), int(part["hi"])
        assert (lo, hi) == (1024 * r, 1024 * (r + 1))
        for k in ("state", "istate", "obs_state", "obs_priv", "reward", "done", "ep_metrics"):
            a, b = full[k], part[k]
            sl = a[lo:hi] if a.shape[0] == total else a[..., lo:hi]
            assert sl.shape == b.shape and np.array_equal(sl.view(np.uint32), b.view(np.uint32)), (r, k)
    assert l1[0]["env_steps"] == total * steps and all(l["env_steps"] == total * steps and l["world"] == 2 for l in l2)
    assert l2[0]["done_count"] == l2[1]["done_count"] == l1[0]["done_count"] > 0
    assert abs(l2[0]["reward_mean"] - l1[0]["reward_mean"]) < 1e-6 and l2[0]["reward_mean"] == l2[1]["reward_mean"]


def test_train_two_ranks_torchrun(tmp_path):
    """train.py data parallel over two GPUs for two PPO iterations (gradient all-reduce, summed normaliser moments, rank-0 logging)"""
    _need_two_gpus()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29656",
           os.path.join(ROOT, "train.py"), "--task_name", "stairs", "--terrain_file", "level4", "--num_envs", "1024", "--batch_size", "64", "--num_minibatches", "16",
           "--num_timesteps", str(2 * 20 * 1024), "--num_evals", "2", "--index", "901"]
    p = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=1800, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-3000:]
    rows = [l for l in p.stdout.splitlines() if l.startswith("steps ")]
    assert len(rows) >= 2 and "time to train" in p.stdout                        # rank 0 logged every evaluation, once
    assert int(rows[-1].split()[1]) == 2 * 20 * 1024
