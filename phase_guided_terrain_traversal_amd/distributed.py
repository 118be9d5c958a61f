"""Multi-GPU layer: envs shard trivially (one process per GPU, contiguous global env-id ranges, terrain table
and model replicated); the ONLY collective on the path is the all-reduce of the episodic return / metric sums
(reference: Brax's Episode-wrapper metrics are `pmean`-ed across devices inside ppo.train — SURVEY 2.1, 5).

`torch.distributed` backend "nccl" is RCCL on ROCm (xGMI inside a node); the message is 25 floats, i.e. pure
latency, so it is issued as ONE fused buffer every `interval` steps, never per metric.  With backend "gloo" the
same code runs on CPU tensors (used by the world_size-2 tests).
"""
from __future__ import annotations

import os
from collections.abc import Mapping
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist

from . import abi


def shard_range(num_envs_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous env-id range [lo, hi) of `rank`; the remainder goes to the lowest ranks."""
    base, rem = divmod(int(num_envs_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend: Optional[str] = None, force: bool = False) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from the torchrun environment; initialises the process group when world > 1 (or when
    `force` is set: a one-rank group still routes the collective through RCCL, which is how the single-GPU test box
    exercises the library)."""
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            # one process per GPU: bind the device BEFORE the communicator exists (RCCL otherwise guesses it at the first
            # collective), and hand it to init_process_group so that barrier() / all_reduce run on it
            if not torch.cuda.is_available() or torch.cuda.device_count() <= local:
                raise RuntimeError(f"rank {rank}: LOCAL_RANK {local} has no GPU (device_count = "
                                   f"{torch.cuda.device_count() if torch.cuda.is_available() else 0})")
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        try:
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
        except TypeError:          # torch without the device_id argument
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


class _Reduced(Mapping):
    """result of a reduction: the all-reduced sums [22 metric sums, sum_reward, n_done, n_env_steps] behind the keys `metrics_mean`,
    `reward_mean`, `done_count`, `env_steps` (and `sums`).  The two means cost a kernel each and only a logging step reads them, so they are
    formed when first asked for - but they ARE keys: `in`, `.get()`, iteration and `**` see the same five whichever reduction returned the object."""

    KEYS = ("metrics_mean", "reward_mean", "done_count", "env_steps", "sums")

    def __init__(self, sums: torch.Tensor):
        self._sums, self._cache = sums, {}

    def __getitem__(self, key):
        if key in self._cache:
            return self._cache[key]
        buf = self._sums
        if key == "sums":
            v = buf
        elif key == "done_count":
            v = buf[abi.NMETRIC + 1]
        elif key == "env_steps":
            v = buf[abi.NMETRIC + 2]
        elif key == "metrics_mean":
            v = buf[:abi.NMETRIC] / buf[abi.NMETRIC + 2].clamp(min=1.0)
        elif key == "reward_mean":
            v = buf[abi.NMETRIC] / buf[abi.NMETRIC + 2].clamp(min=1.0)
        else:
            raise KeyError(key)
        self._cache[key] = v
        return v

    def __iter__(self):
        return iter(self.KEYS)

    def __len__(self):
        return len(self.KEYS)


class MetricReducer:
    """Fused all-reduce of [22 metric sums, sum_reward, n_done, n_envs] (= 25 floats)."""

    SIZE = abi.NMETRIC + 3

    def __init__(self, device: torch.device):
        self.device = device
        self.acc = torch.zeros(self.SIZE, dtype=torch.float32, device=device)
        self.count = 0.0          # env-steps accumulated since the last reduce (kept on the host: no kernel per step)
        self._dirty = False       # the accumulator holds contributions since the last reduce
        self._ones = None

    def accumulate(self, metrics: torch.Tensor, reward: torch.Tensor, done: torch.Tensor) -> None:
        """metrics [22, N], reward [N], done [N] of one step (local shard)."""
        self._dirty = True
        self.acc[:abi.NMETRIC] += metrics.sum(dim=1)
        self.acc[abi.NMETRIC] += reward.sum()
        self.acc[abi.NMETRIC + 1] += done.sum()
        self.count += float(reward.shape[0])

    def accumulate_block(self, block: torch.Tensor) -> None:
        """`Joystick.step_block` ([22 metrics; reward; done][N]) of one step: ONE launch, acc += block @ 1 (GEMV with
        beta = 1) instead of a reduction kernel plus an add (~5 us of a 0.24 ms step each)."""
        n = block.shape[1]
        if self._ones is None or self._ones.shape[0] != n:
            self._ones = torch.ones(n, dtype=torch.float32, device=block.device)
        self._dirty = True
        self.acc[:abi.NMETRIC + 2].addmv_(block, self._ones)
        self.count += float(n)

    def reduce_block(self, sums: torch.Tensor, env_steps: float) -> Dict[str, torch.Tensor]:
        """`Joystick.buffers["interval_sums"]` ([22 metrics; reward; done][N], kept by the step kernels) over `env_steps` env-steps
        of this rank: ONE GEMV over the envs for the whole interval, the block cleared for the next one, then the fused all-reduce."""
        n = sums.shape[1]
        if self._ones is None or self._ones.shape[0] != n:
            self._ones = torch.ones(n, dtype=torch.float32, device=sums.device)
        self._dirty = True
        self.acc[:abi.NMETRIC + 2].addmv_(sums, self._ones)
        sums.zero_()
        self.count += float(env_steps)
        return self.reduce()

    def reduce_env(self, env, env_steps: float) -> Dict[str, torch.Tensor]:
        """reduce_block for a `Joystick` that keeps interval sums: the sum over the envs, the clearing of the block and the accumulation
        are ONE launch of libpgtt (pgtt_interval_reduce) instead of a GEMV, a fill and an add; then the fused all-reduce."""
        if self._dirty or self.count != 0.0:                   # per-step contributions are waiting in the accumulator: the general path
            env.interval_reduce(self.acc, 0.0, accumulate=True)
            self.count += float(env_steps)
            return self.reduce()
        buf = torch.empty(self.SIZE, dtype=torch.float32, device=self.acc.device)
        env.interval_reduce(buf, float(env_steps))
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        return _Reduced(buf)

    def reduce(self) -> Dict[str, torch.Tensor]:
        """Sum over ranks (one RCCL all-reduce), reset the local accumulator, return global means."""
        buf = self.acc.clone()
        buf[abi.NMETRIC + 2] += self.count
        self.count = 0.0
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        self.acc.zero_()
        self._dirty = False
        return _Reduced(buf)
