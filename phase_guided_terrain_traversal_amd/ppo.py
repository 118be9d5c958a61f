"""PPO learner around the env (N1 of SURVEY 8f): a torch restatement of what the reference gets from
`brax.training.agents.ppo.train` with the settings of training/train.py:135-161:

    unroll_length 20, num_minibatches 32, batch_size 256, num_updates_per_batch 4, lr 3e-4, discounting 0.97,
    entropy_cost 1e-2, max_grad_norm 1.0, reward_scaling 1.0, normalize_observations, policy / value MLPs
    (512, 256, 128) with swish, policy obs key "state" (171), value obs key "privileged_state" (215).

Brax defaults it relies on [UPSTREAM-RECALL]: tanh-normal action distribution with scale = softplus(raw) + 1e-3,
clipping_epsilon 0.3, gae_lambda 0.95, normalize_advantage (population std), value loss 0.5 * 0.5 * mse, compute_gae with
zeroed temporal-difference error and carry at truncated steps.  This is the CALLER of the hot path (out of the §8 a-e scope); it exists so that env-steps/s can be
measured inside a real training loop and `train.py` keeps the reference's CLI.  Dense layers are plain library GEMMs
(torch -> hipBLASLt); two launch-bound pieces of the update are hand-written HIP in csrc/pgtt_ppo.hip (the fused policy
loss with its gradient, and the split-K fp32-MFMA weight / bias gradient of the Linear layers over the long minibatch).
"""
from __future__ import annotations

import math
import os
import time
from dataclasses import dataclass
from typing import Callable, Dict, Optional

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import abi


def _world():
    """(world size, rank) of the data-parallel job; (1, 0) without a process group."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def _dp() -> bool:
    """Are the data-parallel code paths (all-reduces of gradients / statistics / logs) active?  Always with more than one rank;
    PGTT_PPO_FORCE_DP=1 switches them on for a ONE-rank process group too - the collectives are then identities through RCCL,
    which is how a single-GPU box runs exactly the code an 8-GPU job runs (tests/test_gpu_train.py)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("PGTT_PPO_FORCE_DP", "0") == "1")


class RunningNorm:
    """Welford running mean/std per observation key (brax running_statistics): count starts at 0 with std = 1, std floor 1e-6."""

    def __init__(self, dim: int, device):
        self.count = torch.zeros((), device=device, dtype=torch.float64)
        self.mean = torch.zeros(dim, device=device)
        self.m2 = torch.zeros(dim, device=device)

    @torch.no_grad()
    def update(self, x: torch.Tensor) -> None:
        """Chan's parallel update with the batch moments; column sums as (1 x M) @ (M x d) products — torch's strided
        dim-0 reductions of a [163840, 171] tensor take 6 ms, the GEMV form 0.3 ms — and no host synchronisation."""
        x = x.reshape(-1, x.shape[-1])
        n = x.shape[0]
        ones = torch.ones(1, n, device=x.device, dtype=x.dtype)
        if _dp():
            # data parallel (brax running_statistics.update with pmap_axis_name: the step's moments are psum-ed): the batch
            # is the union of the ranks' shards (equal heights: ppo.train gives every rank the same number of envs) - its
            # mean from one all-reduce of the column sums, its squared deviations about THAT mean from a second one (no raw
            # sums of squares in fp32); every rank ends with the same statistics
            col = (ones @ x).squeeze(0)
            dist.all_reduce(col)
            n = n * _world()[0]
            mean_b = col / n
            xc = x - mean_b
            batch_m2 = (ones @ (xc * xc)).squeeze(0)
            dist.all_reduce(batch_m2)
        else:
            mean_b = (ones @ x).squeeze(0) / n
            xc = x - mean_b
            batch_m2 = (ones @ (xc * xc)).squeeze(0)
        new_count = self.count + n
        delta = mean_b - self.mean
        w = (self.count * n / new_count).to(x.dtype)
        self.m2 += batch_m2 + delta ** 2 * w
        self.mean += delta * (n / new_count).to(x.dtype)
        self.count.copy_(new_count)           # in place: a captured graph keeps reading this tensor

    @property
    def std(self) -> torch.Tensor:
        var = self.m2 / self.count.clamp(min=1.0).to(self.m2.dtype)        # device-only: usable inside a captured graph
        std = torch.sqrt(torch.clamp(var, min=1e-12)).clamp(min=1e-6)
        # brax running_statistics.init_state: std = 1 before the first update (graph warm-up and the first rollout would
        # otherwise feed obs * 1e6 to the policy)
        return torch.where(self.count > 0, std, torch.ones_like(std))

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return (x - self.mean) / self.std


class _LinearLongBatch(torch.autograd.Function):
    """y = x W^T + b with the weight / bias gradient through libpgtt's pgtt_ppo_linear_backward (split-K fp32 MFMA, the
    column sums of dY in the same launch): for K = 5120 rows and 1..512 columns the library GEMM walks K in a handful of
    workgroups (34 us whatever the size) and the bias gradient is a second reduction."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        from . import native
        import ctypes
        x, w = ctx.saved_tensors
        dy = dy.contiguous(); x = x.contiguous()
        K, M = x.shape
        N = w.shape[0]
        dx = dy @ w if ctx.needs_input_grad[0] else None
        tiles = ((M + 63) // 64) * ((N + 63) // 64)
        S = max(1, min(64, int(os.environ.get("PGTT_PPO_SPLITK", "768")) // tiles, K // 16))
        partial = torch.empty(S * (N * M + N), dtype=torch.float32, device=x.device)
        dw = torch.empty_like(w); db = torch.empty(N, dtype=torch.float32, device=x.device)
        native.check(native.lib().pgtt_ppo_linear_backward(
            ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(dy.data_ptr()), ctypes.c_int(K), ctypes.c_int(M), ctypes.c_int(N), ctypes.c_int(S),
            ctypes.c_void_p(partial.data_ptr()), ctypes.c_void_p(dw.data_ptr()), ctypes.c_void_p(db.data_ptr()),
            ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        return dx, dw, db


class LongBatchLinear(nn.Linear):
    """nn.Linear (same parameters, same state_dict) whose backward over a long batch runs on the hand-written kernel."""

    def forward(self, x):
        if (x.is_cuda and x.dim() == 2 and x.shape[0] >= 1024 and torch.is_grad_enabled() and self.weight.requires_grad
                and x.dtype == torch.float32 and os.environ.get("PGTT_PPO_FUSED", "1") != "0" and os.environ.get("PGTT_PPO_LINEAR", "1") != "0"):
            return _LinearLongBatch.apply(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)


def mlp(sizes, out):
    layers, d = [], sizes[0]
    for h in sizes[1:]:
        layers += [LongBatchLinear(d, h), nn.SiLU()]
        d = h
    layers.append(LongBatchLinear(d, out))
    return nn.Sequential(*layers)


class ActorCritic(nn.Module):
    def __init__(self, obs_dim=abi.OBS, priv_dim=abi.PRIV, act_dim=abi.NU, hidden=(512, 256, 128)):
        super().__init__()
        self.policy = mlp((obs_dim,) + tuple(hidden), 2 * act_dim)
        self.value = mlp((priv_dim,) + tuple(hidden), 1)
        self.act_dim = act_dim

    def dist(self, obs):
        loc, raw = torch.chunk(self.policy(obs), 2, dim=-1)
        return loc, F.softplus(raw) + 1e-3

    @staticmethod
    def log_prob(loc, scale, pre_tanh):
        # tanh-normal: log N(u; loc, scale) - sum log(1 - tanh(u)^2), numerically stable form
        lp = -0.5 * ((pre_tanh - loc) / scale) ** 2 - torch.log(scale) - 0.5 * math.log(2 * math.pi)
        lp = lp - 2.0 * (math.log(2.0) - pre_tanh - F.softplus(-2.0 * pre_tanh))
        return lp.sum(-1)

    @staticmethod
    def entropy(loc, scale, sample):
        ent = 0.5 + 0.5 * math.log(2 * math.pi) + torch.log(scale)
        ent = ent + 2.0 * (math.log(2.0) - sample - F.softplus(-2.0 * sample))      # E[log det] estimated at one sample (brax)
        return ent.sum(-1)


class _FusedPolicyLoss(torch.autograd.Function):
    """Policy part of the PPO loss through libpgtt's pgtt_ppo_policy_loss (one HIP launch + a finish instead of ~100
    elementwise launches forward and backward).  Same arithmetic as `_Learner._loss_torch`; tests/test_gpu_train.py compares."""

    @staticmethod
    def forward(ctx, out, u, logp_old, adv, eps, clip, cost):
        from . import native
        import ctypes
        out, u, logp_old, adv, eps = (t.contiguous() for t in (out, u, logp_old, adv, eps))
        B, A = u.shape
        grad = torch.empty_like(out)
        partial = torch.empty(2 * ((B + 63) // 64), dtype=torch.float32, device=out.device)
        loss = torch.empty(3, dtype=torch.float32, device=out.device)
        native.check(native.lib().pgtt_ppo_policy_loss(
            ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(u.data_ptr()), ctypes.c_void_p(logp_old.data_ptr()),
            ctypes.c_void_p(adv.data_ptr()), ctypes.c_void_p(eps.data_ptr()), ctypes.c_int(B), ctypes.c_int(A),
            ctypes.c_float(clip), ctypes.c_float(cost), ctypes.c_void_p(partial.data_ptr()), ctypes.c_void_p(loss.data_ptr()),
            ctypes.c_void_p(grad.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream(out.device).cuda_stream)))
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None, None, None


@dataclass
class PPOConfig:
    num_timesteps: int = 300_000_000
    num_evals: int = 31
    unroll_length: int = 20
    num_minibatches: int = 32
    batch_size: int = 256
    num_updates_per_batch: int = 4
    learning_rate: float = 3e-4
    discounting: float = 0.97
    gae_lambda: float = 0.95
    entropy_cost: float = 1e-2
    clipping_epsilon: float = 0.3
    max_grad_norm: float = 1.0
    reward_scaling: float = 1.0
    seed: int = 0


class _Actor:
    """T acting steps into preallocated [T, N, ...] storage, one acting step captured ONCE in a HIP graph and replayed (the write row
    is a device-side counter, so one graph serves every t).

    On a GPU the step is FOUR launches (round 4, acting.FusedActor): pgtt_policy_act (normalise, the 171-512-256-128-24 SiLU MLP on
    fp32 MFMA, tanh-normal sample, log-probability, tanh, the observation / action rows of the storage), the two kernels of env.step,
    pgtt_rollout_record (reward / done / truncation rows, finished-episode sums).  PGTT_PPO_ACT_FUSED=0 - and any env that is not a
    Joystick on a GPU (the CPU stub of the gloo tests) - takes the PyTorch-op form of the same step: ~60 small launches, library GEMMs."""

    def __init__(self, env, model, norm_s, T, cfg, L, acc, use_graph=True, ep_sums=None):
        self.env, self.model, self.norm_s, self.T, self.cfg, self.L = env, model, norm_s, T, cfg, L
        self.ep_ret_sum, self.ep_len_sum, self.ep_cnt, self.ep_metric_sum = acc
        dev, n = env.device, env.num_envs
        od, pd = env.observation_size["state"], env.observation_size["privileged_state"]
        z = lambda *sh: torch.zeros(*sh, device=dev)
        self.fused = None
        if (torch.device(dev).type == "cuda" and ep_sums is not None and hasattr(env, "env_id_offset") and os.environ.get("PGTT_PPO_ACT_FUSED", "1") != "0"
                and tuple(m.out_features for m in model.policy if isinstance(m, nn.Linear)) == (512, 256, 128, 2 * abi.NU)):
            from .acting import FusedActor
            self.fused = FusedActor(env, T, seed=cfg.seed + 7919 * _world()[1], reward_scaling=cfg.reward_scaling, episode_sums=ep_sums)
            self.S, self.act, self.t = self.fused.storage, self.fused.action, self.fused.counters[:1]
            self.fused.load_sequential(model.policy, norm_s.mean, norm_s.std)
        else:
            self.S = {"obs": z(T, n, od), "priv": z(T, n, pd), "u": z(T, n, abi.NU), "logp": z(T, n), "rew": z(T, n), "done": z(T, n), "trunc": z(T, n)}
            self.t = torch.zeros(1, dtype=torch.long, device=dev)
            self.act = z(n, abi.NU)
        self.graph = None
        if use_graph:
            try:
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    for _ in range(2):
                        self._step()                         # warm-up outside capture (allocator, lazy init)
                torch.cuda.current_stream(dev).wait_stream(side)
                self.t.zero_()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._step()
                self.graph = g
                self.t.zero_()
            except Exception as exc:                           # capture not available: plain launches
                print(f"[ppo] HIP graph capture of the acting step failed ({exc}); running eagerly")
                self.graph = None
                self.t.zero_()

    def _step(self):
        if self.fused is not None:
            self.fused.step()
            return
        env, S, t, cfg = self.env, self.S, self.t, self.cfg
        o, p = env.buffers["obs_state"], env.buffers["obs_priv"]
        loc, scale = self.model.dist(self.norm_s(o))
        u = loc + scale * torch.randn_like(loc)
        S["obs"].index_copy_(0, t, o[None]); S["priv"].index_copy_(0, t, p[None]); S["u"].index_copy_(0, t, u[None])
        S["logp"].index_copy_(0, t, self.model.log_prob(loc, scale, u)[None])
        self.act.copy_(torch.tanh(u))
        _, reward, done, info = env.step(self.act)
        fallen = env.buffers["frame"][abi.F_UPVECTOR + 2] < 0
        trunc = (env.buffers["istate"][abi.I_EP_STEPS] >= self.L) & ~fallen
        S["rew"].index_copy_(0, t, (reward * cfg.reward_scaling)[None]); S["done"].index_copy_(0, t, done[None]); S["trunc"].index_copy_(0, t, trunc.float()[None])
        epm = info["episode_metrics"]
        self.ep_ret_sum += (epm[abi.NMETRIC] * done).sum(); self.ep_len_sum += (epm[abi.NMETRIC + 1] * done).sum(); self.ep_cnt += done.sum()
        self.ep_metric_sum += (epm[:abi.NMETRIC] * done).sum(1)
        t += 1

    def rollout(self):
        self.t.zero_()
        if self.fused is not None:          # the learner has moved the weights and the statistics since the last roll-out: repack (in place)
            self.fused.load_sequential(self.model.policy, self.norm_s.mean, self.norm_s.std)
        for _ in range(self.T):
            if self.graph is not None:
                self.graph.replay()
            else:
                self._step()
        return self.S


class _Learner:
    """One PPO minibatch update (two MLPs forward / backward, clip, Adam: ~150 small launches) captured in a HIP graph
    after two eager updates; the minibatch is selected through a static index buffer."""

    def __init__(self, model, opt, norm_s, norm_p, B, mb, cfg, use_graph=True):
        self.model, self.opt, self.norm_s, self.norm_p, self.B, self.cfg = model, opt, norm_s, norm_p, B, cfg
        self.idx = torch.zeros(mb, dtype=torch.long, device=B["obs"].device)
        self.loss = torch.zeros((), device=B["obs"].device)
        self.use_graph, self.graph, self.calls = use_graph, None, 0
        # data parallel (one process per GPU; brax pmean-s the gradients over its devices): ONE flat fp32 bucket with every
        # gradient of both MLPs (0.53 M floats = 2.1 MB), one RCCL all-reduce per minibatch update.  The update's backward is
        # ~1 ms of launch-bound small kernels and the ring all-reduce of 2.1 MB over xGMI some tens of us, so the bucket is
        # not split to overlap with the backward.
        self.world, self.dp = _world()[0], _dp()
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.flat = torch.zeros(sum(p.numel() for p in self.params), device=B["obs"].device) if self.dp else None
        # the value branch (its own MLP, its own inputs) runs on a side stream next to the policy branch, forward and - since
        # autograd replays a node on the stream of its forward - backward: the ~150 launches of an update are 4-30 us each
        # and launch-latency bound, two independent chains overlap
        self.side = (torch.cuda.Stream(device=B["obs"].device)
                     if B["obs"].is_cuda and os.environ.get("PGTT_PPO_STREAMS", "2") == "2" else None)

    def _loss(self):
        """policy part of the loss: fused HIP kernel (PGTT_PPO_FUSED=0: the PyTorch-op form below)"""
        if os.environ.get("PGTT_PPO_FUSED", "1") == "0" or not self.B["obs"].is_cuda:
            return self._loss_torch()
        B, idx, cfg, model = self.B, self.idx, self.cfg, self.model
        out = model.policy(self.norm_s(B["obs"][idx]))
        a = B["adv"][idx]
        a = (a - a.mean()) / (a.std(unbiased=False) + 1e-8)        # jnp.std: population standard deviation
        eps = torch.randn(out.shape[0], out.shape[1] // 2, dtype=out.dtype, device=out.device)
        return _FusedPolicyLoss.apply(out, B["u"][idx], B["logp"][idx], a, eps, float(cfg.clipping_epsilon), float(cfg.entropy_cost))

    def _loss_torch(self):
        B, idx, cfg, model = self.B, self.idx, self.cfg, self.model
        loc, scale = model.dist(self.norm_s(B["obs"][idx]))
        logp = model.log_prob(loc, scale, B["u"][idx])
        a = B["adv"][idx]
        a = (a - a.mean()) / (a.std(unbiased=False) + 1e-8)        # jnp.std: population standard deviation
        ratio = torch.exp(logp - B["logp"][idx])
        pol = -torch.min(ratio * a, torch.clamp(ratio, 1 - cfg.clipping_epsilon, 1 + cfg.clipping_epsilon) * a).mean()
        ent = model.entropy(loc, scale, loc + scale * torch.randn_like(loc)).mean()
        return pol - cfg.entropy_cost * ent

    def _value_loss(self):
        B, idx = self.B, self.idx
        v = self.model.value(self.norm_p(B["priv"][idx])).squeeze(-1)
        return 0.5 * 0.5 * ((B["ret"][idx] - v) ** 2).mean()

    def _total_loss(self):
        if self.side is None:
            return self._loss() + self._value_loss()
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            v_loss = self._value_loss()
        pol = self._loss()
        main.wait_stream(self.side)
        return pol + v_loss

    def _forward_backward(self):
        loss = self._total_loss()
        loss.backward()
        self.loss.copy_(loss.detach())
        if self.dp:                        # pack every gradient into the one bucket the all-reduce works on
            torch.cat([p.grad.reshape(-1) for p in self.params], out=self.flat)

    def _all_reduce(self):
        if self.dp:
            dist.all_reduce(self.flat)

    def _apply(self):
        if self.dp:                        # mean of the ranks' gradients back into .grad, then the same clip + Adam in every rank
            self.flat.mul_(1.0 / self.world)
            grads = [p.grad for p in self.params]
            torch._foreach_copy_(grads, [v.view_as(g) for v, g in zip(self.flat.split([g.numel() for g in grads]), grads)])
        nn.utils.clip_grad_norm_(self.model.parameters(), self.cfg.max_grad_norm)
        self.opt.step()

    def _eager(self):
        self.opt.zero_grad(set_to_none=True)
        self._forward_backward()
        self._all_reduce()
        self._apply()

    def update(self, idx):
        self.idx.copy_(idx)
        self.calls += 1
        if self.use_graph and self.graph is None and self.calls == 3:        # two eager updates warmed everything up
            try:
                dev = self.idx.device
                torch.cuda.synchronize(dev)
                self.opt.zero_grad(set_to_none=True)
                if not self.dp:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self._forward_backward()
                        self._apply()
                    self.graph = (g,)
                else:
                    # the collective stays OUTSIDE the captured work: forward / backward / pack is one graph, unpack / clip / Adam
                    # a second one in the same memory pool (the gradients the first allocates are read by the second), the
                    # all-reduce an ordinary RCCL call between the two replays
                    ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                    with torch.cuda.graph(ga):
                        self._forward_backward()
                    with torch.cuda.graph(gb, pool=ga.pool()):
                        self._apply()
                    self.graph = (ga, gb)
            except Exception as exc:
                if self.dp:                 # a rank that alone drops to eager launches would still issue the same collectives,
                    raise                   # but a half-captured state is not worth continuing from in a multi-rank job
                print(f"[ppo] HIP graph capture of the update failed ({exc}); running eagerly")
                self.use_graph = False
        if self.graph is not None:
            self.graph[0].replay()
            if self.dp:
                self._all_reduce()
                self.graph[1].replay()
        else:
            self._eager()
        return self.loss


def train(env, cfg: PPOConfig, progress_fn: Optional[Callable[[int, Dict[str, float]], bool]] = None,
          policy_params_fn: Optional[Callable[[int, Dict], None]] = None, restore: Optional[Dict] = None, use_graph: bool = True):
    """PPO on a `Joystick` env created with autoreset=True.  Returns (model, normalisers, metrics history).

    Data parallel when a `torch.distributed` process group exists (one process per GPU, `torchrun train.py ...`): `env` is this
    rank's shard of the envs, `cfg.batch_size` the GLOBAL minibatch height (brax divides num_envs and batch_size by its device
    count the same way), every rank collects its own rollout and runs the same number of minibatch updates on 1 / world of
    each minibatch; gradients are averaged (one flat RCCL all-reduce per update), the observation statistics and the
    logged episode sums are summed over ranks, and every rank holds the same weights at every step.  `progress_fn` runs in
    every rank on identical numbers (its early-stop decision must not differ between ranks), `policy_params_fn` in rank 0."""
    dev = torch.device(env.device)
    on_gpu = dev.type == "cuda"
    use_graph = use_graph and on_gpu
    sync = (lambda: torch.cuda.synchronize(dev)) if on_gpu else (lambda: None)
    (world, rank), dp = _world(), _dp()
    n = env.num_envs
    torch.manual_seed(cfg.seed)
    od, pd = env.observation_size["state"], env.observation_size["privileged_state"]
    model = ActorCritic(od, pd).to(dev)
    norm_s, norm_p = RunningNorm(od, dev), RunningNorm(pd, dev)
    if restore is not None:
        model.load_state_dict(restore["model"])
        for nm, st in ((norm_s, restore["norm_state"]), (norm_p, restore["norm_priv"])):
            nm.count.copy_(st["count"].to(dev)); nm.mean.copy_(st["mean"].to(dev)); nm.m2.copy_(st["m2"].to(dev))
    if dp:
        for t in list(model.parameters()) + [norm_s.count, norm_s.mean, norm_s.m2, norm_p.count, norm_p.mean, norm_p.m2]:
            dist.broadcast(t.data, src=0)                         # one set of initial weights whatever the ranks' generators did
        if rank > 0:                                              # ... and independent exploration / minibatch noise per rank
            torch.manual_seed(cfg.seed + 1_000_003 * rank)        # (rank 0 keeps the stream a single-process run has)
    try:        # one fused multi-tensor Adam launch instead of ~10 foreach launches per update
        opt = torch.optim.Adam(model.parameters(), lr=cfg.learning_rate, capturable=use_graph, fused=on_gpu)
    except (RuntimeError, TypeError):
        opt = torch.optim.Adam(model.parameters(), lr=cfg.learning_rate, capturable=use_graph)
    assert (cfg.batch_size * cfg.num_minibatches) % (n * world) == 0, \
        "batch_size * num_minibatches must be a multiple of the total number of envs (num_envs per rank x world size)"
    unrolls = cfg.batch_size * cfg.num_minibatches // (n * world)
    T = unrolls * cfg.unroll_length
    steps_per_iter = T * n * world                                # env-steps of the whole job per iteration
    iters = max(1, math.ceil(cfg.num_timesteps / steps_per_iter))
    eval_every = max(1, iters // max(cfg.num_evals - 1, 1))
    L = env.config["episode_length"]
    obs = env.reset(seed=cfg.seed)
    actor, learner = None, None
    history, env_steps, t_env, t_sgd = [], 0, 0.0, 0.0
    # finished-episode sums since the last evaluation, ONE block [22 metric sums; return; length; count] (the layout pgtt_rollout_record adds to)
    ep_sums = torch.zeros(abi.NMETRIC + 3, device=dev)
    ep_metric_sum, ep_ret_sum, ep_len_sum, ep_cnt = ep_sums[:abi.NMETRIC], ep_sums[abi.NMETRIC], ep_sums[abi.NMETRIC + 1], ep_sums[abi.NMETRIC + 2]
    for it in range(iters):
        t0 = time.perf_counter()
        with torch.no_grad():
            if actor is None:
                actor = _Actor(env, model, norm_s, T, cfg, L, acc=(ep_ret_sum, ep_len_sum, ep_cnt, ep_metric_sum), use_graph=use_graph, ep_sums=ep_sums)
            batch = actor.rollout()
            obs = env._obs()
            last_priv = obs["privileged_state"].clone()
            norm_s.update(batch["obs"]); norm_p.update(batch["priv"])
            values = model.value(norm_p(torch.cat([batch["priv"], last_priv[None]], 0))).squeeze(-1)     # [T+1, N]
            # brax.training.agents.ppo.losses.compute_gae [UPSTREAM-RECALL]: termination = done * (1 - truncation); a TRUNCATED
            # step contributes no temporal-difference error and stops the backward recursion (the next observation belongs
            # to the next episode after AutoReset, its value is not a bootstrap for this one), a terminated step bootstraps with
            # 0; value targets vs = acc + V, advantages = (r + gamma (1 - term) vs[t+1] - V) (1 - trunc) with vs[T] = V(last obs)
            adv, ret = compute_gae(batch["trunc"], batch["done"] * (1.0 - batch["trunc"]), batch["rew"], values[:-1], values[-1],
                                   cfg.gae_lambda, cfg.discounting)
        sync(); t1 = time.perf_counter(); t_env += t1 - t0
        flat = lambda x: x.reshape(T * n, *x.shape[2:])
        if learner is None:
            B = {k: flat(batch[k]) for k in ("obs", "priv", "u", "logp")}          # views of the actor's static storage
            B["adv"], B["ret"] = torch.zeros(T * n, device=dev), torch.zeros(T * n, device=dev)
            learner = _Learner(model, opt, norm_s, norm_p, B, T * n // cfg.num_minibatches, cfg, use_graph=use_graph)
        learner.B["adv"].copy_(flat(adv)); learner.B["ret"].copy_(flat(ret))
        mb = T * n // cfg.num_minibatches
        for _ in range(cfg.num_updates_per_batch):
            perm = torch.randperm(T * n, device=dev)
            for k in range(cfg.num_minibatches):
                loss = learner.update(perm[k * mb:(k + 1) * mb])
        sync(); t_sgd += time.perf_counter() - t1
        env_steps += steps_per_iter
        if (it + 1) % eval_every == 0 or it == iters - 1:
            # one fused buffer: [episode return sum, length sum, count, 22 metric sums, loss, mean step reward, t_env, t_sgd],
            # summed over the ranks (the last four then divided by the world size) - every rank reports the same numbers
            log = torch.cat([torch.stack([ep_ret_sum, ep_len_sum, ep_cnt]), ep_metric_sum,
                             torch.stack([loss.detach().reshape(()), batch["rew"].mean()]),
                             torch.tensor([t_env, t_sgd], device=dev, dtype=ep_ret_sum.dtype)])
            if dp:
                dist.all_reduce(log)
                log[3 + abi.NMETRIC:] /= world
            log = log.tolist()
            c = max(log[2], 1.0)
            te, ts = (log[-2], log[-1]) if dp else (t_env, t_sgd)
            m = {"eval/episode_reward": log[0] / c, "eval/avg_episode_length": log[1] / c,
                 "episodes": log[2], "env_steps_per_s_rollout": (it + 1) * steps_per_iter / max(te, 1e-9),
                 "env_steps_per_s_total": (it + 1) * steps_per_iter / max(te + ts, 1e-9),
                 "loss": log[3 + abi.NMETRIC], "mean_step_reward": log[4 + abi.NMETRIC]}
            for i, k in enumerate(abi.REWARD_KEYS):
                m[f"eval/episode_reward/{k}"] = log[3 + i] / c
            history.append((env_steps, m))
            ep_sums.zero_()
            if policy_params_fn is not None and rank == 0:
                policy_params_fn(env_steps, checkpoint(model, norm_s, norm_p))
            if progress_fn is not None and progress_fn(env_steps, m):
                break
    return model, (norm_s, norm_p), history


def compute_gae(truncation: torch.Tensor, termination: torch.Tensor, rewards: torch.Tensor, values: torch.Tensor, bootstrap_value: torch.Tensor,
                lambda_: float, discount: float):
    """Generalised advantage estimation exactly as Brax's PPO computes it (brax/training/agents/ppo/losses.py compute_gae,
    [UPSTREAM-RECALL]); all inputs [T, N] except bootstrap_value [N].  Returns (advantages, value targets vs)."""
    mask = 1.0 - truncation
    nonterm = 1.0 - termination
    v_next = torch.cat([values[1:], bootstrap_value[None]], 0)
    deltas = (rewards + discount * nonterm * v_next - values) * mask
    carry = discount * lambda_ * nonterm * mask
    acc = torch.empty_like(deltas); last = torch.zeros_like(bootstrap_value)
    for t in reversed(range(deltas.shape[0])):
        last = torch.addcmul(deltas[t], carry[t], last, out=acc[t])                             # one launch per row
    vs = acc + values
    vs_next = torch.cat([vs[1:], bootstrap_value[None]], 0)
    adv = (rewards + discount * nonterm * vs_next - values) * mask
    return adv, vs


def checkpoint(model, norm_s, norm_p) -> Dict:
    st = lambda nm: {"count": nm.count.clone(), "mean": nm.mean.clone(), "m2": nm.m2.clone()}
    return {"model": {k: v.detach().clone() for k, v in model.state_dict().items()}, "norm_state": st(norm_s), "norm_priv": st(norm_p)}


def export_policy_npz(ckpt: Dict, path: str) -> None:
    """Write a checkpoint of `train` in the layout `policy.PolicyMLP` / tools/rollout_policy.py load - the role of the pickled
    `policy{index}` file the reference writes next to each Orbax checkpoint (training/train.py:189-195, read by
    deploy/policy_net.py:6-33): mean / std of the "state" normaliser, w{i} as [in, out] (Flax `kernel`), b{i}."""
    import numpy as np
    ns = ckpt["norm_state"]
    cnt = float(ns["count"])
    var = (ns["m2"].double() / max(cnt, 1.0)).clamp(min=1e-12)
    std = torch.sqrt(var).clamp(min=1e-6) if cnt > 0 else torch.ones_like(var)
    out = {"mean": ns["mean"].float().cpu().numpy(), "std": std.float().cpu().numpy()}
    keys = sorted({int(k.split(".")[1]) for k in ckpt["model"] if k.startswith("policy.") and k.endswith(".weight")})
    for i, li in enumerate(keys):
        out[f"w{i}"] = ckpt["model"][f"policy.{li}.weight"].float().cpu().numpy().T.copy()
        out[f"b{i}"] = ckpt["model"][f"policy.{li}.bias"].float().cpu().numpy()
    np.savez_compressed(path, **out)
