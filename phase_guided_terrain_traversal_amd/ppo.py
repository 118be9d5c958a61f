"""PPO learner around the env (N1 of SURVEY 8f): a torch restatement of what the reference gets from
`brax.training.agents.ppo.train` with the settings of training/train.py:135-161:

    unroll_length 20, num_minibatches 32, batch_size 256, num_updates_per_batch 4, lr 3e-4, discounting 0.97,
    entropy_cost 1e-2, max_grad_norm 1.0, reward_scaling 1.0, normalize_observations, policy / value MLPs
    (512, 256, 128) with swish, policy obs key "state" (171), value obs key "privileged_state" (215).

Brax defaults it relies on [UPSTREAM-RECALL]: tanh-normal action distribution with scale = softplus(raw) + 1e-3,
clipping_epsilon 0.3, gae_lambda 0.95, normalize_advantage, value loss 0.5 * 0.5 * mse, bootstrap on truncation
with V(next_obs).  This is the CALLER of the hot path (out of the §8 a-e scope); it exists so that env-steps/s can be
measured inside a real training loop and `train.py` keeps the reference's CLI.  Dense layers are plain library GEMMs
(torch -> hipBLASLt); nothing here is a custom kernel.
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass
from typing import Callable, Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import abi


class RunningNorm:
    """Welford running mean/std per observation key (brax running_statistics), count starts at 0, std floor 1e-6."""

    def __init__(self, dim: int, device):
        self.count = torch.zeros((), device=device, dtype=torch.float64)
        self.mean = torch.zeros(dim, device=device)
        self.m2 = torch.zeros(dim, device=device)

    @torch.no_grad()
    def update(self, x: torch.Tensor) -> None:
        x = x.reshape(-1, x.shape[-1])
        n = x.shape[0]
        new_count = self.count + n
        delta = x.mean(0) - self.mean
        batch_m2 = ((x - x.mean(0)) ** 2).sum(0)
        self.m2 += batch_m2 + delta ** 2 * float(self.count) * n / float(new_count)
        self.mean += delta * n / float(new_count)
        self.count = new_count

    @property
    def std(self) -> torch.Tensor:
        var = self.m2 / max(float(self.count), 1.0)
        return torch.sqrt(torch.clamp(var, min=1e-12)).clamp(min=1e-6)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return (x - self.mean) / self.std


def mlp(sizes, out):
    layers, d = [], sizes[0]
    for h in sizes[1:]:
        layers += [nn.Linear(d, h), nn.SiLU()]
        d = h
    layers.append(nn.Linear(d, out))
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


def train(env, cfg: PPOConfig, progress_fn: Optional[Callable[[int, Dict[str, float]], bool]] = None,
          policy_params_fn: Optional[Callable[[int, Dict], None]] = None, restore: Optional[Dict] = None):
    """PPO on a `Joystick` env created with autoreset=True.  Returns (model, normalisers, metrics history)."""
    dev = env.device
    n = env.num_envs
    torch.manual_seed(cfg.seed)
    od, pd = env.observation_size["state"], env.observation_size["privileged_state"]
    model = ActorCritic(od, pd).to(dev)
    norm_s, norm_p = RunningNorm(od, dev), RunningNorm(pd, dev)
    if restore is not None:
        model.load_state_dict(restore["model"])
        for nm, st in ((norm_s, restore["norm_state"]), (norm_p, restore["norm_priv"])):
            nm.count, nm.mean, nm.m2 = st["count"].to(dev), st["mean"].to(dev), st["m2"].to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=cfg.learning_rate)
    assert (cfg.batch_size * cfg.num_minibatches) % n == 0, "batch_size * num_minibatches must be a multiple of num_envs"
    unrolls = cfg.batch_size * cfg.num_minibatches // n
    T = unrolls * cfg.unroll_length
    steps_per_iter = T * n
    iters = max(1, math.ceil(cfg.num_timesteps / steps_per_iter))
    eval_every = max(1, iters // max(cfg.num_evals - 1, 1))
    L = env.config["episode_length"]
    obs = env.reset(seed=cfg.seed)
    history, env_steps, t_env, t_sgd = [], 0, 0.0, 0.0
    ep_ret_sum = torch.zeros((), device=dev); ep_len_sum = torch.zeros((), device=dev); ep_cnt = torch.zeros((), device=dev)
    ep_metric_sum = torch.zeros(abi.NMETRIC, device=dev)
    for it in range(iters):
        t0 = time.perf_counter()
        S = {k: [] for k in ("obs", "priv", "u", "logp", "rew", "done", "trunc", "val")}
        with torch.no_grad():
            for t in range(T):
                o, p = obs["state"].clone(), obs["privileged_state"].clone()
                loc, scale = model.dist(norm_s(o))
                u = loc + scale * torch.randn_like(loc)
                obs, reward, done, info = env.step(torch.tanh(u))
                fallen = env.buffers["frame"][abi.F_UPVECTOR + 2] < 0
                trunc = (env.buffers["istate"][abi.I_EP_STEPS] >= L) & ~fallen
                S["obs"].append(o); S["priv"].append(p); S["u"].append(u); S["logp"].append(model.log_prob(loc, scale, u))
                S["rew"].append(reward.clone() * cfg.reward_scaling); S["done"].append(done.clone()); S["trunc"].append(trunc.float())
                d = done.bool()
                epm = info["episode_metrics"]
                ep_ret_sum += (epm[abi.NMETRIC] * d).sum(); ep_len_sum += (epm[abi.NMETRIC + 1] * d).sum(); ep_cnt += d.sum()
                ep_metric_sum += (epm[:abi.NMETRIC] * d).sum(1)
            last_priv = obs["privileged_state"].clone()
            batch = {k: torch.stack(v) for k, v in S.items() if v}
            norm_s.update(batch["obs"]); norm_p.update(batch["priv"])
            values = model.value(norm_p(torch.cat([batch["priv"], last_priv[None]], 0))).squeeze(-1)     # [T+1, N]
            # with AutoReset the obs after a done is the first obs of the env: V(next) at a truncation is V(first obs),
            # exactly as in brax's acting loop (Transition.next_observation = nstate.obs)
            term = batch["done"] * (1.0 - batch["trunc"])
            adv = torch.zeros_like(batch["rew"]); last = torch.zeros(n, device=dev)
            for t in reversed(range(T)):
                nonterm = 1.0 - term[t]
                delta = batch["rew"][t] + cfg.discounting * values[t + 1] * nonterm - values[t]
                last = delta + cfg.discounting * cfg.gae_lambda * nonterm * (1.0 - batch["done"][t]) * last
                adv[t] = last
            ret = adv + values[:-1]
        torch.cuda.synchronize(dev); t1 = time.perf_counter(); t_env += t1 - t0
        flat = lambda x: x.reshape(T * n, *x.shape[2:])
        B = {k: flat(batch[k]) for k in ("obs", "priv", "u", "logp")}
        B["adv"], B["ret"] = flat(adv), flat(ret)
        mb = T * n // cfg.num_minibatches
        for _ in range(cfg.num_updates_per_batch):
            perm = torch.randperm(T * n, device=dev)
            for k in range(cfg.num_minibatches):
                idx = perm[k * mb:(k + 1) * mb]
                loc, scale = model.dist(norm_s(B["obs"][idx]))
                logp = model.log_prob(loc, scale, B["u"][idx])
                a = B["adv"][idx]
                a = (a - a.mean()) / (a.std() + 1e-8)
                ratio = torch.exp(logp - B["logp"][idx])
                pol = -torch.min(ratio * a, torch.clamp(ratio, 1 - cfg.clipping_epsilon, 1 + cfg.clipping_epsilon) * a).mean()
                v = model.value(norm_p(B["priv"][idx])).squeeze(-1)
                v_loss = 0.5 * 0.5 * ((B["ret"][idx] - v) ** 2).mean()
                ent = model.entropy(loc, scale, loc + scale * torch.randn_like(loc)).mean()
                loss = pol + v_loss - cfg.entropy_cost * ent
                opt.zero_grad(set_to_none=True)
                loss.backward()
                nn.utils.clip_grad_norm_(model.parameters(), cfg.max_grad_norm)
                opt.step()
        torch.cuda.synchronize(dev); t_sgd += time.perf_counter() - t1
        env_steps += steps_per_iter
        if (it + 1) % eval_every == 0 or it == iters - 1:
            c = max(float(ep_cnt), 1.0)
            m = {"eval/episode_reward": float(ep_ret_sum) / c, "eval/avg_episode_length": float(ep_len_sum) / c,
                 "episodes": float(ep_cnt), "env_steps_per_s_rollout": (it + 1) * steps_per_iter / max(t_env, 1e-9),
                 "env_steps_per_s_total": (it + 1) * steps_per_iter / max(t_env + t_sgd, 1e-9),
                 "loss": float(loss.detach()), "mean_step_reward": float(batch["rew"].mean())}
            for i, k in enumerate(abi.REWARD_KEYS):
                m[f"eval/episode_reward/{k}"] = float(ep_metric_sum[i]) / c
            history.append((env_steps, m))
            ep_ret_sum.zero_(); ep_len_sum.zero_(); ep_cnt.zero_(); ep_metric_sum.zero_()
            if policy_params_fn is not None:
                policy_params_fn(env_steps, checkpoint(model, norm_s, norm_p))
            if progress_fn is not None and progress_fn(env_steps, m):
                break
    return model, (norm_s, norm_p), history


def checkpoint(model, norm_s, norm_p) -> Dict:
    st = lambda nm: {"count": nm.count.clone(), "mean": nm.mean.clone(), "m2": nm.m2.clone()}
    return {"model": {k: v.detach().clone() for k, v in model.state_dict().items()}, "norm_state": st(norm_s), "norm_priv": st(norm_p)}
