"""The acting step of a roll-out as two launches of libpgtt.so around `Joystick.step` (include/pgtt_train.h, csrc/pgtt_policy.hip):

    actor = FusedActor(env, T=20, seed=0)
    actor.load(policy_layers, mean, std)          # [(W [out, in], b [out])] x 4, observation statistics
    for t in range(T):
        actor.act(); env.step(actor.action); actor.record()
    actor.storage["obs" | "priv" | "u" | "logp" | "rew" | "done" | "trunc"]      # [T, N, ...]

What the reference gets from Brax's `generate_unroll` with the networks of training/train.py:135-161 (the deployed form of the same
policy network is deploy/policy_net.py:36-71): normalise -> 171-512-256-128-24 SiLU MLP -> tanh-normal head -> sample / log-probability ->
env.step -> reward / done / truncation / finished-episode sums.  This is the CALLER of the hot path; nothing here is
needed to step the env.  There is no CPU path: the class refuses non-cuda envs.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import abi, native

_f, _i32, _p = C.c_float, C.c_int32, C.c_void_p


class PgttPolicyActArgs(C.Structure):
    _fields_ = [("obs", _p), ("priv", _p), ("mean", _p), ("std", _p), ("w", _p * 4), ("b", _p * 4), ("eps", _p), ("act", _p), ("head", _p),
                ("store_obs", _p), ("store_priv", _p), ("store_u", _p), ("store_logp", _p), ("counters", _p), ("seed", C.c_uint64),
                ("env_id_offset", C.c_int64), ("num_envs", _i32), ("obs_dim", _i32), ("priv_dim", _i32), ("deterministic", _i32), ("store_rows", _i32)]


class PgttRolloutRecordArgs(C.Structure):
    _fields_ = [("reward", _p), ("done", _p), ("ep_steps", _p), ("up_z", _p), ("ep_metrics", _p), ("store_rew", _p), ("store_done", _p),
                ("store_trunc", _p), ("counters", _p), ("episode_sums", _p), ("reward_scaling", _f),
                ("num_envs", _i32), ("episode_length", _i32), ("store_rows", _i32)]


HIDDEN = (512, 256, 128)      # the kernel's layer widths = the reference's policy_hidden_layer_sizes (training/train.py:158)


def _lib():
    L = native.lib()
    if not getattr(L, "_acting_ready", False):
        L.pgtt_policy_act.argtypes = [C.POINTER(PgttPolicyActArgs), _p]
        L.pgtt_rollout_record.argtypes = [C.POINTER(PgttRolloutRecordArgs), _p]
        L.pgtt_policy_packed_floats.argtypes = [C.c_int, C.c_int]
        assert L.pgtt_sizeof_policy_act_args() == C.sizeof(PgttPolicyActArgs) and L.pgtt_sizeof_rollout_record_args() == C.sizeof(PgttRolloutRecordArgs)
        L._acting_ready = True
    return L


def pack_linear(w: torch.Tensor, b: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """torch Linear weight [out, in] -> the MFMA tile order of pgtt_train.h: zero-padded to multiples of 16 and laid out as
    [out / 16][in / 16][g][i][s] = W[16 tile + i][16 kb + 4 g + s]; the bias zero-padded.  A handful of torch ops on the weight's device."""
    n, k = w.shape
    npad, kpad = -(-n // 16) * 16, -(-k // 16) * 16
    wp = F.pad(w.detach().float(), (0, kpad - k, 0, npad - n))
    packed = wp.view(npad // 16, 16, kpad // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous().reshape(-1)
    return packed, F.pad(b.detach().float(), (0, npad - n)).contiguous()


class FusedActor:
    """Roll-out storage + the two acting kernels for one `Joystick` (autoreset=True)."""

    def __init__(self, env, T: int, seed: int = 0, reward_scaling: float = 1.0, episode_sums: Optional[torch.Tensor] = None):
        if env.device.type != "cuda":
            raise native.PgttError("FusedActor needs a Joystick on a ROCm device; there is no CPU path")
        self.env, self.T = env, int(T)
        self._L = _lib()
        dev, n = env.device, env.num_envs
        od, pd = env.observation_size["state"], env.observation_size["privileged_state"]
        self.od, self.pd = od, pd
        z = lambda *sh: torch.zeros(*sh, device=dev)
        self.storage: Dict[str, torch.Tensor] = {"obs": z(T, n, od), "priv": z(T, n, pd), "u": z(T, n, abi.NU), "logp": z(T, n), "rew": z(T, n),
                                                  "done": z(T, n), "trunc": z(T, n)}
        self.action = z(n, abi.NU)
        self.counters = torch.zeros(2, dtype=torch.int64, device=dev)            # {storage row, draw counter}
        self.episode_sums = z(abi.NMETRIC + 3) if episode_sums is None else episode_sums      # 22 metric sums, return, length, count
        assert self.episode_sums.numel() == abi.NMETRIC + 3 and self.episode_sums.is_contiguous()
        self.mean, self.std = z(od), torch.ones(od, device=dev)
        dims = (od,) + HIDDEN + (2 * abi.NU,)
        self._w = [z(self._L.pgtt_policy_packed_floats(dims[l], dims[l + 1])) for l in range(4)]
        self._b = [z(-(-dims[l + 1] // 16) * 16) for l in range(4)]
        self._dims = dims
        a = PgttPolicyActArgs()
        a.obs, a.priv = env.buffers["obs_state"].data_ptr(), env.buffers["obs_priv"].data_ptr()
        a.mean, a.std = self.mean.data_ptr(), self.std.data_ptr()
        for l in range(4):
            a.w[l], a.b[l] = self._w[l].data_ptr(), self._b[l].data_ptr()
        a.eps, a.head, a.act = None, None, self.action.data_ptr()
        S = self.storage
        a.store_obs, a.store_priv, a.store_u, a.store_logp = S["obs"].data_ptr(), S["priv"].data_ptr(), S["u"].data_ptr(), S["logp"].data_ptr()
        a.counters, a.seed, a.env_id_offset = self.counters.data_ptr(), int(seed) & 0xFFFFFFFFFFFFFFFF, int(env.env_id_offset)
        a.num_envs, a.obs_dim, a.priv_dim, a.deterministic, a.store_rows = n, od, pd, 0, self.T
        self._act_args = a
        r = PgttRolloutRecordArgs()
        r.reward, r.done = env.buffers["reward"].data_ptr(), env.buffers["done"].data_ptr()
        r.ep_steps = env.buffers["istate"][abi.I_EP_STEPS].data_ptr()
        r.up_z = env.buffers["frame"][abi.F_UPVECTOR + 2].data_ptr()
        r.ep_metrics = env.buffers["ep_metrics"].data_ptr()
        r.store_rew, r.store_done, r.store_trunc = S["rew"].data_ptr(), S["done"].data_ptr(), S["trunc"].data_ptr()
        r.counters, r.episode_sums = self.counters.data_ptr(), self.episode_sums.data_ptr()
        r.reward_scaling, r.num_envs, r.episode_length, r.store_rows = float(reward_scaling), n, int(env.config["episode_length"]), self.T
        self._rec_args = r

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.env.device).cuda_stream

    @torch.no_grad()
    def load(self, layers: Sequence[Tuple[torch.Tensor, torch.Tensor]], mean: torch.Tensor, std: torch.Tensor) -> None:
        """(re)pack the four (weight [out, in], bias [out]) pairs and the observation statistics into the buffers the kernels read
        (in place: a captured graph keeps reading the same addresses)"""
        assert len(layers) == 4
        for l, (w, b) in enumerate(layers):
            assert tuple(w.shape) == (self._dims[l + 1], self._dims[l]), (l, tuple(w.shape), self._dims)
            pw, pb = pack_linear(w.to(self.mean.device), b.to(self.mean.device))
            self._w[l].copy_(pw); self._b[l].copy_(pb)
        self.mean.copy_(mean); self.std.copy_(std)

    def load_sequential(self, seq: torch.nn.Sequential, mean: torch.Tensor, std: torch.Tensor) -> None:
        lins = [m for m in seq if isinstance(m, torch.nn.Linear)]
        self.load([(m.weight, m.bias) for m in lins], mean, std)

    def act(self, deterministic: bool = False, eps: Optional[torch.Tensor] = None, head: Optional[torch.Tensor] = None, store: bool = True) -> torch.Tensor:
        """policy forward + sample for the env's current observation -> self.action ([N, 12], also returned); storage row counters[0]"""
        a = self._act_args
        a.deterministic = int(deterministic)
        a.eps = None if eps is None else eps.data_ptr()
        a.head = None if head is None else head.data_ptr()
        if eps is not None:
            assert eps.is_contiguous() and eps.dtype == torch.float32 and tuple(eps.shape) == (self.env.num_envs, abi.NU)
        S = self.storage
        a.store_obs, a.store_priv, a.store_u, a.store_logp = ((S["obs"].data_ptr(), S["priv"].data_ptr(), S["u"].data_ptr(), S["logp"].data_ptr())
                                                              if store else (None, None, None, None))
        native.check(self._L.pgtt_policy_act(C.byref(a), self._stream()))
        return self.action

    def record(self) -> None:
        """after env.step: reward / done / truncation into storage row counters[0], finished episodes into episode_sums, counters advanced"""
        native.check(self._L.pgtt_rollout_record(C.byref(self._rec_args), self._stream()))

    def step(self) -> None:
        self.act()
        self.env.step(self.action)
        self.record()

    def rewind(self) -> None:
        """storage row back to 0 (the draw counter keeps running)"""
        self.counters[0].zero_()
