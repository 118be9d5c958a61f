"""The acting step of a roll-out as two launches of libpgtt.so (include/pgtt_train.h, csrc/pgtt_policy.hip; acting.FusedActor) against the
reference's own policy forward pass (tests/golden/policy_forward.npz = deploy/policy_net.py:6-80 run on policy_folder/policy177 / policy3)
and against the PyTorch-op form of the same step (ppo.ActorCritic, the Brax networks of training/train.py:135-161)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from phase_guided_terrain_traversal_amd import abi, configs, policy, ppo
from phase_guided_terrain_traversal_amd.acting import FusedActor, pack_linear
from phase_guided_terrain_traversal_amd.env import Joystick


def _env(n, method="pgtt", **kw):
    return Joystick("flat_terrain", configs.training_config(method), num_envs=n, device="cuda:0", autoreset=True, **kw)


@pytest.mark.parametrize("name", ["policy177", "policy3"])
def test_fused_policy_forward_matches_the_reference_fixture(golden_dir, name):
    """the kernel's tanh(loc) on the 64 fixture observations equals the reference's policy_net() output to 1e-5 (policy.PolicyMLP in torch fp32
    meets 1e-6 on the same vectors) - also with more envs than observations (the rest of the batch holds other rows) and a ragged env count"""
    g = np.load(os.path.join(golden_dir, "policy_forward.npz"))
    net = policy.load_policy(name, device="cuda:0")
    obs = torch.from_numpy(g[f"{name}_obs"]).cuda()
    for n in (64, 77):
        env = _env(n)
        fa = FusedActor(env, T=1)
        fa.load([(l.weight, l.bias) for l in net.layers], net.mean, net.std)
        env.buffers["obs_state"].normal_()
        env.buffers["obs_state"][:64].copy_(obs)
        head = torch.zeros(n, 24, device="cuda")
        act = fa.act(deterministic=True, head=head, store=False)
        torch.cuda.synchronize()
        assert np.abs(act[:64].cpu().numpy() - g[f"{name}_action"]).max() < 1e-5
        want = net.head(env.buffers["obs_state"])
        assert float((head - want).abs().max()) < 2e-5 * (1 + float(want.abs().max()))
        env.close()


@pytest.mark.parametrize("n,method", [(4096, "pgtt"), (1000, "pgtt"), (333, "baseline")])
def test_fused_sample_and_log_prob_match_the_torch_ops(n, method):
    """given the SAME standard-normal draws: pre-tanh sample, log-probability, action and the storage rows of the step equal what
    ppo._Actor's PyTorch-op form computes (random network, running statistics far from (0, 1), storage row 3 of 5)"""
    env = _env(n, method)
    env.reset(seed=2)
    od = env.observation_size["state"]
    torch.manual_seed(5)
    model = ppo.ActorCritic(od, env.observation_size["privileged_state"]).cuda()
    with torch.no_grad():
        for p in model.policy.parameters():
            p.mul_(1.7)
    mean, std = torch.randn(od, device="cuda") * 0.5, torch.rand(od, device="cuda") * 2 + 0.05
    env.buffers["obs_state"].mul_(3.0).add_(0.3)
    fa = FusedActor(env, T=5)
    fa.load_sequential(model.policy, mean, std)
    fa.counters[0] = 3
    eps = torch.randn(n, 12, device="cuda")
    head = torch.zeros(n, 24, device="cuda")
    act = fa.act(eps=eps, head=head).clone()
    torch.cuda.synchronize()
    with torch.no_grad():
        o = env.buffers["obs_state"]
        loc, scale = model.dist((o - mean) / std)
        u = loc + scale * eps
        lp = model.log_prob(loc, scale, u)
    S = fa.storage
    assert float((head[:, :12] - loc).abs().max()) < 2e-5 * (1 + float(loc.abs().max()))
    assert float((S["u"][3] - u).abs().max()) < 2e-5 * (1 + float(u.abs().max()))
    assert float((act - torch.tanh(u)).abs().max()) < 2e-5
    assert float(((S["logp"][3] - lp).abs() / (1 + lp.abs())).max()) < 1e-4
    assert torch.equal(S["obs"][3], o) and torch.equal(S["priv"][3], env.buffers["obs_priv"])
    assert float(S["obs"][2].abs().sum()) == 0.0 and float(S["obs"][4].abs().sum()) == 0.0 and float(S["u"][4].abs().sum()) == 0.0
    env.close()


def test_in_kernel_draws_are_standard_normal_and_keyed_by_global_env_and_step():
    """eps = (u - loc) / scale recovered from the kernel's own Philox draws: N(0, 1) moments, different per step, and a shard at a global
    env-id offset reproduces the slice of the full batch (draws are keyed by global ids like the env's own)"""
    n = 4096
    env = _env(n)
    env.reset(seed=1)
    torch.manual_seed(0)
    model = ppo.ActorCritic(env.observation_size["state"], env.observation_size["privileged_state"]).cuda()
    mean, std = torch.zeros(171, device="cuda"), torch.ones(171, device="cuda")

    def draws(e, steps):
        fa = FusedActor(e, T=steps, seed=11)
        fa.load_sequential(model.policy, mean, std)
        out = []
        for _ in range(steps):
            head = torch.zeros(e.num_envs, 24, device="cuda")
            fa.act(head=head)
            sc = torch.nn.functional.softplus(head[:, 12:]) + 1e-3
            out.append(((fa.storage["u"][int(fa.counters[0])] - head[:, :12]) / sc).clone())
            fa.record()
        return torch.stack(out)
    E = draws(env, 4)
    torch.cuda.synchronize()
    assert abs(float(E.mean())) < 0.01 and abs(float(E.std()) - 1.0) < 0.01
    assert abs(float((E ** 3).mean())) < 0.03 and abs(float((E ** 4).mean()) - 3.0) < 0.1
    assert float((E[0] - E[1]).abs().min()) > 0 and abs(float((E[0] * E[1]).mean())) < 0.02          # fresh draws every step
    assert abs(float((E[:, :, 0] * E[:, :, 1]).mean())) < 0.02                                       # the two halves of a Box-Muller pair
    part = _env(512, env_id_offset=1024)
    part.reset(seed=1)
    part.buffers["obs_state"].copy_(env.buffers["obs_state"][1024:1536])
    P = draws(part, 2)
    assert float((P - E[:2, 1024:1536]).abs().max()) < 1e-5
    env.close(); part.close()


def test_rollout_record_matches_the_torch_bookkeeping():
    """reward / done / truncation rows and the finished-episode sums of pgtt_rollout_record against the PyTorch-op form of ppo._Actor over a
    roll-out with short episodes (AutoReset and truncation both occur), ragged env count, reward scaling"""
    n, T, L = 1000, 40, 13
    env = Joystick("flat_terrain", configs.with_overrides(configs.training_config(), episode_length=L), num_envs=n, device="cuda:0", autoreset=True)
    env.reset(seed=3)
    torch.manual_seed(1)
    model = ppo.ActorCritic().cuda()
    fa = FusedActor(env, T=T, seed=2, reward_scaling=0.5)
    fa.load_sequential(model.policy, torch.zeros(171, device="cuda"), torch.ones(171, device="cuda"))
    ref = {k: torch.zeros(T, n, device="cuda") for k in ("rew", "done", "trunc")}
    sums = torch.zeros(abi.NMETRIC + 3, device="cuda", dtype=torch.float64)
    for t in range(T):
        fa.act()
        _, reward, done, info = env.step(fa.action * 3.0)            # wild actions: robots fall, episodes end early as well
        fallen = env.buffers["frame"][abi.F_UPVECTOR + 2] < 0
        trunc = (env.buffers["istate"][abi.I_EP_STEPS] >= L) & ~fallen
        ref["rew"][t], ref["done"][t], ref["trunc"][t] = reward * 0.5, done, trunc.float()
        epm = info["episode_metrics"].double()
        sums[:abi.NMETRIC] += (epm[:abi.NMETRIC] * done).sum(1); sums[abi.NMETRIC] += (epm[abi.NMETRIC] * done).sum()
        sums[abi.NMETRIC + 1] += (epm[abi.NMETRIC + 1] * done).sum(); sums[abi.NMETRIC + 2] += done.sum()
        fa.record()
    torch.cuda.synchronize()
    assert int(fa.counters[0]) == T and int(fa.counters[1]) == T
    for k in ("rew", "done", "trunc"):
        assert torch.equal(fa.storage[k], ref[k]), k
    assert float(ref["trunc"].sum()) > 0 and float(ref["done"].sum()) > float(ref["trunc"].sum())
    assert torch.allclose(fa.episode_sums.double(), sums, rtol=2e-5, atol=1e-4), (fa.episode_sums, sums)
    fa.rewind()
    assert int(fa.counters[0]) == 0 and int(fa.counters[1]) == T
    env.close()


def test_pack_linear_layout():
    """[out / 16][in / 16][g][i][s] = W[16 tile + i][16 kb + 4 g + s], zero padding"""
    w = torch.arange(24 * 171, dtype=torch.float32, device="cuda").reshape(24, 171) + 1
    p, b = pack_linear(w, torch.ones(24, device="cuda"))
    assert p.numel() == 32 * 176 and b.numel() == 32 and float(b[24:].abs().sum()) == 0
    P = p.view(2, 11, 4, 16, 4).cpu()
    for tile, kb, g, i, s in ((0, 0, 0, 0, 0), (1, 10, 2, 7, 2), (0, 5, 3, 15, 3), (1, 10, 3, 7, 3)):
        n_, k_ = 16 * tile + i, 16 * kb + 4 * g + s
        want = float(w[n_, k_]) if n_ < 24 and k_ < 171 else 0.0
        assert float(P[tile, kb, g, i, s]) == want, (tile, kb, g, i, s)


def test_fused_and_torch_acting_steps_train_alike():
    """ppo.train with the fused acting step against PGTT_PPO_ACT_FUSED=0 (PyTorch ops): different random streams, same learning problem -
    both runs stop the robot from falling within the same budget and report finite, comparable numbers"""
    res = {}
    for mode in ("0", "1"):
        os.environ["PGTT_PPO_ACT_FUSED"] = mode
        try:
            env = _env(2048)
            cfg = ppo.PPOConfig(num_timesteps=3_000_000, num_evals=4, seed=1)
            model, norms, hist = ppo.train(env, cfg)
            res[mode] = hist
            env.close()
        finally:
            os.environ.pop("PGTT_PPO_ACT_FUSED", None)
    for mode, hist in res.items():
        first, last = hist[0][1], hist[-1][1]
        print("fused" if mode == "1" else "torch", [(s, round(m["eval/avg_episode_length"], 1), round(m["eval/episode_reward"], 3), round(m["env_steps_per_s_rollout"] / 1e6, 2)) for s, m in hist])
        assert last["eval/avg_episode_length"] > 2 * first["eval/avg_episode_length"]
    a, b = res["1"][-1][1], res["0"][-1][1]
    assert 0.5 < a["eval/avg_episode_length"] / b["eval/avg_episode_length"] < 2.0


def test_fused_acting_step_is_faster_than_the_torch_ops():
    """ppo._Actor.rollout() on level4 at 4096 envs, both forms captured in a HIP graph, timed warm: the four-launch step against ~60 launches"""
    import time
    terrain = np.load(os.path.join(os.path.dirname(abi.__file__), "assets", "terrains", "level4.npy"))
    from phase_guided_terrain_traversal_amd import mjcf
    from phase_guided_terrain_traversal_amd.randomize import domain_randomize
    rate = {}
    for mode in ("0", "1"):
        os.environ["PGTT_PPO_ACT_FUSED"] = mode
        try:
            n, T = 4096, 40
            variant = torch.from_numpy(domain_randomize(mjcf.load_model("stairs"), n, seed=2, terrain=terrain, enable=False)["variant"])
            env = Joystick("stairs", configs.training_config(), num_envs=n, terrain=terrain, device="cuda:0", autoreset=True, variant=variant)
            env.reset(seed=0)
            torch.manual_seed(0)
            model = ppo.ActorCritic().cuda()
            norm = ppo.RunningNorm(171, "cuda")
            ep = torch.zeros(abi.NMETRIC + 3, device="cuda")
            with torch.no_grad():
                actor = ppo._Actor(env, model, norm, T, ppo.PPOConfig(seed=1), 1000, acc=(ep[abi.NMETRIC], ep[abi.NMETRIC + 1], ep[abi.NMETRIC + 2], ep[:abi.NMETRIC]), ep_sums=ep)
                assert (actor.fused is not None) == (mode == "1") and actor.graph is not None
                for _ in range(3):
                    actor.rollout()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5):
                    actor.rollout()
                torch.cuda.synchronize(); rate[mode] = 5 * T * n / (time.perf_counter() - t0)
            assert torch.isfinite(actor.S["logp"]).all() and float(actor.S["u"].abs().sum()) > 0
            env.close()
        finally:
            os.environ.pop("PGTT_PPO_ACT_FUSED", None)
    # informational only (a shared or down-clocked GPU must not fail a correctness suite): the throughput guard is bench.py's `rollout` row,
    # asserted by tests/test_gpu_bench.py with a floor far below the measured 18 - 19 M
    print("acting step, env-steps/s: torch ops %.2f M, fused %.2f M" % (rate["0"] / 1e6, rate["1"] / 1e6))


def test_acting_kernels_do_not_store_past_their_storage():
    """a caller that runs past its T storage rows (warm-up steps before a rewind, a forgotten rewind) gets actions and counters but NO store beyond
    row T - 1: pgtt_policy_act / pgtt_rollout_record carry the row count (store_rows) - with unroll_length = 1 the two warm-up steps of ppo._Actor
    used to write rows 1 and 2 of a one-row storage"""
    n, T = 300, 2
    env = _env(n)
    env.reset(seed=4)
    net = policy.load_policy("policy177", device="cuda:0")
    fa = FusedActor(env, T=T)
    fa.load([(l.weight, l.bias) for l in net.layers], net.mean, net.std)
    # guard rows behind every storage block: the blocks are views into larger allocations filled with a sentinel
    guards = {}
    for k, v in list(fa.storage.items()):
        big = torch.full((T + 3,) + tuple(v.shape[1:]), 777.0, device="cuda")
        guards[k] = big
        fa.storage[k] = big[:T]
    S = fa.storage
    a, r = fa._act_args, fa._rec_args
    a.store_obs, a.store_priv, a.store_u, a.store_logp = S["obs"].data_ptr(), S["priv"].data_ptr(), S["u"].data_ptr(), S["logp"].data_ptr()
    r.store_rew, r.store_done, r.store_trunc = S["rew"].data_ptr(), S["done"].data_ptr(), S["trunc"].data_ptr()
    for _ in range(T + 3):
        fa.step()
    torch.cuda.synchronize()
    assert int(fa.counters[0]) == T + 3 and int(fa.counters[1]) == T + 3
    for k, big in guards.items():
        assert bool((big[:T] != 777.0).any()), k                 # rows 0 .. T - 1 were written
        assert bool((big[T:] == 777.0).all()), k                 # nothing behind them
    assert torch.isfinite(fa.action).all() and float(fa.action.abs().sum()) > 0
    import ctypes as C
    from phase_guided_terrain_traversal_amd import native
    a.store_rows = 0                                              # stores requested without a row count: refused
    assert fa._L.pgtt_policy_act(C.byref(a), None) == -1
    a.store_rows = T
    env.close()


@pytest.mark.parametrize("od", [215, 100, 16])
def test_policy_act_other_observation_widths(od):
    """pgtt_policy_act through the bare C ABI with observation widths other than the two tasks' (171 / 162 -> 11 k-blocks): 215 takes the second tuned
    instantiation (14 k-blocks), anything else the generic loop - head equal to the PyTorch-op MLP, ragged env count"""
    import ctypes as C
    from phase_guided_terrain_traversal_amd import acting, native
    L = acting._lib()
    n = 203
    torch.manual_seed(od)
    dims = (od, 512, 256, 128, 24)
    lins = [torch.nn.Linear(dims[i], dims[i + 1]).cuda() for i in range(4)]
    obs = torch.randn(n, od, device="cuda") * 2
    mean, std = torch.randn(od, device="cuda") * 0.3, torch.rand(od, device="cuda") + 0.5
    packed = [pack_linear(l.weight, l.bias) for l in lins]
    act, head = torch.zeros(n, 12, device="cuda"), torch.zeros(n, 24, device="cuda")
    a = acting.PgttPolicyActArgs()
    a.obs, a.priv, a.mean, a.std = obs.data_ptr(), None, mean.data_ptr(), std.data_ptr()
    for i, (w, b) in enumerate(packed):
        assert w.numel() == L.pgtt_policy_packed_floats(dims[i], dims[i + 1])
        a.w[i], a.b[i] = w.data_ptr(), b.data_ptr()
    a.eps, a.act, a.head = None, act.data_ptr(), head.data_ptr()
    a.store_obs = a.store_priv = a.store_u = a.store_logp = None
    a.counters, a.seed, a.env_id_offset = None, 1, 0
    a.num_envs, a.obs_dim, a.priv_dim, a.deterministic = n, od, 0, 1
    native.check(L.pgtt_policy_act(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    with torch.no_grad():
        x = (obs - mean) / std
        for l in lins[:-1]:
            x = torch.nn.functional.silu(l(x))
        want = lins[-1](x)
    assert float((head - want).abs().max()) < 2e-5 * (1 + float(want.abs().max()))
    assert float((act - torch.tanh(want[:, :12])).abs().max()) < 2e-5
    a.obs_dim = 300                                   # wider than the kernel's LDS rows: refused, not truncated
    assert L.pgtt_policy_act(C.byref(a), None) != 0
