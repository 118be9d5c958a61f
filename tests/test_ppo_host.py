"""Host-side checks of the PPO loop's building blocks (no GPU): the Linear layers with the hand-written backward keep
nn.Linear's parameters / state_dict and fall back to the library path off the GPU; GAE-free pieces of the loss agree."""
import math

import pytest

torch = pytest.importorskip("torch")

from phase_guided_terrain_traversal_amd import ppo


def test_long_batch_linear_is_a_drop_in_for_nn_linear():
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.SiLU(), torch.nn.Linear(5, 3))
    net = ppo.mlp((7, 5), 3)
    assert [type(m).__name__ for m in net] == ["LongBatchLinear", "SiLU", "LongBatchLinear"]
    assert list(net.state_dict().keys()) == list(ref.state_dict().keys())          # policy import / export relies on the names
    net.load_state_dict(ref.state_dict())
    x = torch.randn(2048, 7, requires_grad=True)                                    # long batch, but on the CPU: library path
    y1, y2 = ref(x), net(x)
    assert torch.equal(y1, y2)
    g1 = torch.autograd.grad(y1.sum(), list(ref.parameters()))
    g2 = torch.autograd.grad(y2.sum(), list(net.parameters()))
    assert all(torch.equal(a, b) for a, b in zip(g1, g2))


def test_tanh_normal_log_prob_and_entropy_formulas():
    """log_prob / entropy of ppo.ActorCritic against a direct evaluation through torch.distributions (the fused HIP loss
    kernel is checked against these on the GPU)"""
    torch.manual_seed(1)
    loc, scale = torch.randn(64, 12), torch.rand(64, 12) + 0.1
    u = loc + scale * torch.randn(64, 12)
    base = torch.distributions.Normal(loc, scale)
    want = (base.log_prob(u) - 2.0 * (math.log(2.0) - u - torch.nn.functional.softplus(-2.0 * u))).sum(-1)
    assert torch.allclose(ppo.ActorCritic.log_prob(loc, scale, u), want, atol=1e-5)
    ent = (base.entropy() + 2.0 * (math.log(2.0) - u - torch.nn.functional.softplus(-2.0 * u))).sum(-1)
    assert torch.allclose(ppo.ActorCritic.entropy(loc, scale, u), ent, atol=1e-5)
    # log(1 - tanh(u)^2) = 2 (log 2 - u - softplus(-2u))
    assert torch.allclose(torch.log1p(-torch.tanh(u) ** 2), 2.0 * (math.log(2.0) - u - torch.nn.functional.softplus(-2.0 * u)), atol=1e-5)


def test_compute_gae_is_brax_compute_gae():
    """the vectorised form against a literal scalar transcription of brax.training.agents.ppo.losses.compute_gae [UPSTREAM-RECALL]:
    zero TD error and zero carry at truncated steps, zero bootstrap at terminated ones, advantages from the lambda-returns"""
    torch.manual_seed(3)
    T, N, lam, g = 23, 7, 0.95, 0.97
    rew, val, boot = torch.rand(T, N), torch.randn(T, N), torch.randn(N)
    done = (torch.rand(T, N) < 0.2).float()
    trunc = done * (torch.rand(T, N) < 0.5).float()
    term = done * (1 - trunc)
    adv, vs = ppo.compute_gae(trunc, term, rew, val, boot, lam, g)
    for e in range(N):
        acc, vs_ref = 0.0, [0.0] * T
        for t in reversed(range(T)):
            vn = boot[e] if t == T - 1 else val[t + 1, e]
            delta = (rew[t, e] + g * (1 - term[t, e]) * vn - val[t, e]) * (1 - trunc[t, e])
            acc = delta + g * (1 - term[t, e]) * (1 - trunc[t, e]) * lam * acc
            vs_ref[t] = acc + val[t, e]
        for t in range(T):
            vn = boot[e] if t == T - 1 else vs_ref[t + 1]
            a = (rew[t, e] + g * (1 - term[t, e]) * vn - val[t, e]) * (1 - trunc[t, e])
            assert abs(float(vs[t, e]) - float(vs_ref[t])) < 1e-5 and abs(float(adv[t, e]) - float(a)) < 1e-5
    assert float(adv[trunc.bool()].abs().max()) == 0.0          # a truncated step carries no advantage


def test_running_norm_starts_at_unit_std_and_exports():
    nm = ppo.RunningNorm(5, "cpu")
    x = torch.randn(64, 5) * 3 + 2
    assert torch.equal(nm(x), x)                                  # brax init_state: mean 0, std 1 before the first update
    nm.update(x)
    assert torch.allclose(nm.mean, x.mean(0), atol=1e-5) and torch.allclose(nm.std, x.std(0, unbiased=False), atol=1e-5)
    # checkpoint -> the npz layout policy.PolicyMLP loads: same action as the trained actor's mean
    import os, tempfile
    from phase_guided_terrain_traversal_amd import policy
    model = ppo.ActorCritic(obs_dim=5, priv_dim=6, act_dim=2, hidden=(8, 4))
    ck = ppo.checkpoint(model, nm, ppo.RunningNorm(6, "cpu"))
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "policy0.npz")
        ppo.export_policy_npz(ck, path)
        pm = policy.PolicyMLP(path)
        loc, _ = model.dist(nm(x))
        assert torch.allclose(pm(x), torch.tanh(loc), atol=1e-5)


def test_pack_linear_layout_on_cpu():
    """acting.pack_linear: [out / 16][in / 16][g][i][s] = W[16 tile + i][16 kb + 4 g + s], zero padding (the order pgtt_policy_act's fp32 MFMA tiles read;
    include/pgtt_train.h) - and a product against the packed form equals the plain one"""
    import torch
    from phase_guided_terrain_traversal_amd.acting import pack_linear
    torch.manual_seed(0)
    w, b = torch.randn(24, 171), torch.randn(24)
    p, pb = pack_linear(w, b)
    assert p.numel() == 32 * 176 and pb.numel() == 32 and float(pb[24:].abs().sum()) == 0 and torch.equal(pb[:24], b)
    P = p.view(2, 11, 4, 16, 4)
    for tile, kb, g, i, s_ in ((0, 0, 0, 0, 0), (1, 10, 2, 7, 2), (0, 5, 3, 15, 3), (1, 10, 3, 7, 3), (1, 3, 1, 9, 0)):
        n_, k_ = 16 * tile + i, 16 * kb + 4 * g + s_
        assert float(P[tile, kb, g, i, s_]) == (float(w[n_, k_]) if n_ < 24 and k_ < 171 else 0.0)
    # unpack = inverse permutation; x W^T through it
    W2 = P.permute(0, 3, 1, 2, 4).reshape(32, 176)
    x = torch.randn(5, 171)
    assert torch.allclose(torch.nn.functional.pad(x, (0, 5)) @ W2.T[:, :24], x @ w.T, atol=1e-5)
