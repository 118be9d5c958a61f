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
