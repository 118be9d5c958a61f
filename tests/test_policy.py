"""policy.PolicyMLP (N2: import of reference-trained policies) against the reference's own forward pass: the fixture was produced by
running deploy/policy_net.py:6-80 (`policy_net()` = get_params + MLP) on policy_folder/policy177 and policy3 (tools/gen_golden_policy.py)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from phase_guided_terrain_traversal_amd import policy


@pytest.mark.parametrize("name", ["policy177", "policy3"])
def test_policy_mlp_matches_reference_forward(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "policy_forward.npz"))
    net = policy.load_policy(name, device="cpu")
    act = net(torch.from_numpy(g[f"{name}_obs"])).numpy()
    assert act.shape == (64, 12) and np.abs(act).max() <= 1.0
    assert np.abs(act - g[f"{name}_action"]).max() < 1e-6
    assert np.abs(g[f"{name}_action"]).mean() > 0.05            # not a saturated / degenerate sample
