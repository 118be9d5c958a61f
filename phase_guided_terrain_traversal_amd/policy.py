"""Reference-trained policies in torch (N2 of SURVEY 8f): semantics of the reference's deploy/policy_net.py:6-80 —
`(obs - mean) / std` -> MLP 171-512-256-128-24 with SiLU -> `tanh(first 12 of 24)` (the mean of the Brax
tanh-normal head).  Weights are the numeric content of policy_folder/policyNNN exported by tools/export_policy.py
(assets/policies/*.npz); the action is in actuator order FR,FL,RR,RL like `Joystick.step` expects.
"""
from __future__ import annotations

import os

import numpy as np
import torch

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "policies")


class PolicyMLP(torch.nn.Module):
    def __init__(self, npz_path: str):
        super().__init__()
        d = np.load(npz_path)
        self.register_buffer("mean", torch.from_numpy(d["mean"]))
        self.register_buffer("std", torch.from_numpy(d["std"]))
        n = len([k for k in d.files if k.startswith("w")])
        self.layers = torch.nn.ModuleList()
        for i in range(n):
            w, b = d[f"w{i}"], d[f"b{i}"]
            lin = torch.nn.Linear(w.shape[0], w.shape[1])
            with torch.no_grad():
                lin.weight.copy_(torch.from_numpy(w).T)
                lin.bias.copy_(torch.from_numpy(b))
            self.layers.append(lin)

    @torch.no_grad()
    def head(self, obs: torch.Tensor) -> torch.Tensor:
        x = (obs - self.mean) / self.std
        for lin in self.layers[:-1]:
            x = torch.nn.functional.silu(lin(x))
        return self.layers[-1](x)

    @torch.no_grad()
    def forward(self, obs: torch.Tensor) -> torch.Tensor:
        loc, _ = torch.chunk(self.head(obs), 2, dim=-1)
        return torch.tanh(loc)

    @torch.no_grad()
    def sample(self, obs: torch.Tensor, generator=None) -> torch.Tensor:
        """an action drawn from the policy's tanh-normal head as Brax's training rollouts draw it (scale = softplus(raw) + 1e-3):
        the normaliser statistics of a reference policy were recorded under these stochastic actions, not under the mean"""
        loc, raw = torch.chunk(self.head(obs), 2, dim=-1)
        scale = torch.nn.functional.softplus(raw) + 1e-3
        return torch.tanh(loc + scale * torch.randn(loc.shape, device=loc.device, dtype=loc.dtype, generator=generator))


def load_policy(name: str = "policy177", device: str = "cuda:0") -> PolicyMLP:
    return PolicyMLP(os.path.join(_DIR, name + ".npz")).to(device)
