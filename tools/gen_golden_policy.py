"""Golden vectors of the reference's policy forward (THIS container only; the reference is imported, never copied):

    python tools/gen_golden_policy.py   ->  tests/golden/policy_forward.npz

`deploy/policy_net.py` (torch, imports without MJX) is loaded from /root/reference; its own `policy_net()` = `get_params` + `MLP`
(policy_net.py:6-80) is run on policy_folder/policy177 and policy3 with `pickle.load` routed through the stub Unpickler of
tools/export_policy.py (Brax / JAX are not installed: their classes unpickle to inert stand-ins, JAX arrays to numpy).  The
fixture holds 64 seeded observations per policy and the reference's actions; tests/test_policy.py checks policy.PolicyMLP on them."""
import importlib.util
import io
import os
import pickle
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


class Stand:          # attribute bag: Brax dataclasses pickle as (class, state dict)
    def __init__(self, *a, **k):
        self.args, self.kwargs = a, k

    def __setstate__(self, st):
        self.__dict__.update(st if isinstance(st, dict) else {"state": st})


def _reconstruct_array(fun, args, arr_state, aval_state=None):
    arr = fun(*args)
    arr.__setstate__(arr_state)
    return arr


class U(pickle.Unpickler):
    def find_class(self, module, name):
        # exactly the numpy globals an ndarray pickle needs - nothing else of the numpy namespace is reachable from the untrusted file
        if (module, name) in (("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"), ("numpy", "dtype"),
                              ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar")):
            return super().find_class(module, name)
        if name == "_reconstruct_array":
            return _reconstruct_array
        return type(name, (Stand,), {})


spec = importlib.util.spec_from_file_location("ref_policy_net", os.path.join(REF, "deploy", "policy_net.py"))
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)
ref.pickle = types.SimpleNamespace(load=lambda f: U(io.BytesIO(f.read())).load())     # the ONLY substitution: how the file is unpickled

out = {}
for name in ("policy177", "policy3"):
    net = ref.policy_net(os.path.join(REF, "policy_folder", name))                   # the reference's get_params + MLP, SiLU default
    g = torch.Generator().manual_seed(177)
    mean, std = net.mean, net.std
    obs = mean + std * torch.randn(64, mean.shape[0], generator=g) * 1.5             # observations spread like the policy's own statistics
    with torch.no_grad():
        act = net(obs)
    out[f"{name}_obs"] = obs.numpy(); out[f"{name}_action"] = act.numpy()
    print(name, "obs", tuple(obs.shape), "action", tuple(act.shape), "layers", [tuple(l.weight.shape) for l in net.layers])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "policy_forward.npz"), **out)
