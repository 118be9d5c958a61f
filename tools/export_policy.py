"""Export the numeric content of a reference policy pickle (policy_folder/policyNNN) to .npz (THIS container only).

    python tools/export_policy.py [/root/reference] [policy177 policy3 ...]

The pickles hold `(RunningStatisticsState, PPONetworkParams[, ...])` of Brax classes backed by JAX arrays; neither
Brax nor JAX is installed, so a stub Unpickler maps every unknown class to a dummy and JAX arrays to numpy.  Only
numbers are stored: observation normaliser mean/std (171) and the policy MLP kernels / biases
(171-512-256-128-24).  Used by policy.py (N2: closed-loop rollout of a reference-trained policy in this simulator).
"""
import io
import os
import pickle
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
NAMES = sys.argv[2:] or ["policy177", "policy3"]
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "phase_guided_terrain_traversal_amd", "assets", "policies")


class Dummy:
    def __init__(self, *a, **k):
        self.args, self.kwargs = a, k

    def __setstate__(self, st):
        self.__dict__.update(st if isinstance(st, dict) else {"state": st})


def _reconstruct_array(fun, args, arr_state, aval_state=None):
    arr = fun(*args)                 # numpy's _reconstruct: an empty ndarray ...
    arr.__setstate__(arr_state)      # ... filled from the pickled numpy state
    return arr


class U(pickle.Unpickler):
    def find_class(self, module, name):
        # exactly the numpy globals an ndarray pickle needs - nothing else of the numpy namespace is reachable from the untrusted file
        if (module, name) in (("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"), ("numpy", "dtype"),
                              ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar")):
            return super().find_class(module, name)
        if name == "_reconstruct_array":
            return _reconstruct_array
        # everything that is not numpy maps to an inert stand-in: the pickles are untrusted input, no other importable
        # callable may be resolved (a default find_class would run arbitrary code)
        return type(name, (Dummy,), {})


def get(obj, key):
    if isinstance(obj, dict):
        return obj[key]
    return getattr(obj, key, None) if not hasattr(obj, "__dict__") or key in obj.__dict__ or hasattr(obj, key) else obj.__dict__[key]


def as_dict(o):
    if isinstance(o, dict):
        return o
    d = dict(getattr(o, "__dict__", {}))
    if not d and hasattr(o, "args") and o.args:
        return o.args
    return d


os.makedirs(OUT, exist_ok=True)
for name in NAMES:
    with open(os.path.join(REF, "policy_folder", name), "rb") as f:
        params = U(io.BytesIO(f.read())).load()
    norm, net = params[0], params[1]
    nd = as_dict(norm)
    mean = np.asarray(as_dict(nd["mean"])["state"] if not isinstance(nd["mean"], dict) else nd["mean"]["state"], dtype=np.float32)
    std = np.asarray(as_dict(nd["std"])["state"] if not isinstance(nd["std"], dict) else nd["std"]["state"], dtype=np.float32)
    pol = net["params"] if isinstance(net, dict) else as_dict(net)["policy"]["params"]
    out = {"mean": mean, "std": std}
    # statistics of the privileged observation (215) as the training run saw them: distribution-level evidence about the
    # reference's simulator (accelerometer, actuator forces, contact duty)
    try:
        pm = as_dict(nd["mean"]) if not isinstance(nd["mean"], dict) else nd["mean"]
        ps = as_dict(nd["std"]) if not isinstance(nd["std"], dict) else nd["std"]
        out["mean_priv"] = np.asarray(pm["privileged_state"], dtype=np.float32); out["std_priv"] = np.asarray(ps["privileged_state"], dtype=np.float32)
    except Exception as ex:
        print("  (no privileged statistics:", ex, ")")
    for i, layer in enumerate(pol):
        out[f"w{i}"] = np.asarray(pol[layer]["kernel"], dtype=np.float32)
        out[f"b{i}"] = np.asarray(pol[layer]["bias"], dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "obs", mean.shape, "layers", [out[f"w{i}"].shape for i in range(len(pol))], "count", as_dict(norm).get("count"))
