"""Domain randomisation, sampled ONCE per env index (never at reset), mirroring the reference's
``randomization_fn`` protocol:

    go2/randomize.py:23-171         (task "stairs": dynamics + terrain variant + per-box friction)
    go2/randomize_simple.py:24-138  (task "flat_terrain": dynamics + floor friction)

The reference returns a batched ``mjx.Model`` + ``in_axes``; here the same 12 per-env model fields are
packed into the SoA ``params`` block (rows ``abi.P_*``), the per-env terrain ``variant`` and the per-env
per-box sliding ``box_friction`` that libpgtt.so consumes through ``PgttBuffers``.

Distributions are the reference's, including its quirks:
  * stairs: the floor-friction draw is dead code (randomize.py:30-36 restarts from model.geom_friction),
    so the plane keeps its nominal friction; boxes get U(0.4, 1) each;
  * body masses scale by U(0.9, 1.1) (all bodies) and the base gets an extra U(-1, 1) kg, inertias are NOT
    rescaled; frictionloss is nominally 0 so its scaling is a no-op;
  * gainprm[:,0] and biasprm[:,1] share one U(0.9, 1.1) factor per actuator.
Streams: numpy Philox keyed by (seed, global env id) so a shard draws the same numbers as the full batch
(JAX threefry equivalence is not required).
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np

from . import abi


def domain_randomize(model: Dict[str, Any], num_envs: int, seed: int = 0, terrain: Optional[np.ndarray] = None,
                     env_id_offset: int = 0, enable: bool = True, _frac: Optional[float] = None) -> Dict[str, np.ndarray]:
    """`_frac` (tests only): every draw returns lo + _frac * (hi - lo) instead of a Philox sample, which is how
    tests/golden/domain_randomize.npz was recorded from the reference's own functions."""
    n = int(num_envs)
    P = np.zeros((abi.NPARAM, n), dtype=np.float32)
    nbox = 0 if terrain is None else terrain.shape[1]
    T = 0 if terrain is None else terrain.shape[0]
    variant = np.zeros(n, dtype=np.int32)
    box_friction = np.full((abi.MAX_BOX, n), float(model["box_friction"][0]), dtype=np.float32) if nbox else None
    mass0 = np.asarray(model["body_mass"], dtype=np.float64)
    for e in range(n):
        g = np.random.Generator(np.random.Philox(key=[int(seed), int(env_id_offset + e)]))
        if _frac is None:
            u = lambda lo, hi, size=None: g.uniform(lo, hi, size)
        else:
            u = lambda lo, hi, size=None: (lo + _frac * (hi - lo)) * (np.ones(size) if size is not None else 1.0)
        floor_fr = u(0.4, 1.0)                                   # drawn in both variants
        if nbox:
            bf = u(0.4, 1.0, nbox)
            floor_fr = float(model["floor_friction"][0])         # dead draw on the stairs task
        _ = u(0.9, 1.1, 12)                                      # frictionloss scale (nominal 0 => no-op)
        armature = np.asarray(model["dof_armature"][6:]) * u(1.0, 1.05, 12)
        dpos = u(-0.05, 0.05, 3)
        # randomize.py draws for model.nbody bodies (world + robot + boxes); only the 13 robot bodies matter
        dmass = u(0.9, 1.1, 14 + nbox)[1:14]
        mass = mass0 * dmass
        mass[0] += u(-1.0, 1.0)
        qpos0 = np.asarray(model["qpos0"][7:]) + u(-0.05, 0.05, 12)
        damping = np.asarray(model["dof_damping"][6:]) * u(0.9, 1.1, 12)
        dgain = u(0.9, 1.1, 12)
        if not enable:
            armature, dpos, mass = np.asarray(model["dof_armature"][6:]), np.zeros(3), mass0.copy()
            qpos0, damping, dgain = np.asarray(model["qpos0"][7:]), np.asarray(model["dof_damping"][6:]), np.ones(12)
            floor_fr = float(model["floor_friction"][0])
        P[abi.P_BODY_MASS:abi.P_BODY_MASS + 13, e] = mass
        P[abi.P_BASE_IPOS:abi.P_BASE_IPOS + 3, e] = np.asarray(model["body_ipos"][0]) + dpos
        P[abi.P_QPOS0:abi.P_QPOS0 + 12, e] = qpos0
        P[abi.P_ARMATURE:abi.P_ARMATURE + 12, e] = armature
        P[abi.P_DAMPING:abi.P_DAMPING + 12, e] = damping
        P[abi.P_GAIN:abi.P_GAIN + 12, e] = np.asarray(model["act_gain"]) * dgain
        P[abi.P_BIAS1:abi.P_BIAS1 + 12, e] = np.asarray(model["act_bias"])[:, 1] * dgain
        P[abi.P_FLOOR_FRICTION, e] = floor_fr
        if nbox:
            variant[e] = int(g.integers(0, T)) if _frac is None else int(_frac * (T - 1))
            if enable:
                box_friction[:nbox, e] = bf
    out = {"params": P, "variant": variant}
    if nbox:
        out["box_friction"] = box_friction
    return out
