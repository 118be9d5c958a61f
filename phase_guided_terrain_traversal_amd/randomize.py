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
Streams: ONE numpy Philox stream per seed in which env e owns a fixed block of 280 outputs (vectorised: no per-env Python
loop), so a shard draws the same numbers as the full batch
(JAX threefry equivalence is not required).

Terrain variants are drawn per env like the reference's ``rand_idx`` (randomize.py:97-101) and then, by default, handed out in ascending
order within every block of ``VARIANT_GROUP`` consecutive GLOBAL env ids (`group_variants`): the multiset of draws of a block is untouched
(env ids are exchangeable labels and every other per-env draw is independent of the variant), but neighbouring envs now stand on the same
variant, and since ``physics_kernel`` hands each XCD a contiguous range of envs, each XCD's L2 pulls ~1/8 of the terrain table per launch
instead of all of it (profiles/hbm_traffic.json).  The order depends on global ids and the job's total only, so shards stay invariant -
which is why a call for a shard (``env_id_offset != 0``) must pass ``total_envs``.  A consequence for data-parallel jobs with fewer than
``VARIANT_GROUP`` envs per rank: a rank owns a contiguous SLICE of a sorted block, i.e. a contiguous range of variant ids (rank 0 the lowest),
so per-rank metrics are skewed by variant difficulty; only the all-reduced means are comparable between job shapes.  With >= 4096 envs per
rank (the reference's batch) every rank owns whole blocks and sees every variant.  ``group_variants=False`` gives the draw order.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np

from . import abi


# layout of one env's block of uniform draws (fixed width, whatever the task / terrain: a shard then finds its envs' blocks by
# env id alone).  280 doubles = 70 Philox4x64 counter steps per env.
_D_FLOOR, _D_BOXF, _D_FLOSS, _D_ARM, _D_IPOS, _D_MASS, _D_BASEM, _D_QPOS0, _D_DAMP, _D_GAIN, _D_VAR = 0, 1, 101, 113, 125, 128, 242, 243, 255, 267, 279
_DRAWS = 280
VARIANT_GROUP = 4096        # global env ids per block within which the variant draws are handed out in ascending order


def _uniform_blocks(seed: int, first_env: int, n: int) -> np.ndarray:
    """[n, 280] uniforms in [0, 1): ONE Philox4x64 stream keyed by the seed, env e owns outputs [280 e, 280 (e + 1)) - vectorised
    (no per-env generator objects: 32 768 envs take about a second instead of a Python loop of 32 768 generators) and independent of how the envs are sharded"""
    bg = np.random.Philox(key=[int(seed), 0x5047_5454])
    bg.advance(int(first_env) * (_DRAWS // 4))
    return np.random.Generator(bg).random((n, _DRAWS))


def _variant_draws(seed: int, first_env: int, n: int, T: int) -> np.ndarray:
    """randint(0, T) per global env id (randomize.py:97-100), in draw order"""
    U = _uniform_blocks(seed, first_env, n)[:, _D_VAR]
    return np.minimum((U * T).astype(np.int64), T - 1).astype(np.int32)


def grouped_variants(seed: int, first_env: int, n: int, T: int, total_envs: Optional[int] = None) -> np.ndarray:
    """the draws of `_variant_draws`, ascending within each block of VARIANT_GROUP global env ids (the last block ends at `total_envs`, the
    job's env count - default: this shard is the job's last): a permutation of every block's own draws, a function of global ids alone"""
    if total_envs is None and first_env != 0:
        # a shard that is not the job's last one and does not end on a block boundary would sort a PARTIAL block and silently differ from the
        # full batch: a call with an offset must say how many envs the job has
        raise ValueError("grouped terrain variants of a shard (env_id_offset != 0) need total_envs = the env count of the whole job")
    total = first_env + n if total_envs is None else int(total_envs)
    if first_env + n > total:
        raise ValueError("shard reaches beyond the job's env count (total_envs)")
    g0 = (first_env // VARIANT_GROUP) * VARIANT_GROUP
    g1 = min(-(-(first_env + n) // VARIANT_GROUP) * VARIANT_GROUP, total)
    v = _variant_draws(seed, g0, g1 - g0, T)
    for b in range(0, g1 - g0, VARIANT_GROUP):
        v[b:b + VARIANT_GROUP] = np.sort(v[b:b + VARIANT_GROUP], kind="stable")
    return v[first_env - g0:first_env - g0 + n]


def domain_randomize(model: Dict[str, Any], num_envs: int, seed: int = 0, terrain: Optional[np.ndarray] = None,
                     env_id_offset: int = 0, enable: bool = True, _frac: Optional[float] = None,
                     group_variants: bool = True, total_envs: Optional[int] = None) -> Dict[str, np.ndarray]:
    """`group_variants` / `total_envs`: see the module text (`total_envs` = env count of the whole job when this call draws one shard of it).
    `_frac` (tests only): every draw returns lo + _frac * (hi - lo) instead of a Philox sample, which is how
    tests/golden/domain_randomize.npz was recorded from the reference's own functions."""
    n = int(num_envs)
    P = np.zeros((abi.NPARAM, n), dtype=np.float32)
    nbox = 0 if terrain is None else terrain.shape[1]
    T = 0 if terrain is None else terrain.shape[0]
    U = _uniform_blocks(seed, env_id_offset, n) if _frac is None else np.full((n, _DRAWS), float(_frac))
    u = lambda lo, hi, col, width=1: lo + U[:, col:col + width] * (hi - lo)         # [n, width]
    nominal_floor = float(model["floor_friction"][0])
    # stairs: the floor-friction draw is dead code in the reference (randomize.py:30-36), the plane keeps its nominal value
    floor_fr = np.full(n, nominal_floor) if nbox else u(0.4, 1.0, _D_FLOOR)[:, 0]
    armature = np.asarray(model["dof_armature"][6:])[None] * u(1.0, 1.05, _D_ARM, 12)      # frictionloss (_D_FLOSS): nominal 0, scaling is a no-op
    dpos = u(-0.05, 0.05, _D_IPOS, 3)
    # randomize.py scales model.nbody masses (world + robot + boxes); only the 13 robot bodies matter: draws 1..13 of the block
    mass = np.asarray(model["body_mass"], dtype=np.float64)[None] * u(0.9, 1.1, _D_MASS + 1, 13)
    mass[:, 0] += u(-1.0, 1.0, _D_BASEM)[:, 0]
    qpos0 = np.asarray(model["qpos0"][7:])[None] + u(-0.05, 0.05, _D_QPOS0, 12)
    damping = np.asarray(model["dof_damping"][6:])[None] * u(0.9, 1.1, _D_DAMP, 12)
    dgain = u(0.9, 1.1, _D_GAIN, 12)
    if not enable:
        armature = np.repeat(np.asarray(model["dof_armature"][6:])[None], n, 0); dpos = np.zeros((n, 3))
        mass = np.repeat(np.asarray(model["body_mass"], dtype=np.float64)[None], n, 0)
        qpos0 = np.repeat(np.asarray(model["qpos0"][7:])[None], n, 0); damping = np.repeat(np.asarray(model["dof_damping"][6:])[None], n, 0)
        dgain = np.ones((n, 12)); floor_fr = np.full(n, nominal_floor)
    P[abi.P_BODY_MASS:abi.P_BODY_MASS + 13] = mass.T
    P[abi.P_BASE_IPOS:abi.P_BASE_IPOS + 3] = (np.asarray(model["body_ipos"][0])[None] + dpos).T
    P[abi.P_QPOS0:abi.P_QPOS0 + 12] = qpos0.T
    P[abi.P_ARMATURE:abi.P_ARMATURE + 12] = armature.T
    P[abi.P_DAMPING:abi.P_DAMPING + 12] = damping.T
    P[abi.P_GAIN:abi.P_GAIN + 12] = (np.asarray(model["act_gain"])[None] * dgain).T
    P[abi.P_BIAS1:abi.P_BIAS1 + 12] = (np.asarray(model["act_bias"])[:, 1][None] * dgain).T
    P[abi.P_FLOOR_FRICTION] = floor_fr
    out = {"params": P, "variant": np.zeros(n, dtype=np.int32)}
    if nbox:
        # randint(0, T) (randomize.py:97-100); the recorded fixture pins it to int(f (T - 1))
        if _frac is not None:
            out["variant"] = np.full(n, int(_frac * (T - 1)), dtype=np.int32)
        elif group_variants:
            out["variant"] = grouped_variants(seed, env_id_offset, n, T, total_envs)
        else:
            out["variant"] = np.minimum((U[:, _D_VAR] * T).astype(np.int64), T - 1).astype(np.int32)
        bf = np.full((abi.MAX_BOX, n), float(model["box_friction"][0]), dtype=np.float32)
        if enable:
            bf[:nbox] = u(0.4, 1.0, _D_BOXF, nbox).T
        out["box_friction"] = bf
    return out
