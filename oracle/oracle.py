"""ctypes loader for the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY — importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Any, Dict, Optional

import numpy as np

from phase_guided_terrain_traversal_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[C.CDLL] = None


def build() -> None:
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def use_fast_build() -> str:
    """bench.py's cpu_baseline leg only: compile the oracle on THIS host with -O3 -march=native (`make fast`)
    and route the following calls through it.  Returns the flags of the build that will be timed; falls back to the
    portable -O2 -ffp-contract=off checker build when no compiler is available."""
    global _LIB
    try:
        subprocess.run(["make", "-C", _HERE, "-s", "fast"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _LIB = None
        _load(os.path.join(_HERE, "_fast", "liboracle_fast.so"))
        return "gcc -O3 -march=native, built on this host"
    except Exception:
        _LIB = None
        lib()
        return "gcc -O2 -ffp-contract=off portable build (no compiler on this host for -march=native)"


def lib() -> C.CDLL:
    if _LIB is None:
        # PGTT_ORACLE_LIB: another build of the same checker (oracle/Makefile: `make san` ASan + UBSan, `make flip`)
        path = os.environ.get("PGTT_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _load(path)
    return _LIB


def _load(path: str) -> C.CDLL:
    global _LIB
    if True:
        _LIB = C.CDLL(path)
        _LIB.pgtt_oracle_get_z.restype = C.c_double
        _LIB.pgtt_oracle_get_z.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int]
        _LIB.pgtt_oracle_quat_to_yaw.restype = C.c_double
        _LIB.pgtt_oracle_uniform.restype = C.c_float
        _LIB.pgtt_oracle_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    return _LIB


class Dump(C.Structure):
    d, i = C.c_double, C.c_int32
    _fields_ = [
        ("qpos_in_normalized", d * 19), ("xpos", d * 42), ("xquat", d * 56), ("xipos", d * 42), ("com", d * 3),
        ("cdof", d * 108), ("cinert", d * 140), ("qM", d * 324), ("qfrc_bias", d * 18), ("qfrc_passive", d * 18),
        ("qfrc_actuator", d * 18), ("actuator_force", d * 12), ("qfrc_smooth", d * 18), ("qacc_smooth", d * 18),
        ("con_dist", d * 8), ("con_pos", d * 24), ("con_frame", d * 72), ("con_friction", d * 8),
        ("con_solimp", d * 40), ("con_solref", d * 16), ("con_foot", i * 8), ("con_box", i * 8),
        ("efc_J", d * (44 * 18)), ("efc_D", d * 44), ("efc_aref", d * 44), ("efc_pos", d * 44),
        ("efc_force", d * 44), ("efc_active", i * 44), ("qacc", d * 18), ("qfrc_constraint", d * 18),
        ("niter", i), ("sensordata", d * 49), ("site_imu_mat", d * 9), ("site_foot", d * 12),
        ("foot_xpos", d * 12), ("qpos_next", d * 19), ("qvel_next", d * 18),
    ]

    def as_dict(self) -> Dict[str, np.ndarray]:
        out = {}
        shapes = dict(xpos=(14, 3), xquat=(14, 4), xipos=(14, 3), cdof=(18, 6), cinert=(14, 10), qM=(18, 18),
                      con_pos=(8, 3), con_frame=(8, 3, 3), con_solimp=(8, 5), con_solref=(8, 2), efc_J=(44, 18),
                      site_imu_mat=(3, 3), site_foot=(4, 3), foot_xpos=(4, 3))
        for name, ct in self._fields_:
            v = getattr(self, name)
            if hasattr(ct, "_length_"):
                a = np.ctypeslib.as_array(v).copy()
                out[name] = a.reshape(shapes[name]) if name in shapes else a
            else:
                out[name] = v
        return out


def _dp(a):
    return np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.POINTER(C.c_double))


def _fp(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(C.POINTER(C.c_float))


def forward(model: abi.PgttModel, qpos, qvel, ctrl, warm=None, boxes=None, box_friction=None, params=None,
            fp64: bool = True, lib: Optional[C.CDLL] = None) -> Dict[str, Any]:
    """One mjx.forward + Euler for a single env; returns every intermediate as float64 arrays.  `lib`: another build of the checker
    (tests/parity_explain.py: the -O3 -march=native build as a stand-in device)."""
    L = lib if lib is not None else globals()["lib"]()
    assert L.pgtt_oracle_sizeof_dump() == C.sizeof(Dump)
    d = Dump()
    warm = np.zeros(18) if warm is None else warm
    bx = None if boxes is None else np.ascontiguousarray(boxes, dtype=np.float32)
    nbox = 0 if bx is None else bx.shape[0]
    bf = None if box_friction is None else np.ascontiguousarray(box_friction, dtype=np.float32)
    pr = None if params is None else np.ascontiguousarray(params, dtype=np.float32)
    L.pgtt_oracle_forward(C.byref(model), _fp(pr), _fp(bx), _fp(bf), nbox, _dp(qpos), _dp(qvel), _dp(warm),
                          _dp(ctrl), int(fp64), C.byref(d))
    return d.as_dict()


def scan(cfg: abi.PgttConfig, boxes, center, yaw: float, fp64: bool = True) -> np.ndarray:
    out = np.zeros((abi.NSCAN, 3))
    bx = None if boxes is None else np.ascontiguousarray(boxes, dtype=np.float32)
    lib().pgtt_oracle_scan(C.byref(cfg), _fp(bx), 0 if bx is None else bx.shape[0], _dp(center), C.c_double(yaw),
                           int(fp64), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out.reshape(abi.SCAN_H, abi.SCAN_W, 3)


def get_z(phi, h, smin, fp64=True) -> float:
    return lib().pgtt_oracle_get_z(float(phi), float(h), float(smin), int(fp64))


def quat_to_yaw(q, fp64=True) -> float:
    return lib().pgtt_oracle_quat_to_yaw(_dp(q), int(fp64))


def uniform(seed, env, epoch, stream, idx) -> float:
    return lib().pgtt_oracle_uniform(seed, env, epoch, stream, idx)


class HostBuffers:
    """numpy twins of the PgttBuffers SoA layout, for driving the batch oracle."""

    def __init__(self, n: int, with_params=False, with_variant=False, with_box_friction=False, debug=True, method="pgtt"):
        self.n = n
        self.arrays: Dict[str, np.ndarray] = {}
        for spec in abi.BUFFER_SPECS:
            self.arrays[spec[0]] = np.zeros(abi.buffer_shape(spec, n, method), dtype=spec[2])
        opt = dict(params=with_params, variant=with_variant, box_friction=with_box_friction, dbg_contact=debug,
                   dbg_dist=debug, dbg_niter=debug, interval_sums=True)
        for spec in abi.OPTIONAL_SPECS:
            if opt[spec[0]]:
                self.arrays[spec[0]] = np.zeros(abi.buffer_shape(spec, n), dtype=spec[2])

    def __getitem__(self, k):
        return self.arrays[k]

    def struct(self) -> abi.PgttBuffers:
        s = abi.PgttBuffers()
        for name, _ in abi.PgttBuffers._fields_:
            a = self.arrays.get(name)
            setattr(s, name, None if a is None else a.ctypes.data)
        return s


def reset(cfg, model, terrain, bufs: HostBuffers, seed: int, env_id_offset: int = 0, mask=None, fp64=False, nthreads=1):
    T, B = (0, 0) if terrain is None else terrain.shape[:2]
    t = None if terrain is None else np.ascontiguousarray(terrain, dtype=np.float32)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).ctypes.data_as(C.POINTER(C.c_uint8))
    s = bufs.struct()
    lib().pgtt_oracle_reset(C.byref(cfg), C.byref(model), _fp(t), T, B, bufs.n, C.byref(s), C.c_uint64(seed),
                            C.c_int64(env_id_offset), m, int(fp64), int(nthreads))


def step(cfg, model, terrain, bufs: HostBuffers, action, seed: int, env_id_offset: int = 0, fp64=False, nthreads=1, resid=None, flags=None, fresh_rbound=False):
    """`resid`: optional float64 [N] array receiving, per env, the largest scaled gradient norm at the Newton solver's exit
    over the substeps (how far from the minimiser the 5-iteration cut left the solve); `flags`: optional int32 [N] receiving OData.diag_flags;
    `fresh_rbound`: diagnostic switch of the max_geom_pairs ranking (physics_impl.h broad_phase)"""
    lib().pgtt_oracle_set_diag(None if resid is None else resid.ctypes.data_as(C.c_void_p))
    lib().pgtt_oracle_set_diag_flags(None if flags is None else flags.ctypes.data_as(C.c_void_p))
    lib().pgtt_oracle_set_fresh_rbound(int(fresh_rbound))
    T, B = (0, 0) if terrain is None else terrain.shape[:2]
    t = None if terrain is None else np.ascontiguousarray(terrain, dtype=np.float32)
    a = np.ascontiguousarray(action, dtype=np.float32)
    assert a.shape == (bufs.n, 12)
    s = bufs.struct()
    lib().pgtt_oracle_step(C.byref(cfg), C.byref(model), _fp(t), T, B, bufs.n, C.byref(s), _fp(a), C.c_uint64(seed),
                           C.c_int64(env_id_offset), int(fp64), int(nthreads))
    lib().pgtt_oracle_set_diag(None); lib().pgtt_oracle_set_diag_flags(None); lib().pgtt_oracle_set_fresh_rbound(0)
