"""Loader for libpgtt.so (the HIP/gfx950 product library).  Fails loudly when the library is missing
or when no GPU is usable — there is no CPU fallback anywhere in this package."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PGTT_LIB", os.path.join(_HERE, "libpgtt.so"))   # PGTT_LIB: build experiments only
_LIB: Optional[C.CDLL] = None

EXPORTS = ["pgtt_create", "pgtt_destroy", "pgtt_set_terrain", "pgtt_bind", "pgtt_reset", "pgtt_step",
           "pgtt_physics", "pgtt_observe", "pgtt_scan", "pgtt_interval_reduce", "pgtt_set_test_overrides", "pgtt_enable_timing", "pgtt_last_kernel_ms", "pgtt_kernel_ms_mean",
           "pgtt_obs_dims", "pgtt_sizeof_model", "pgtt_sizeof_config", "pgtt_sizeof_buffers", "pgtt_version", "pgtt_build_info", "pgtt_last_error"]
TRAIN_EXPORTS = ["pgtt_ppo_policy_loss", "pgtt_ppo_linear_backward", "pgtt_policy_act", "pgtt_policy_packed_floats", "pgtt_rollout_record",
                 "pgtt_sizeof_policy_act_args", "pgtt_sizeof_rollout_record_args"]      # include/pgtt_train.h: trainer helpers, not the env boundary


class PgttError(RuntimeError):
    pass


def source_sha256() -> str:
    """SHA-256 over the sources physics_kernel is built from, as they are ON DISK (srchash.py).  csrc/Makefile embeds the same hash in the library
    at build time (`build_info()`): tools/collect_profiles.py stores it next to the rocprofv3 counters of that kernel, bench.py compares the record
    with what the LOADED library says about itself before quoting them."""
    from . import srchash
    return srchash.source_sha256()


def build_info(path: Optional[str] = None) -> dict:
    """What the library says it was built from: {"src": <source_sha256 at build time>, "flavor": "product" | "fastdiv" | "flip" | ...}
    (pgtt_build_info(), no GPU needed)."""
    if path or _LIB is None:
        import torch  # noqa: F401  (same order as lib(): torch's HIP runtime first, or a later torch.cuda in this process finds no GPU)
        L = C.CDLL(path or LIB_PATH)
    else:
        L = _LIB
    L.pgtt_build_info.restype = C.c_char_p
    return dict(kv.split("=", 1) for kv in L.pgtt_build_info().decode().split(";"))


def library_sha256(path: Optional[str] = None) -> str:
    import hashlib
    with open(path or LIB_PATH, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise PgttError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        # torch ships its own HIP runtime: load it FIRST so that libpgtt.so binds to the same libamdhip64
        # (loading /opt/rocm's copy first leaves torch with "No HIP GPUs are available")
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        L.pgtt_last_error.restype = C.c_char_p
        L.pgtt_version.restype = C.c_char_p
        L.pgtt_create.argtypes = [C.POINTER(abi.PgttConfig), C.POINTER(abi.PgttModel), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.pgtt_destroy.argtypes = [C.c_void_p]
        L.pgtt_set_terrain.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.pgtt_bind.argtypes = [C.c_void_p, C.POINTER(abi.PgttBuffers)]
        L.pgtt_reset.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p]
        for fn in ("pgtt_step", "pgtt_physics", "pgtt_observe"):
            getattr(L, fn).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.pgtt_scan.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
        L.pgtt_interval_reduce.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
        L.pgtt_enable_timing.argtypes = [C.c_void_p, C.c_int]
        L.pgtt_set_test_overrides.argtypes = [C.c_void_p, C.c_float, C.c_int]
        L.pgtt_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        assert L.pgtt_sizeof_model() == C.sizeof(abi.PgttModel)
        assert L.pgtt_sizeof_config() == C.sizeof(abi.PgttConfig)
        assert L.pgtt_sizeof_buffers() == C.sizeof(abi.PgttBuffers)
        _LIB = L
    return _LIB


def check(rc: int) -> None:
    if rc != 0:
        raise PgttError(f"libpgtt error {rc}: {lib().pgtt_last_error().decode()}")
