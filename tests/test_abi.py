"""The C-ABI library loads and exports every symbol include/pgtt.h and include/pgtt_train.h declare; struct layouts agree (no GPU needed)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from phase_guided_terrain_traversal_amd import abi, configs, mjcf, native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="pgtt.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(r"\b(pgtt_[a-z_0-9]+)\s*\(", text)) - {"pgtt_env"})


def test_header_symbols_exported():
    if not os.path.exists(native.LIB_PATH):
        pytest.skip("libpgtt.so not built (run __graft_entry__.build())")
    import torch  # noqa: F401  (load torch's HIP runtime first, see native.py)
    lib = C.CDLL(native.LIB_PATH)
    env_names, train_names = _declared(), _declared("pgtt_train.h")
    assert set(env_names) == set(native.EXPORTS)
    assert set(train_names) == set(native.TRAIN_EXPORTS) and not set(train_names) & set(env_names)      # trainer helpers stay out of the env ABI
    names = env_names + train_names
    for n in names:
        assert hasattr(lib, n), n
    assert lib.pgtt_sizeof_model() == C.sizeof(abi.PgttModel)
    assert lib.pgtt_sizeof_config() == C.sizeof(abi.PgttConfig)
    assert lib.pgtt_sizeof_buffers() == C.sizeof(abi.PgttBuffers)
    lib.pgtt_version.restype = C.c_char_p
    assert b"gfx950" in lib.pgtt_version()


def test_row_enums_match_header():
    text = open(os.path.join(ROOT, "include", "pgtt.h")).read()
    for name, val in re.findall(r"PGTT_([SIFP]_[A-Z_0-9]+)\s*=\s*(\d+)", text):
        assert getattr(abi, name) == int(val), name
    for name in ("NSTATE", "NISTATE", "NFRAME", "NPARAM"):
        assert getattr(abi, name) == int(re.search(rf"PGTT_{name}\s*=\s*(\d+)", text).group(1))
    for name in ("NQ", "NV", "NU", "NBODY", "MAX_BOX", "NCON", "NEFC", "NSCAN", "OBS", "PRIV", "NREW", "NMETRIC"):
        assert getattr(abi, name) == int(re.search(rf"#define PGTT_{name}\s+(\d+)", text).group(1))
    keys = re.search(r"enum \{\s*PGTT_R_TRACKING_LIN_VEL = 0,(.*?)\};", text, re.S).group(1)
    order = ["tracking_lin_vel"] + [k.strip()[len("PGTT_R_"):].lower() for k in keys.replace("\n", " ").split(",") if k.strip()]
    assert order == abi.REWARD_KEYS


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product must fail loudly, never route to the oracle."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    if not os.path.exists(native.LIB_PATH):
        pytest.skip("libpgtt.so not built")
    L = native.lib()
    cs = abi.config_struct(configs.default_config()); ms = abi.model_struct(mjcf.load_model("flat_terrain"))
    h = C.c_void_p()
    rc = L.pgtt_create(C.byref(cs), C.byref(ms), 0, 64, C.byref(h))
    assert rc in (-4, -3), rc                              # PGTT_E_NODEVICE / PGTT_E_HIP
    assert b"no HIP device" in L.pgtt_last_error() or b"hip" in L.pgtt_last_error().lower()
    from phase_guided_terrain_traversal_amd.env import Joystick
    with pytest.raises(Exception):
        Joystick("flat_terrain", num_envs=8, device="cpu")


def test_create_refuses_what_the_kernels_cannot_compute():
    """argument checks run before the device is touched: values the kernels hold as compile-time shapes (two history samples, <= 4 box contacts per foot,
    the 4-row pyramid of condim 3) or need consistent (n_substeps, timestep) are a PGTT_E_ARG, never a silently different simulation"""
    if not os.path.exists(native.LIB_PATH):
        pytest.skip("libpgtt.so not built")
    L = native.lib()
    base_m = mjcf.load_model("stairs")
    h = C.c_void_p()

    def rc(cfg_over=None, model_over=None, n=64):
        c = abi.config_struct(dict(configs.training_config(), **(cfg_over or {})))
        if cfg_over and "n_substeps" in cfg_over:
            c.n_substeps = cfg_over["n_substeps"]
        m = abi.model_struct(dict(base_m, **(model_over or {})))
        return L.pgtt_create(C.byref(c), C.byref(m), 0, n, C.byref(h)), L.pgtt_last_error()
    for over, word in (({"history_len": 3}, b"history_len"), ({"history_update_steps": 0}, b"history_update_steps"), ({"episode_length": 0}, b"episode_length"),
                       ({"n_substeps": 3}, b"n_substeps"), ({"sim_dt": 0.004}, b"timestep")):
        r, msg = rc(cfg_over=over)
        assert r == -1 and word in msg, (over, r, msg)
    for over, word in (({"max_contact_points": 8}, b"max_contact_points"), ({"max_contact_points": -1}, b"max_contact_points"), ({"timestep": 0.002}, b"timestep"),
                       ({"box_margin": 0.002}, b"margin"), ({"foot_margin": 0.001, "foot_gap": 0.0005}, b"margin"), ({"floor_condim": 1, "foot_condim": 1}, b"condim"), ({"box_condim": 4}, b"condim"), ({"iterations": 0}, b"iteration")):
        r, msg = rc(model_over=over)
        assert r == -1 and word in msg, (over, r, msg)
    assert rc(n=0)[0] == -1 and rc(n=(1 << 22) + 1)[0] == -1
    r, msg = rc()                                           # the shipped values pass the checks (and then meet the device, or its absence)
    assert r in (0, -3, -4), (r, msg)
    if r == 0:
        L.pgtt_destroy(h)


def test_execution_options_come_through_the_abi_not_the_environment():
    """lane layout / observe form / test hooks are PgttConfig fields; the library reads no environment variable"""
    cfg = dict(configs.training_config(), lane_layout="oct", observe_form="split", test_hooks=True)
    c = abi.config_struct(cfg)
    assert (c.lane_layout, c.observe_form, c.test_hooks) == (2, 1, 1)
    c0 = abi.config_struct(configs.training_config())
    assert (c0.lane_layout, c0.observe_form, c0.test_hooks) == (0, 0, 0)
    text = open(os.path.join(ROOT, "include", "pgtt.h")).read()
    for name, val in (("AUTO", 0), ("QUAD", 1), ("OCT", 2), ("HEX", 4)):
        assert int(re.search(rf"PGTT_LAYOUT_{name}\s*=\s*(\d+)", text).group(1)) == val == abi.LAYOUTS[name.lower()]
    for f in os.listdir(os.path.join(ROOT, "phase_guided_terrain_traversal_amd", "csrc")):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(ROOT, "phase_guided_terrain_traversal_amd", "csrc", f)).read(), f


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "phase_guided_terrain_traversal_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "from oracle" not in src and "import oracle" not in src and "liboracle" not in src, f


def test_model_struct_roundtrip():
    m = mjcf.load_model("stairs")
    s = abi.model_struct(m)
    assert np.allclose(np.ctypeslib.as_array(s.body_mass), m["body_mass"])
    assert list(s.act_dof) == [9, 10, 11, 6, 7, 8, 15, 16, 17, 12, 13, 14]
    assert s.iterations == 5 and s.ls_iterations == 5 and s.max_geom_pairs == 25 and s.max_contact_points == 4
    assert abs(s.timestep - 0.005) < 1e-9 and abs(s.impratio - 100) < 1e-6
    assert np.allclose(np.ctypeslib.as_array(s.act_bias)[0], [0, -40, -0.5])
    c = abi.config_struct(configs.training_config())
    assert c.n_substeps == 4 and abs(c.cmd_u_max[2] - 1.0) < 1e-7 and abs(c.gait_freq[1] - 3) < 1e-7
    assert abs(c.reward_scale[abi.REWARD_KEYS.index("feet_phase")] - 0.5) < 1e-7
    assert abs(c.reward_scale[abi.REWARD_KEYS.index("contact")] - 2.0) < 1e-7


def test_evaluation_config_has_the_reference_evaluators_command_range():
    """training/evaluate.py:127-129 evaluates on u_max = [0.4, 0.4, 0.7], gait_freq = [1, 3] - not on training's +-[0.6, 0.6, 1.0]"""
    for method in ("pgtt", "baseline"):
        c = configs.evaluation_config(method)
        assert c["command_config"]["u_max"] == [0.4, 0.4, 0.7] and c["command_config"]["u_min"] == [-0.4, -0.4, -0.7] and c["gait_freq"] == [1, 3]
        assert c["method"] == method and c["reward_config"] == configs.training_config(method)["reward_config"]
    assert configs.training_config()["command_config"]["u_max"] == [0.6, 0.6, 1.0]


def test_tools_and_entry_points_compile():
    """the GPU-side helper scripts cannot run here; at least they must be syntactically valid"""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, "tools", "*.py")) + [os.path.join(root, f) for f in ("bench.py", "train.py", "__graft_entry__.py")]
    assert len(files) > 10
    for f in files:
        compile(open(f).read(), f, "exec")
