"""ctypes mirror of include/pgtt.h (structs, row enums) shared by the product loader and the tests.

Only layout lives here; no arithmetic.  `sizeof` of every struct is checked against the values the
shared libraries report (pgtt_sizeof_*), see tests/test_abi.py.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict

import numpy as np

NQ, NV, NU, NBODY, NLEG = 19, 18, 12, 13, 4
MAX_BOX, NCON, NEFC = 100, 8, 44
SCAN_H, SCAN_W, NSCAN = 13, 9, 117
OBS, PRIV, NREW, NMETRIC = 171, 215, 21, 22
OBS_BASELINE, PRIV_BASELINE = 162, 206          # go2/joystick.py: no phase (8) / gait_freq (1) rows
METHODS = {"pgtt": 0, "baseline": 1}
LAYOUTS = {"auto": 0, "quad": 1, "oct": 2, "hex": 4}        # PGTT_LAYOUT_*: 4 / 8 / 16 lanes per env in physics_kernel
OBSERVE_FORMS = {"fused": 0, "split": 1}                    # PGTT_OBSERVE_*


def obs_dims(method="pgtt"):
    """(state, privileged_state) widths of the task definition `method` (name or PGTT_METHOD_* code)."""
    code = METHODS[method] if isinstance(method, str) else int(method)
    return (OBS_BASELINE, PRIV_BASELINE) if code == 1 else (OBS, PRIV)

REWARD_KEYS = ["tracking_lin_vel", "tracking_ang_vel", "lin_vel_z", "ang_vel_xy", "orientation",
               "dof_pos_limits", "pose", "termination", "stand_still", "torques", "action_rate",
               "energy", "feet_clearance", "feet_height", "feet_slip", "feet_air_time", "feet_phase",
               "feet_swing", "body_height", "contact", "center"]

# float SoA state rows
S_QPOS, S_QVEL, S_QWARM, S_CMD, S_PHASE, S_PHASE_DT, S_GAIT_FREQ = 0, 19, 37, 55, 58, 62, 63
S_LAST_ACT, S_LAST_LAST_ACT, S_AIR_TIME, S_SWING_PEAK, S_HMAX, S_HMIN = 64, 76, 88, 92, 96, 100
S_MOTOR_TARGETS, S_QERR_HIST, S_QVEL_HIST, S_LAST_CONTACT, NSTATE = 104, 116, 140, 164, 168
I_STEP, I_STEPS_UNTIL_CMD, I_RNG_CTR, I_EP_STEPS, NISTATE = 0, 1, 2, 3, 4
F_GYRO, F_ACCEL, F_GLOBAL_LINVEL, F_GLOBAL_ANGVEL, F_LOCAL_LINVEL, F_UPVECTOR, F_GRAVITY = 0, 3, 6, 9, 12, 15, 18
F_FEET_POS, F_FEET_VEL, F_ACT_FORCE, F_CONTACT, F_FOOT_SITE_Z, NFRAME = 21, 33, 45, 57, 61, 65
P_BODY_MASS, P_BASE_IPOS, P_QPOS0, P_ARMATURE, P_DAMPING, P_GAIN, P_BIAS1, P_FLOOR_FRICTION, NPARAM = \
    0, 13, 16, 28, 40, 52, 64, 76, 77

f, i32 = C.c_float, C.c_int32


class PgttModel(C.Structure):
    _fields_ = [
        ("body_pos", f * 3 * NBODY), ("body_quat", f * 4 * NBODY), ("body_ipos", f * 3 * NBODY),
        ("body_iquat", f * 4 * NBODY), ("body_mass", f * NBODY), ("body_inertia", f * 3 * NBODY),
        ("body_invweight0", f * 2 * NBODY), ("jnt_axis", f * 3 * 12), ("jnt_range", f * 2 * 12),
        ("jnt_solref", f * 2), ("jnt_solimp", f * 5), ("qpos0", f * NQ), ("dof_armature", f * NV),
        ("dof_damping", f * NV), ("dof_invweight0", f * NV), ("act_dof", i32 * NU), ("act_gain", f * NU),
        ("act_bias", f * 3 * NU), ("act_ctrlrange", f * 2 * NU), ("act_forcerange", f * 2 * NU),
        ("foot_geom_pos", f * 3 * NLEG), ("foot_radius", f * NLEG), ("foot_site_pos", f * 3 * NLEG),
        ("imu_pos", f * 3),
        ("foot_friction", f * 3), ("foot_solref", f * 2), ("foot_solimp", f * 5), ("foot_margin", f),
        ("foot_gap", f), ("foot_solmix", f),
        ("floor_friction", f * 3), ("floor_solref", f * 2), ("floor_solimp", f * 5), ("floor_margin", f),
        ("floor_gap", f), ("floor_solmix", f),
        ("box_friction", f * 3), ("box_solref", f * 2), ("box_solimp", f * 5), ("box_margin", f),
        ("box_gap", f), ("box_solmix", f),
        ("box_rbound", f), ("foot_condim", i32), ("floor_condim", i32), ("box_condim", i32),
        ("timestep", f), ("gravity", f * 3), ("impratio", f), ("tolerance", f), ("ls_tolerance", f),
        ("meaninertia", f), ("iterations", i32), ("ls_iterations", i32), ("max_geom_pairs", i32),
        ("max_contact_points", i32), ("key_qpos", f * NQ),
    ]


class PgttConfig(C.Structure):
    _fields_ = [
        ("ctrl_dt", f), ("sim_dt", f), ("n_substeps", i32), ("episode_length", i32), ("action_scale", f),
        ("history_len", i32), ("history_update_steps", i32), ("soft_joint_pos_limit_factor", f),
        ("noise_level", f), ("noise_joint_pos", f), ("noise_joint_vel", f), ("noise_gyro", f),
        ("noise_gravity", f), ("noise_linvel", f), ("noise_heightscan", f),
        ("reward_scale", f * NREW), ("tracking_sigma", f), ("swing_height", f), ("base_feet_distance", f),
        ("phase_sigma", f), ("cmd_u_max", f * 3), ("cmd_u_min", f * 3), ("cmd_b", f * 3),
        ("gait_freq", f * 2), ("scan_dist_x", f), ("scan_dist_y", f), ("scan_z_offset", f),
        ("autoreset", i32), ("method", i32), ("lane_layout", i32), ("observe_form", i32), ("test_hooks", i32),
    ]


class PgttBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "state", "istate", "frame", "scan_z", "obs_state", "obs_priv", "reward", "done", "metrics",
        "first_state", "first_obs", "ep_metrics", "params", "variant", "box_friction", "dbg_contact",
        "dbg_dist", "dbg_niter", "interval_sums")]


# (name, rows-or-cols, dtype, layout) ; layout "soa" => [rows][N], "aos" => [N][cols]
BUFFER_SPECS = [
    ("state", NSTATE, np.float32, "soa"), ("istate", NISTATE, np.int32, "soa"),
    ("frame", NFRAME, np.float32, "soa"), ("scan_z", NSCAN, np.float32, "aos"),
    ("obs_state", OBS, np.float32, "aos"), ("obs_priv", PRIV, np.float32, "aos"),
    ("reward", 1, np.float32, "vec"), ("done", 1, np.float32, "vec"),
    ("metrics", NMETRIC, np.float32, "soa"), ("first_state", S_CMD, np.float32, "soa"),
    ("first_obs", OBS + PRIV, np.float32, "aos"), ("ep_metrics", NMETRIC + 2, np.float32, "soa"),
]
OPTIONAL_SPECS = [
    ("params", NPARAM, np.float32, "soa"), ("variant", 1, np.int32, "vec"),
    ("box_friction", MAX_BOX, np.float32, "soa"), ("dbg_contact", NCON * 2, np.int32, "aos"),
    ("dbg_dist", NCON, np.float32, "aos"), ("dbg_niter", 1, np.int32, "vec"),
    ("interval_sums", NMETRIC + 2, np.float32, "soa"),
]


def buffer_shape(spec, n: int, method="pgtt"):
    name, k, dt, layout = spec
    od, pd = obs_dims(method)
    k = {"obs_state": od, "obs_priv": pd, "first_obs": od + pd}.get(name, k)
    return {"soa": (k, n), "aos": (n, k), "vec": (n,)}[layout]


def _fill(dst, src):
    a = np.asarray(src)
    flat = np.ctypeslib.as_array(dst).reshape(-1)
    flat[...] = a.reshape(-1).astype(flat.dtype)


def model_struct(m: Dict[str, Any]) -> PgttModel:
    """dict from mjcf.compile_mjcf / mjcf.load_model -> PgttModel (float32)."""
    s = PgttModel()
    for name, ctype in PgttModel._fields_:
        v = m[name]
        if hasattr(ctype, "_length_"):
            _fill(getattr(s, name), v)
        else:
            setattr(s, name, int(v) if ctype is i32 else float(v))
    return s


def config_struct(cfg: Dict[str, Any]) -> PgttConfig:
    """nested dict from configs.default_config() -> PgttConfig."""
    s = PgttConfig()
    s.ctrl_dt, s.sim_dt = cfg["ctrl_dt"], cfg["sim_dt"]
    s.n_substeps = int(round(cfg["ctrl_dt"] / cfg["sim_dt"]))
    s.episode_length = cfg["episode_length"]
    s.action_scale = cfg["action_scale"]
    s.history_len, s.history_update_steps = cfg["history_len"], cfg["history_update_steps"]
    s.soft_joint_pos_limit_factor = cfg["soft_joint_pos_limit_factor"]
    nz = cfg["noise_config"]
    s.noise_level = nz["level"]
    for k in ("joint_pos", "joint_vel", "gyro", "gravity", "linvel", "heightscan"):
        setattr(s, "noise_" + k, nz["scales"][k])
    rc = cfg["reward_config"]
    for i, k in enumerate(REWARD_KEYS):
        s.reward_scale[i] = rc["scales"][k]
    s.tracking_sigma, s.swing_height = rc["tracking_sigma"], rc["swing_height"]
    s.base_feet_distance, s.phase_sigma = rc["base_feet_distance"], rc["phase_sigma"]
    cc = cfg["command_config"]
    for i in range(3):
        s.cmd_u_max[i], s.cmd_u_min[i], s.cmd_b[i] = cc["u_max"][i], cc["u_min"][i], cc["b"][i]
    s.gait_freq[0], s.gait_freq[1] = cfg["gait_freq"]
    s.scan_dist_x, s.scan_dist_y, s.scan_z_offset = cfg["scan_dist_x"], cfg["scan_dist_y"], cfg["scan_z_offset"]
    s.autoreset = int(cfg.get("autoreset", 0))
    s.method = METHODS[cfg.get("method", "pgtt")]
    s.lane_layout = LAYOUTS[cfg.get("lane_layout", "auto")]
    s.observe_form = OBSERVE_FORMS[cfg.get("observe_form", "fused")]
    s.test_hooks = int(bool(cfg.get("test_hooks", False)))
    return s
