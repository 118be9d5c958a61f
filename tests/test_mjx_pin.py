"""Consumers of the MJX pin kit (tools/gen_golden_mjx.py): step-level vectors of the reference's OWN physics - one `mjx.step` and one
`Joystick.step` (go2/joystick_pgtt.py:141-231, mjx_env.step at :146-148) from recorded states, contact.geom / contact.dist as go2/base.py:153-171
reads them, the compiled MuJoCo model's constants - against the CPU oracle (float64) and, `-m gpu`, against the HIP kernels through the C ABI.

Two fixtures go through the SAME checks:
  * `mjx`     tests/golden/mjx_step.npz - produced on a machine that has mujoco / mujoco-mjx / playground.  It does not exist
              in this repository yet (none of those packages is installable in the build container): every test SKIPS for it.
  * `dry_run` the same recorder running on stand-in jax / mujoco / mjx / reference modules (tools/fake_mjx.py) with the oracle doing the arithmetic,
              generated into a temporary folder by the test session.  It pins nothing about MJX; it proves that the recorder's calls, the file layout, the
              raw-MuJoCo-field -> PgttBuffers conversion, the geom-id -> (foot, box) mapping (go2/base.py:87-105; the stand-ins number their geoms
              differently from the real model on purpose) and the comparison code work, so that the first real file comes from, and is checked by, code that has run.
Bars (north star): float32 state within 1e-4 after the step (qvel: 1e-4 / dt), ACTIVE (foot, geom) contact set identical, on the cases whose Newton
solve stops before the 5-iteration cap in the float64 oracle (a solve that is cut returns a point that depends on rounding, in MJX too);
the real fixture may miss them on 2 % of those cases (fp32 MJX against fp64), the dry run on none.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle
from phase_guided_terrain_traversal_amd import abi, configs, mjcf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REAL = os.path.join(ROOT, "tests", "golden", "mjx_step.npz")
ROBOT_BODIES = slice(1, 14)          # world 0, base 1 (TORSO_BODY_ID, go2/randomize.py:22), 12 links; the boxes follow (randomize.py:24)
LEG_OF_FOOT = [1, 0, 3, 2]           # go2_constants FEET order FR, FL, RR, RL -> body-tree legs FL, FR, RL, RR


# ---------------------------------------------------------------------------------------------------------------------- fixture access
class Pin:
    def __init__(self, path):
        self.g = np.load(path, allow_pickle=False)
        self.meta = json.loads(str(self.g["meta"]))
        self.dry = bool(self.meta["dry_run"])
        self.groups = self.meta["groups"]

    def __call__(self, group, key):
        return self.g[f"{group}/{key}"]

    def has(self, group, key):
        return f"{group}/{key}" in self.g.files

    def n(self, group):
        return self(group, "in_qpos").shape[0]

    def task(self, group):
        return str(self(group, "task"))

    # ---- raw per-env MuJoCo model fields (go2/randomize.py:150-163) -> PgttBuffers.params / variant / box_friction + the terrain table
    def model_inputs(self, group):
        n = self.n(group)
        floor = self(group, "ids_floor_geom_id").astype(int)
        plane, box_geoms = int(floor[0]), floor[1:]
        P = np.zeros((abi.NPARAM, n), np.float32)
        P[abi.P_BODY_MASS:abi.P_BODY_MASS + 13] = self(group, "dr_body_mass")[:, ROBOT_BODIES].T
        P[abi.P_BASE_IPOS:abi.P_BASE_IPOS + 3] = self(group, "dr_body_ipos")[:, 1].T
        P[abi.P_QPOS0:abi.P_QPOS0 + 12] = self(group, "dr_qpos0")[:, 7:].T
        P[abi.P_ARMATURE:abi.P_ARMATURE + 12] = self(group, "dr_dof_armature")[:, 6:].T
        P[abi.P_DAMPING:abi.P_DAMPING + 12] = self(group, "dr_dof_damping")[:, 6:].T
        P[abi.P_GAIN:abi.P_GAIN + 12] = self(group, "dr_actuator_gainprm")[:, :, 0].T
        P[abi.P_BIAS1:abi.P_BIAS1 + 12] = self(group, "dr_actuator_biasprm")[:, :, 1].T
        P[abi.P_FLOOR_FRICTION] = self(group, "dr_geom_friction")[:, plane, 0]
        assert np.abs(self(group, "dr_dof_frictionloss")).max() == 0.0          # nominal 0: its randomisation is a no-op
        out = dict(params=P, terrain=None, variant=None, box_friction=None)
        if len(box_geoms):
            box_bodies = self.g["model/geom_bodyid"].astype(int)[box_geoms]
            boxes = np.concatenate([self(group, "dr_body_pos")[:, box_bodies], self(group, "dr_body_quat")[:, box_bodies],
                                    self(group, "dr_geom_size")[:, box_geoms]], axis=2).astype(np.float32)           # [n, B, 10]
            uniq, variant = np.unique(boxes.reshape(n, -1), axis=0, return_inverse=True)
            out["terrain"] = uniq.reshape(len(uniq), boxes.shape[1], 10)
            out["variant"] = variant.reshape(n).astype(np.int32)
            bf = np.zeros((abi.MAX_BOX, n), np.float32)
            bf[:len(box_geoms)] = self(group, "dr_geom_friction")[:, box_geoms, 0].T
            out["box_friction"] = bf
            # every env's boxes must be one of the rows of the terrain matrix the generator was given (randomize.py:97-108 copies rows)
            tm = self(group, "terrain").reshape(self(group, "terrain").shape[0], -1)
            assert all((np.abs(tm - u[None]).max(1) == 0).any() for u in uniq.reshape(len(uniq), -1))
        return out

    def contacts(self, group, prefix="mjx_"):
        """per env: {(leg FL,FR,RL,RR ; box index or -1 = plane): smallest dist} over the foot-floor pairs of contact.geom (go2/base.py:153-171)"""
        feet = list(self(group, "ids_feet_geom_id").astype(int)); floor = list(self(group, "ids_floor_geom_id").astype(int))
        geom, dist = self(group, prefix + "contact_geom"), self(group, prefix + "contact_dist")
        out = []
        for e in range(geom.shape[0]):
            d = {}
            for (g1, g2), dd in zip(geom[e].astype(int), dist[e]):
                f, o = (g1, g2) if g1 in feet else (g2, g1)
                if f in feet and o in floor:
                    key = (LEG_OF_FOOT[feet.index(f)], floor.index(o) - 1)
                    d[key] = min(d.get(key, np.inf), float(dd))
            out.append(d)
        return out

    def state_rows(self, group):
        """recorded inputs -> PgttBuffers.state [168, n] / istate [4, n] (include/pgtt.h row enums)"""
        n = self.n(group)
        S, I = np.zeros((abi.NSTATE, n), np.float32), np.zeros((abi.NISTATE, n), np.int32)
        k = lambda name: np.asarray(self(group, "in_" + name), dtype=np.float64)
        S[abi.S_QPOS:abi.S_QPOS + 19] = k("qpos").T; S[abi.S_QVEL:abi.S_QVEL + 18] = k("qvel").T; S[abi.S_QWARM:abi.S_QWARM + 18] = k("qacc_warmstart").T
        for off, cnt, name in ((abi.S_CMD, 3, "command"), (abi.S_PHASE, 4, "phase"), (abi.S_LAST_ACT, 12, "last_act"), (abi.S_LAST_LAST_ACT, 12, "last_last_act"),
                               (abi.S_AIR_TIME, 4, "feet_air_time"), (abi.S_SWING_PEAK, 4, "swing_peak"), (abi.S_HMAX, 4, "H_max"), (abi.S_HMIN, 4, "H_min"),
                               (abi.S_MOTOR_TARGETS, 12, "motor_targets"), (abi.S_QERR_HIST, 24, "qpos_error_history"), (abi.S_QVEL_HIST, 24, "qvel_history"),
                               (abi.S_LAST_CONTACT, 4, "last_contact")):
            S[off:off + cnt] = k("info_" + name).reshape(n, cnt).T
        S[abi.S_PHASE_DT] = k("info_phase_dt"); S[abi.S_GAIT_FREQ] = k("info_gait_freq")
        I[abi.I_STEP] = k("info_step"); I[abi.I_STEPS_UNTIL_CMD] = k("info_steps_until_next_cmd")
        return S, I


def frame_from_sensordata(s, act_force):
    """sensordata[49] of go2_mjx_feetonly.xml:258-274 (+ actuator_force) in the order of the PGTT_F_* rows up to PGTT_F_ACT_FORCE + 12, without the gravity rows"""
    return np.concatenate([s[..., 0:3], s[..., 3:6], s[..., 13:16], s[..., 16:19], s[..., 19:22], s[..., 22:25], s[..., 25:37], s[..., 37:49], act_force], axis=-1)


FRAME_ROWS = np.r_[abi.F_GYRO:abi.F_GYRO + 3, abi.F_ACCEL:abi.F_ACCEL + 3, abi.F_GLOBAL_LINVEL:abi.F_GLOBAL_LINVEL + 3, abi.F_GLOBAL_ANGVEL:abi.F_GLOBAL_ANGVEL + 3,
                   abi.F_LOCAL_LINVEL:abi.F_LOCAL_LINVEL + 3, abi.F_UPVECTOR:abi.F_UPVECTOR + 3, abi.F_FEET_POS:abi.F_FEET_POS + 12, abi.F_FEET_VEL:abi.F_FEET_VEL + 12,
                   abi.F_ACT_FORCE:abi.F_ACT_FORCE + 12]


@pytest.fixture(scope="module", params=["dry_run", "mjx"])
def pin(request, tmp_path_factory):
    if request.param == "mjx":
        if not os.path.exists(REAL):
            pytest.skip("tests/golden/mjx_step.npz not present: produce it with tools/gen_golden_mjx.py on a machine that has mujoco-mjx (INTEGRATION.md 4)")
        return Pin(REAL)
    out = str(tmp_path_factory.mktemp("mjx_pin") / "mjx_step_dryrun.npz")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_golden_mjx.py"), "--dry-run", "--out", out, "--envs-flat", "48", "--envs-level4", "64",
                        "--roll", "20"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return Pin(out)


def cap(pin, count):
    """violations allowed among `count` converged cases: none for the dry run (same arithmetic), 2 % for fp32 MJX against the fp64 oracle"""
    return 0 if pin.dry else max(1, int(0.02 * count))


def active(d):
    return sorted(k for k, v in d.items() if v < 0)


def oracle_forward(pin, group, mi, model, e):
    boxes = bf = None
    if mi["terrain"] is not None:
        boxes = mi["terrain"][mi["variant"][e]]; bf = mi["box_friction"][:boxes.shape[0], e]
    return oracle.forward(model, np.asarray(pin(group, "in_qpos")[e], np.float64), np.asarray(pin(group, "in_qvel")[e], np.float64), np.asarray(pin(group, "in_ctrl")[e], np.float64),
                          warm=np.asarray(pin(group, "in_qacc_warmstart")[e], np.float64), boxes=boxes, box_friction=bf, params=mi["params"][:, e], fp64=True)


# ---------------------------------------------------------------------------------------------------------------------- CPU: oracle (float64) against the file
def test_fixture_layout(pin):
    assert pin.meta["format"] == 1 and pin.meta["feet_order"] == ["FR", "FL", "RR", "RL"]
    assert set(pin.groups) >= {"flat", "level4", "targeted0", "targeted1", "targeted2"}
    for grp in pin.groups:
        n = pin.n(grp)
        assert pin(grp, "in_qpos").shape == (n, 19) and pin(grp, "mjx_qpos").shape == (n, 19) and pin(grp, "step_obs_state").shape == (n, abi.OBS)
        assert pin(grp, "mjx_contact_geom").shape[0] == n and pin(grp, "mjx_contact_geom").shape[2] == 2 and pin(grp, "mjx_sensordata").shape == (n, 49)
        assert (pin(grp, "in_info_steps_until_next_cmd") == pin.meta["far_timer"]).all()
        assert len(pin(grp, "ids_feet_geom_id")) == 4 and len(pin(grp, "ids_floor_geom_id")) == (1 if pin.task(grp) == "flat_terrain" else 1 + abi.MAX_BOX)
    if not pin.dry:
        assert "mujoco" in " ".join(pin.meta["versions"])          # a real file says which MuJoCo / MJX / JAX produced it


def test_oracle_one_mjx_step_against_fixture(pin):
    """ONE mjx.step from the recorded (qpos, qvel, qacc_warmstart, ctrl, per-env model, terrain): next state, solver output, contact list, sensors"""
    tot = dict(cases=0, conv=0, bad_q=0, bad_v=0, bad_set=0, bad_dist=0, bad_sens=0)
    eq_all = []
    for grp in pin.groups:
        mi = pin.model_inputs(grp)
        model = abi.model_struct(mjcf.load_model(pin.task(grp)))
        fc = pin.contacts(grp)
        for e in range(pin.n(grp)):
            d = oracle_forward(pin, grp, mi, model, e)
            conv = d["niter"] < model.iterations
            eq = np.abs(d["qpos_next"] - pin(grp, "mjx_qpos")[e]).max(); ev = np.abs(d["qvel_next"] - pin(grp, "mjx_qvel")[e]).max()
            eq_all.append(eq)
            mine = {(int(f), int(b)): float(dd) for f, b, dd in zip(d["con_foot"], d["con_box"], d["con_dist"]) if b != -2}
            same_set = active(mine) == active(fc[e])
            dist_err = max([abs(mine[k] - fc[e][k]) for k in active(mine) if k in fc[e]] + [0.0])
            sens = frame_from_sensordata(np.asarray(pin(grp, "mjx_sensordata")[e], np.float64), np.asarray(pin(grp, "mjx_actuator_force")[e], np.float64))
            got = frame_from_sensordata(d["sensordata"], d["actuator_force"])
            es = (np.abs(got - sens) / (1 + np.abs(sens))).max()
            tot["cases"] += 1
            if conv:
                tot["conv"] += 1; tot["bad_q"] += eq > 1e-4; tot["bad_v"] += ev > 1e-4 / 0.005; tot["bad_set"] += not same_set
                tot["bad_dist"] += dist_err > 1e-5; tot["bad_sens"] += es > 1e-2
            if pin.dry:
                assert eq < 1e-9 and ev < 1e-7 and same_set and dist_err < 1e-12 and es < 1e-9, (grp, e, eq, ev, same_set, dist_err, es)
                assert np.abs(d["qacc"] - pin(grp, "mjx_qacc_warmstart")[e]).max() < 1e-7            # qacc_warmstart' = qacc
    print(f"\n[{pin.meta['backend']}] one mjx.step: {tot}, median |qpos' - fixture| = {np.median(eq_all):.2e}")
    assert tot["conv"] > 0.5 * tot["cases"]
    for k in ("bad_q", "bad_v", "bad_set", "bad_dist", "bad_sens"):
        assert tot[k] <= cap(pin, tot["conv"]), (k, tot)
    assert np.median(eq_all) < 1e-5


def host_buffers(pin, grp, mi, cfg_dict):
    n = pin.n(grp)
    hb = oracle.HostBuffers(n, with_params=True, with_variant=mi["terrain"] is not None, with_box_friction=mi["terrain"] is not None)
    hb["state"][...], hb["istate"][...] = pin.state_rows(grp)
    hb["params"][...] = mi["params"]
    if mi["terrain"] is not None:
        hb["variant"][...] = mi["variant"]; hb["box_friction"][...] = mi["box_friction"]
    return hb


def noise_free_config(**over):
    return configs.with_overrides(configs.training_config(), **{"noise_config.level": 0.0}, **over)


def joystick_step_errors(pin, grp, got):
    """relative / absolute errors per env of one Joystick.step against the file; `got` holds PgttBuffers-shaped arrays"""
    n = pin.n(grp)
    S = got["state"].astype(np.float64)
    keys = [str(k) for k in pin(grp, "step_metric_keys")]
    order = [keys.index(k) for k in abi.REWARD_KEYS + ["swing_peak"]]
    rel = lambda a, b: (np.abs(a - b) / (1 + np.abs(b))).reshape(n, -1).max(1)
    out = dict(qpos=np.abs(S[0:19].T - pin(grp, "step_qpos")).max(1), qvel=np.abs(S[19:37].T - pin(grp, "step_qvel")).max(1),
               obs=rel(got["obs_state"], pin(grp, "step_obs_state")), priv=rel(got["obs_priv"], pin(grp, "step_obs_priv")),
               reward=np.abs(got["reward"] - pin(grp, "step_reward")), done=np.abs(got["done"] - pin(grp, "step_done")),
               metrics=rel(got["metrics"].T, pin(grp, "step_metrics")[:, order]), scan=np.abs(got["scan_z"] - pin(grp, "step_scan_z")).max(1))
    info = np.zeros(n)
    for off, cnt, name in ((abi.S_CMD, 3, "command"), (abi.S_PHASE, 4, "phase"), (abi.S_LAST_ACT, 12, "last_act"), (abi.S_LAST_LAST_ACT, 12, "last_last_act"),
                           (abi.S_AIR_TIME, 4, "feet_air_time"), (abi.S_SWING_PEAK, 4, "swing_peak"), (abi.S_HMAX, 4, "H_max"), (abi.S_HMIN, 4, "H_min"),
                           (abi.S_MOTOR_TARGETS, 12, "motor_targets"), (abi.S_LAST_CONTACT, 4, "last_contact")):
        info = np.maximum(info, np.abs(S[off:off + cnt].T - np.asarray(pin(grp, "step_info_" + name), np.float64).reshape(n, cnt)).max(1))
    out["info"] = info
    out["hist"] = np.maximum(np.abs(S[abi.S_QERR_HIST:abi.S_QERR_HIST + 24].T - pin(grp, "step_info_qpos_error_history")).max(1),
                             np.abs(S[abi.S_QVEL_HIST:abi.S_QVEL_HIST + 24].T - pin(grp, "step_info_qvel_history")).max(1))
    I = got["istate"]
    # the fallen envs draw a new command timer (jax.random in the reference, Philox here): compared only where the episode goes on
    alive = pin(grp, "step_done") == 0
    out["ints"] = ((I[abi.I_STEP] != pin(grp, "step_info_step")) | (alive & (I[abi.I_STEPS_UNTIL_CMD] != pin(grp, "step_info_steps_until_next_cmd")))).astype(float)
    return out


STEP_TOL = dict(qpos=1e-4, qvel=1e-4 / 0.02, obs=6e-3, priv=1e-2, reward=2e-4, done=0.5, metrics=2e-3, scan=1e-5, info=2e-4, hist=2e-2, ints=0.5)


def test_oracle_joystick_step_against_fixture(pin):
    """ONE Joystick.step (4 x mjx.step + contact flags + scan + observations + 21 rewards + bookkeeping) with noise level 0"""
    for grp in pin.groups:
        mi = pin.model_inputs(grp)
        cfg = noise_free_config()
        cs, ms = abi.config_struct(cfg), abi.model_struct(mjcf.load_model(pin.task(grp)))
        hb = host_buffers(pin, grp, mi, cfg)
        resid = np.zeros(pin.n(grp))
        oracle.step(cs, ms, mi["terrain"], hb, np.asarray(pin(grp, "in_action"), np.float32), seed=0, nthreads=8, fp64=True, resid=resid)
        err = joystick_step_errors(pin, grp, hb.arrays)
        conv = resid < 1e-6
        print(f"\n[{pin.meta['backend']}] Joystick.step {grp}: {int(conv.sum())} / {len(conv)} converged;", {k: f"{np.median(v):.1e}/{v[conv].max() if conv.any() else 0:.1e}" for k, v in err.items()})
        for k, v in err.items():
            # float32 buffers on this side, float64 (dry run) or float32 (MJX) numbers in the file
            assert (v[conv] > STEP_TOL[k]).sum() <= cap(pin, conv.sum()), (grp, k, v[conv].max())


def test_model_constants_against_fixture(pin):
    """phase_guided_terrain_traversal_amd/mjcf.py (the compiler of assets/go2_*.json) against the compiled MuJoCo model the reference steps:
    options, inertial parameters, invweight0 / meaninertia, actuator gain and bias parameters (biasprm[2] included), geom contact parameters,
    the STALE compiled bounding radius of the box placeholders, site positions, keyframe"""
    g = lambda k: pin.g["model/" + k]
    m = mjcf.load_model("stairs")
    A = lambda k: np.asarray(m[k], np.float64)
    close = lambda a, b, tol: np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() <= tol * (1 + np.abs(np.asarray(b, np.float64)).max())
    assert close(g("opt_timestep"), 0.005, 1e-9) and close(g("opt_gravity"), A("gravity"), 1e-9) and close(g("opt_impratio"), A("impratio"), 1e-9)
    assert close(g("opt_tolerance"), A("tolerance"), 1e-9) and close(g("opt_ls_tolerance"), A("ls_tolerance"), 1e-9)
    assert int(g("opt_iterations")) == int(m["iterations"]) and int(g("opt_ls_iterations")) == int(m["ls_iterations"])
    assert int(g("numeric_max_contact_points")) == int(m["max_contact_points"]) and int(g("numeric_max_geom_pairs")) == int(m["max_geom_pairs"])
    assert close(g("stat_meaninertia"), A("meaninertia"), 1e-5)
    for k, tol in (("body_mass", 1e-7), ("body_ipos", 1e-7), ("body_pos", 1e-7), ("body_quat", 1e-7), ("body_invweight0", 1e-4)):
        assert close(g(k)[ROBOT_BODIES], A(k), tol), k
    # principal inertias and their frame are unique only up to the order of the axes and the sign of the quaternion: compare the inertia TENSOR in the body frame
    def tensor(quat, diag):
        w, x, y, z = quat
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        return R @ np.diag(diag) @ R.T
    for b in range(13):
        Ti = tensor(g("body_iquat")[1 + b] / np.linalg.norm(g("body_iquat")[1 + b]), g("body_inertia")[1 + b])
        Tm = tensor(A("body_iquat")[b] / np.linalg.norm(A("body_iquat")[b]), A("body_inertia")[b])
        assert np.abs(Ti - Tm).max() < 1e-6 * (1 + np.abs(Tm).max()), ("inertia tensor of body", b)
    for k, tol in (("dof_invweight0", 1e-4), ("dof_armature", 1e-7), ("dof_damping", 1e-7), ("qpos0", 1e-7), ("key_qpos", 1e-7)):
        assert close(g(k), A(k), tol), k
    assert close(g("jnt_range")[1:], A("jnt_range"), 1e-6) and close(g("jnt_axis")[1:], A("jnt_axis"), 1e-7)
    assert close(g("jnt_solref")[1], A("jnt_solref"), 1e-7) and close(g("jnt_solimp")[1], A("jnt_solimp"), 1e-7)
    assert close(g("actuator_gainprm")[:, 0], A("act_gain"), 1e-7)
    assert close(g("actuator_biasprm")[:, :3], A("act_bias"), 1e-7), "biasprm[0:3] of the <position> actuators (go2_mjx_feetonly.xml:27): see mjcf.with_bias_velocity"
    assert close(g("actuator_ctrlrange"), A("act_ctrlrange"), 1e-6) and close(g("actuator_forcerange"), A("act_forcerange"), 1e-7)
    # actuator a drives joint trnid[a, 0]; hinge j is dof 5 + j
    assert np.array_equal(g("actuator_trnid")[:, 0].astype(int) + 5, np.asarray(m["act_dof"], int))
    grp = "level4"
    feet, floor = pin(grp, "ids_feet_geom_id").astype(int), pin(grp, "ids_floor_geom_id").astype(int)
    for kind, ids in (("floor", floor[:1]), ("foot", feet), ("box", floor[1:])):
        for k, tol in (("friction", 1e-7), ("solref", 1e-7), ("solimp", 1e-7), ("margin", 1e-7), ("gap", 1e-7), ("solmix", 1e-7), ("condim", 0)):
            for gid in ids:
                assert close(g("geom_" + k)[gid], A(f"{kind}_{k}"), tol), (kind, k, gid)
    assert close(g("geom_rbound")[floor[1:]], A("box_rbound"), 1e-6), "compiled rbound of the 1 x 1 x 1 box placeholders (the max_geom_pairs cut uses it, SURVEY C4)"
    for i, gid in enumerate(feet):
        leg = LEG_OF_FOOT[i]
        assert close(g("geom_size")[gid, 0], A("foot_radius")[leg], 1e-7) and close(g("geom_pos")[gid], A("foot_geom_pos")[leg], 1e-7)
        assert int(g("geom_bodyid")[gid]) == 1 + (1 + 3 * leg + 2)                 # the calf of that leg (world is body 0)
        assert close(g("site_pos")[int(g("feet_site_id")[i])], A("foot_site_pos")[leg], 1e-7)
    assert close(g("site_pos")[int(g("imu_site_id"))], A("imu_pos"), 1e-7)


def test_targeted_cases_settle_the_recorded_model_questions(pin):
    """(i) _sphere_convex with the sphere centre inside the box; (ii) biasprm[2] through actuator_force at joint speed; (iii) the max_geom_pairs cut
    with the stale rbound.  A REAL file that fails here names the switch to flip."""
    model_d = mjcf.load_model("stairs")
    model = abi.model_struct(model_d)
    r = float(model_d["foot_radius"][0])
    # ---- (i)
    grp = "targeted0"
    mi, fc = pin.model_inputs(grp), pin.contacts(grp)
    inside = 0
    for e in range(pin.n(grp)):
        d = oracle_forward(pin, grp, mi, model, e)
        top = float(mi["terrain"][mi["variant"][e]][0, 2] + mi["terrain"][mi["variant"][e]][0, 9])
        for leg in range(4):
            c_depth = top - d["foot_xpos"][leg, 2]                      # > 0: the sphere CENTRE is below the box top
            k = (leg, 0)
            assert k in fc[e], (str(pin(grp, "case_what")[e]), "the file holds no (foot, slab) pair for this foot")
            keep, flip = -(c_depth + r), c_depth - r                     # inward normal + growing depth (product) / n = normalize(pt - centre) (literal recall)
            if c_depth > 1e-4:
                inside += 1
                assert abs(fc[e][k] - keep) < 1e-5 or abs(fc[e][k] - flip) < 1e-5, (fc[e][k], keep, flip)
                assert abs(fc[e][k] - keep) < 1e-5, (f"{pin(grp, 'case_what')[e]}: with the sphere centre {1e3 * c_depth:.1f} mm inside the box the reference reports dist = "
                                                     f"{fc[e][k]:.5f} = the FLIPPED frame ({flip:.5f}), not {keep:.5f}: build oracle and kernels with -DPGTT_SPHERE_CONVEX_FLIP")
            else:
                assert abs(fc[e][k] - keep) < 1e-5
    assert inside >= 8                                                  # the regime is reached (25 mm and 40 mm cases, four feet each)
    # ---- (ii)
    grp = "targeted1"
    mi = pin.model_inputs(grp)
    for e in range(pin.n(grp)):
        d = oracle_forward(pin, grp, mi, model, e)
        assert not active(pin.contacts(grp)[e])                         # in the air
        qv = np.asarray(pin(grp, "in_qvel")[e], np.float64)[6:]
        assert np.abs(qv).min() > 2.9
        fa = np.asarray(pin(grp, "mjx_actuator_force")[e], np.float64)
        assert np.abs(d["actuator_force"] - fa).max() < 1e-4 * (1 + np.abs(fa).max()), \
            "actuator_force at joint speed differs: biasprm[2] (go2_mjx_feetonly.xml:27) is not what mjcf.py compiles - see mjcf.with_bias_velocity"
        assert np.abs(d["qfrc_passive"] - pin(grp, "mjx_qfrc_passive")[e]).max() < 1e-4 * (1 + np.abs(pin(grp, "mjx_qfrc_passive")[e]).max())
    # ---- (iii)
    grp = "targeted2"
    mi, fc = pin.model_inputs(grp), pin.contacts(grp)
    cut = 0
    for e in range(pin.n(grp)):
        d = oracle_forward(pin, grp, mi, model, e)
        mine = {(int(f), int(b)): float(dd) for f, b, dd in zip(d["con_foot"], d["con_box"], d["con_dist"]) if b != -2}
        assert active(mine) == active(fc[e]), (str(pin(grp, "case_what")[e]), active(mine), active(fc[e]),
                                               "ACTIVE sets differ on the crowded terrain: the max_geom_pairs cut (stale rbound) is not what the reference does")
        boxes = mi["terrain"][mi["variant"][e]]
        for leg in range(4):
            p = d["foot_xpos"][leg]
            for b in range(90, 100):                                    # the ten long slabs
                bx = boxes[b]
                c, s = (1.0, 0.0) if bx[3] == 1 else (-1.0, 0.0)        # yaw 0 or 180 deg
                over = abs(p[0] - bx[0]) < bx[7] and abs(p[1] - bx[1]) < bx[8] and p[2] - r < bx[2] + bx[9]
                cut += bool(over and (leg, b) not in active(mine))
    assert cut >= 3, cut                                                # feet that overlap a slab WITHOUT a contact: the cut really removes pairs here


def test_targeted_check_detects_a_backend_that_flips_the_contact_frame(tmp_path):
    """the check of question (i) has teeth: a dry-run file recorded from the oracle's -DPGTT_SPHERE_CONVEX_FLIP build (the literal recalled
    _sphere_convex) fails it with the message that names the switch - what a real MJX file would do if MJX does flip"""
    flip = os.path.join(ROOT, "oracle", "liboracle_flip.so")
    if not os.path.exists(flip):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "flip"], check=True)
    out = str(tmp_path / "flip.npz")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_golden_mjx.py"), "--dry-run", "--out", out, "--envs-flat", "4", "--envs-level4", "4", "--roll", "1"],
                       env=dict(os.environ, PGTT_ORACLE_LIB=flip), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    with pytest.raises(AssertionError, match="PGTT_SPHERE_CONVEX_FLIP"):
        test_targeted_cases_settle_the_recorded_model_questions(Pin(out))


def test_recorder_uses_names_the_reference_source_really_has():
    """the MJX backend of tools/gen_golden_mjx.py was written against the reference's SOURCE and has never met the real packages: every attribute, key and
    signature it relies on is looked up in that source here (this container only - the GPU box has no /root/reference: skip)"""
    import ast
    import re
    ref = os.environ.get("PGTT_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "go2")):
        pytest.skip("reference checkout not present")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_golden_mjx as G
    base = open(os.path.join(ref, "go2", "base.py")).read()
    joy = open(os.path.join(ref, "go2", "joystick_pgtt.py")).read()
    rnd = open(os.path.join(ref, "go2", "randomize.py")).read()
    cfg = open(os.path.join(ref, "go2", "configs.py")).read()
    for attr in ("_feet_geom_id", "_floor_geom_id", "_imu_site_id", "_feet_site_id", "_mjx_model", "_mj_model"):
        assert re.search(r"self\.%s\s*=" % attr, base), attr                                   # what MjxBackend reads off the env object
    assert re.search(r"def mjx_model\(self\)", base) and re.search(r"def mj_model\(self\)", base)
    # Joystick(task=..., config=...), reset(rng), step(state, action)
    tree = ast.parse(joy)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Joystick")
    fn = {n.name: [a.arg for a in n.args.args] for n in cls.body if isinstance(n, ast.FunctionDef)}
    assert fn["__init__"][:3] == ["self", "task", "config"] and fn["reset"] == ["self", "rng"] and fn["step"] == ["self", "state", "action"]
    # every info key the recorder reads is a key of the dict literal Joystick.reset builds; the observation keys; the metric names
    reset_src = ast.get_source_segment(joy, next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "reset"))
    for k in G.INFO_KEYS + ["rng", "heightscan"]:
        assert re.search(r'"%s"\s*:' % k, reset_src), k
    assert '"state"' in joy and '"privileged_state"' in joy and 'metrics[f"reward/{k}"]' in joy and 'metrics["swing_peak"]' in joy
    assert "self._weights" in reset_src                                  # the attribute Joystick.step needs from a previous reset (why make_batch resets the base env once)
    # the randomization_fn: signature, returned pair, and exactly the per-env fields the recorder records
    assert re.search(r"def domain_randomize\(model[^,]*,\s*rng[^,]*,\s*terrain_matrix", rnd) and "return model, in_axes" in rnd
    marked = set(re.findall(r'"(\w+)"\s*:\s*0', rnd))
    assert marked == set(G.DR_FIELDS), (marked, G.DR_FIELDS)
    # the config fields the recorder overrides exist (training/train.py:127-129 + the noise level)
    for k in ("command_config", "u_max", "u_min", "gait_freq", "noise_config", "level", "reward_config", "scales", "action_scale"):
        assert re.search(r"\b%s\b" % k, cfg), k
    assert re.search(r"action_scale\s*=\s*0\.5", cfg)                       # motor_targets = default_pose + 0.5 * action in record_group


def test_generator_refuses_to_write_a_dry_run_under_the_real_name():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_golden_mjx.py"), "--dry-run"], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "reserved" in p.stderr and not os.path.exists(REAL)


# ---------------------------------------------------------------------------------------------------------------------- GPU: HIP kernels against the file
def _env_from_fixture(pin, grp, cfg, torch):
    from phase_guided_terrain_traversal_amd.env import Joystick
    mi = pin.model_inputs(grp)
    kw = dict(params=torch.from_numpy(mi["params"]))
    if mi["terrain"] is not None:
        kw.update(variant=torch.from_numpy(mi["variant"]), box_friction=torch.from_numpy(mi["box_friction"]))
    env = Joystick(pin.task(grp), cfg, num_envs=pin.n(grp), terrain=mi["terrain"], device="cuda:0", debug_contacts=True, **kw)
    env.reset(seed=0)                # allocates / initialises; the rows the step reads are then overwritten by the recorded state
    S, I = pin.state_rows(grp)
    env.buffers["state"].copy_(torch.from_numpy(S)); env.buffers["istate"].copy_(torch.from_numpy(I))
    return env, mi


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["hex", "oct", "quad"])
def test_hip_one_mjx_step_against_fixture(pin, layout):
    """physics_kernel with ctrl_dt = sim_dt (one control step = ONE mjx.step) from the recorded state, through the C ABI"""
    torch = pytest.importorskip("torch")
    tot = dict(cases=0, conv=0, bad_q=0, bad_v=0, bad_w=0, bad_set=0, bad_sens=0)
    for grp in pin.groups:
        cfg = dict(noise_free_config(), lane_layout=layout); cfg["ctrl_dt"] = 0.005
        env, mi = _env_from_fixture(pin, grp, cfg, torch)
        key = np.asarray(env.model["key_qpos"], np.float64)[7:]
        action = (np.asarray(pin(grp, "in_ctrl"), np.float64) - key[None]) / env.config["action_scale"]
        env.physics(torch.from_numpy(action.astype(np.float32)).cuda())
        torch.cuda.synchronize()
        b = {k: v.cpu().numpy() for k, v in env.buffers.items()}
        model = abi.model_struct(mjcf.load_model(pin.task(grp)))
        fc = pin.contacts(grp)
        con, dist = b["dbg_contact"].reshape(-1, 8, 2), b["dbg_dist"]
        for e in range(pin.n(grp)):
            d = oracle_forward(pin, grp, mi, model, e)                    # only for the set of converged cases (the checker, not the thing compared)
            conv = d["niter"] < model.iterations
            S = b["state"][:, e].astype(np.float64)
            eq, ev = np.abs(S[0:19] - pin(grp, "mjx_qpos")[e]).max(), np.abs(S[19:37] - pin(grp, "mjx_qvel")[e]).max()
            w = np.asarray(pin(grp, "mjx_qacc_warmstart")[e], np.float64)
            # relative to the size of the acceleration VECTOR: the crafted deep-penetration cases have |qacc| ~ 400, where a per-component relative error
            # of 1e-2 is what the oracle's own fp32 build shows against its fp64 build (8e-5 of the vector's magnitude)
            ew = np.abs(S[37:55] - w).max() / (1 + np.abs(w).max())
            mine = {(int(f), int(bb)): float(dd) for (f, bb), dd in zip(con[e], dist[e]) if bb != -2}
            sens = frame_from_sensordata(np.asarray(pin(grp, "mjx_sensordata")[e], np.float64), np.asarray(pin(grp, "mjx_actuator_force")[e], np.float64))
            es = (np.abs(b["frame"][FRAME_ROWS, e] - sens) / (1 + np.abs(sens))).max()
            tot["cases"] += 1
            if conv:
                tot["conv"] += 1; tot["bad_q"] += eq > 1e-4; tot["bad_v"] += ev > 1e-4 / 0.005; tot["bad_w"] += ew > 1e-3
                tot["bad_set"] += active(mine) != active(fc[e]); tot["bad_sens"] += es > 1e-2
        assert np.isfinite(b["state"]).all()
        env.close()
    print(f"\n[{pin.meta['backend']}, {layout}] HIP one mjx.step: {tot}")
    assert tot["conv"] > 0.5 * tot["cases"]
    lim = max(1, int(0.02 * tot["conv"]))          # fp32 kernels against a float64 (dry run) or fp32-MJX file: the caps of tests/test_gpu_parity.py, rounded up for ~250 cases
    for k in ("bad_q", "bad_v", "bad_w", "bad_set", "bad_sens"):
        assert tot[k] <= lim, (k, tot)


@pytest.mark.gpu
def test_hip_joystick_step_against_fixture(pin):
    """pgtt_step (physics_kernel + observe_kernel) from the recorded state and info, noise level 0: observations, reward, done, metrics, info, scan"""
    torch = pytest.importorskip("torch")
    for grp in pin.groups:
        cfg = noise_free_config()
        env, mi = _env_from_fixture(pin, grp, cfg, torch)
        env.step(torch.from_numpy(np.asarray(pin(grp, "in_action"), np.float32)).cuda())
        torch.cuda.synchronize()
        got = {k: v.cpu().numpy() for k, v in env.buffers.items()}
        err = joystick_step_errors(pin, grp, got)
        cs, ms = abi.config_struct(cfg), abi.model_struct(mjcf.load_model(pin.task(grp)))
        hb = host_buffers(pin, grp, mi, cfg)
        resid = np.zeros(pin.n(grp))
        oracle.step(cs, ms, mi["terrain"], hb, np.asarray(pin(grp, "in_action"), np.float32), seed=0, nthreads=8, fp64=True, resid=resid)
        conv = resid < 1e-6
        print(f"\n[{pin.meta['backend']}] HIP Joystick.step {grp}: {int(conv.sum())} / {len(conv)} converged;", {k: f"{np.median(v):.1e}/{v[conv].max() if conv.any() else 0:.1e}" for k, v in err.items()})
        lim = max(1, int(0.03 * conv.sum()))
        for k, v in err.items():
            assert (v[conv] > STEP_TOL[k]).sum() <= lim, (grp, k, np.sort(v[conv])[-3:])
        env.close()
