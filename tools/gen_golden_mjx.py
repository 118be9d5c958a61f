"""MJX PIN KIT - step-level vectors of the reference's un-vendored physics, for the machine that has it.

    # on a machine with  mujoco  mujoco-mjx  mujoco_playground  jax  ml_collections  etils  scipy  (what the reference imports):
    python tools/gen_golden_mjx.py --ref /path/to/phase_guided_terrain_traversal            ->  tests/golden/mjx_step.npz
    # anywhere (this container included): the same recorder with the repo's CPU oracle standing in for mjx
    python tools/gen_golden_mjx.py --dry-run --out /tmp/mjx_step_dryrun.npz

Why: MuJoCo / MJX / Playground are not vendored by the reference, not pinned by it and not installable in the build container
(SURVEY.md 8c), so the physics restatement (oracle/physics_impl.h) and the HIP kernels are held to statistics of the reference's
trained policies and to an independent derivation - but to no OUTPUT of the reference's own step.  This script closes that on the
first machine that has MJX: it imports the reference UNMODIFIED (`sys.path.insert(0, REF)`; nothing is copied, only numbers are
stored), builds `go2.joystick_pgtt.Joystick` (go2/joystick_pgtt.py:38-48) for `flat_terrain` and `stairs`, rolls a batch of envs
out for a few steps and records, for every env of the batch, from ONE identical state:

  * one `mjx.step`  (the call inside mjx_env.step, go2/joystick_pgtt.py:146-148):
      in : qpos, qvel, qacc_warmstart, ctrl, the per-env model fields of go2/randomize.py:23-171 (raw MuJoCo arrays), terrain variant
      out: qpos', qvel', qacc_warmstart', qacc, contact.geom[8, 2], contact.dist[8], contact.frame / pos, sensordata[49], efc_force,
           actuator_force, qfrc_bias / passive / actuator / constraint (whatever the installed MJX exposes)
  * one `Joystick.step` with noise level 0 (go2/joystick_pgtt.py:141-231; the command timer is set far from expiry so that no
    jax.random draw enters a compared quantity):
      in : the same state + every `info` field + action
      out: qpos', qvel', obs['state'][171], obs['privileged_state'][215], reward, done, 22 metrics, every `info` field, scan heights
  * the compiled MuJoCo model's constants (options, inertias, invweight0, actuator gain / bias parameters, geom parameters, rbound,
    meaninertia, the geom-id tables of go2/base.py:87-105): pins phase_guided_terrain_traversal_amd/mjcf.py against the real compiler,

plus a TARGETED group on a hand-made terrain matrix that settles the three recorded model questions:
  (i)   a foot driven 5 .. 40 mm into a box top - beyond the 17.5 mm foot radius - : does `_sphere_convex` flip the contact frame once the
        sphere centre is inside the box?  (the product does NOT flip: -DPGTT_SPHERE_CONVEX_FLIP is the other answer)
  (ii)  joints at speed, robot in the air: actuator_force pins biasprm[2] of the <position> actuators (go2_mjx_feetonly.xml:27)
  (iii) a foot on a long slab whose centre is NOT among the 25 nearest (foot, box) centre pairs: the max_geom_pairs cut with the
        stale compiled rbound (go2_mjx_feetonly.xml:14-15).

tests/test_mjx_pin.py consumes the file: oracle-f64 against it on CPU, the HIP kernels against it on the GPU (`-m gpu`), both
self-skipping while tests/golden/mjx_step.npz does not exist.  The DRY RUN proves the plumbing here: tools/fake_mjx.py supplies
stand-ins for jax / mujoco / mjx / playground and the reference's three modules, with oracle.forward / oracle.step doing the
arithmetic, and the SAME MjxBackend code below runs on them unchanged - every call, attribute name and shape of the recorder is
executed (stand-in "MuJoCo" ids differ from the real model's on purpose, so the id -> (foot, box) mapping of go2/base.py:87-105
is exercised) - and the tests run green on the file in this container.

This file is a tool: nothing in the product, bench.py or __graft_entry__.py imports it.  The MJX backend below cannot be executed in
the build container; it is written against the public API of mujoco >= 3.2 / mujoco_playground >= 0.0.4 and reads every optional field
defensively (a field the installed version does not expose is listed in meta["missing"], not fatal).
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
from typing import Any, Dict, List, Optional

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FORMAT_VERSION = 1
FEET = ["FR", "FL", "RR", "RL"]                          # go2/go2_constants.py:55-74: sensor / geom / site order of the task layer
INFO_KEYS = ["command", "step", "steps_until_next_cmd", "phase", "phase_dt", "gait_freq", "last_act", "last_last_act", "feet_air_time",
             "last_contact", "swing_peak", "H_max", "H_min", "motor_targets", "qpos_error_history", "qvel_history"]
FAR_TIMER = 1000                                         # steps_until_next_cmd of the recorded Joystick.step: never expires in it
# per-env model fields go2/randomize.py:150-163 marks in_axes = 0 (recorded RAW, in MuJoCo's own layout and names)
DR_FIELDS = ["geom_friction", "body_ipos", "body_mass", "qpos0", "dof_frictionloss", "dof_armature", "dof_damping", "actuator_gainprm",
             "actuator_biasprm", "body_pos", "body_quat", "geom_size"]


# ====================================================================================================================== terrains
def parked(n: int) -> List[List[float]]:
    """placeholder boxes far away, like the unused ones of terrain_scene_mjx.xml / terrain/generator.py:368-391"""
    return [[100.0 + k, 100.0 + k, 100.0 + k, 1, 0, 0, 0, 0.5, 0.5, 0.5] for k in range(n)]


def targeted_terrain() -> np.ndarray:
    """[3, 100, 10] hand-made variants for the TARGETED group (pos xyz, quat wxyz, half-size xyz - terrains/level*.npy layout)"""
    T = []
    # variant 0: ONE big slab, top at z = 0.10: feet can be pushed any depth into a flat box top, nothing else nearby
    T.append([[0.0, 0.0, 0.05, 1, 0, 0, 0, 3.0, 3.0, 0.05]] + parked(99))
    # variant 1: the slab again (joints-at-speed cases hang in the air above it; a box terrain so that the stairs model is the one pinned)
    T.append([[0.0, 0.0, 0.05, 1, 0, 0, 0, 3.0, 3.0, 0.05]] + parked(99))
    # variant 2: 90 tiny 2 x 2 cm tiles (2 mm high, 4 cm pitch) crowd the spawn area with box CENTRES while ten long slabs (3 cm high) have theirs
    # 0.9 m away: a foot on a slab is often not among the max_geom_pairs = 25 nearest centre pairs (tests/test_gpu_parity.py::dense_terrain)
    rows = []
    for i in range(10):
        for j in range(9):
            rows.append([(i - 4.5) * 0.04, (j - 4.0) * 0.04, 0.001, 1, 0, 0, 0, 0.01, 0.01, 0.001])
    for k in range(5):
        y = (k - 2) * 0.25
        rows.append([0.9, y, 0.015, 1, 0, 0, 0, 0.85, 0.10, 0.015])
        rows.append([-0.9, y, 0.015, 0, 0, 0, 1, 0.85, 0.10, 0.015])
    T.append(rows)
    return np.asarray(T, dtype=np.float32)


def targeted_cases(key_qpos: np.ndarray) -> List[Dict[str, Any]]:
    """states of the TARGETED group, one per env: (what, variant, qpos, qvel); ctrl = the keyframe's joint angles, action = 0"""
    cases = []
    stand = float(key_qpos[2])                                         # keyframe height of the base above the plane its feet stand on (0.28: feet ~ touching)
    for depth in (0.005, 0.012, 0.017, 0.018, 0.025, 0.040):           # (i): straddles the 17.5 mm foot radius
        q = key_qpos.copy(); q[2] = 0.10 + stand - depth
        cases.append(dict(what="deep_%02dmm" % round(depth * 1000), variant=0, qpos=q, qvel=np.zeros(18)))
    rng = np.random.default_rng(11)
    for k in range(6):                                                 # (ii): in the air, joints at +- 3 .. 15 rad/s
        q = key_qpos.copy(); q[2] = 0.10 + 0.60
        v = np.zeros(18); v[6:] = rng.choice([-1.0, 1.0], 12) * rng.uniform(3.0, 15.0, 12)
        cases.append(dict(what="joint_speed_%d" % k, variant=1, qpos=q, qvel=v))
    for k in range(12):                                                # (iii): standing / slightly sunk on the dense terrain, spread over the slabs
        q = key_qpos.copy()
        q[0], q[1] = rng.uniform(-0.45, 0.45), rng.uniform(-0.45, 0.45)
        yaw = rng.uniform(-3.14, 3.14); q[3:7] = [np.cos(yaw / 2), 0.0, 0.0, np.sin(yaw / 2)]
        q[2] = 0.03 + stand - 0.004
        cases.append(dict(what="rbound_cut_%d" % k, variant=2, qpos=q, qvel=np.zeros(18)))
    return cases


# ====================================================================================================================== backend: real MJX
class MjxBackend:
    """the reference itself: go2.joystick_pgtt.Joystick on mujoco.mjx (needs the packages the reference imports; cwd = the reference root,
    its XML paths are relative: go2/go2_constants.py:21)"""
    name = "mjx"

    def __init__(self, ref: Optional[str], standins: Optional[Dict[str, Any]] = None):
        """`standins`: the module set of tools/fake_mjx.install() instead of the real packages (--dry-run); everything below this constructor is the
        same code either way"""
        self.missing: List[str] = []
        if standins is not None:
            self.name = "dry-run(oracle behind stand-in jax / mujoco / mjx / go2 modules)"
            for k, v in standins.items():
                setattr(self, k, v)
            return
        self.ref = os.path.abspath(ref)
        os.chdir(self.ref)
        sys.path.insert(0, self.ref)
        import jax
        import jax.numpy as jp
        import mujoco
        from mujoco import mjx
        from mujoco_playground._src import mjx_env
        jax.config.update("jax_default_matmul_precision", "highest")             # training/train.py:94
        self.jax, self.jp, self.mujoco, self.mjx, self.mjx_env = jax, jp, mujoco, mjx, mjx_env
        import go2.configs as rconfigs
        import go2.joystick_pgtt as jpg
        import go2.randomize as rrand
        self.rconfigs, self.jpg, self.rrand = rconfigs, jpg, rrand

    def versions(self) -> Dict[str, str]:
        import importlib.metadata as md
        out = {"python": sys.version.split()[0], "numpy": np.__version__}
        for pkg in ("mujoco", "mujoco-mjx", "jax", "jaxlib", "playground", "mujoco_playground", "brax", "ml_collections", "scipy"):
            try:
                out[pkg] = md.version(pkg)
            except Exception:
                pass
        out["mujoco.__version__"] = getattr(self.mujoco, "__version__", "?")
        out["jax_backend"] = self.jax.default_backend()
        return out

    # ---- env construction
    def _config(self):
        cfg = self.rconfigs.default_config()
        cfg.command_config.u_max = [0.6, 0.6, 1.0]; cfg.command_config.u_min = [-0.6, -0.6, -1.0]; cfg.gait_freq = [1, 3]      # training/train.py:127-129
        cfg.noise_config.level = 0.0
        return cfg

    def make_batch(self, task: str, terrain: Optional[np.ndarray], n: int, seed: int, dr: bool) -> Dict[str, Any]:
        jax, jp = self.jax, self.jp
        env = self.jpg.Joystick(task=task, config=self._config())
        keys = jax.random.split(jax.random.PRNGKey(seed), n)
        # Joystick.reset leaves attributes on the env object that Joystick.step reads (`_weights`, joystick_pgtt.py:129): one plain reset on the
        # base object, so that the shallow copies below carry them (the training wrappers call reset and step on one object)
        env.reset(keys[0])
        if terrain is not None:
            # the reference's own randomization_fn places the boxes (go2/randomize.py:97-108); with dr=False the dynamics fields are put back to nominal
            model_v, in_axes = self.rrand.domain_randomize(env.mjx_model, keys, jp.asarray(terrain))
            if not dr:
                nominal = env.mjx_model
                keep = {k: jp.broadcast_to(getattr(nominal, k), (n,) + getattr(nominal, k).shape) for k in DR_FIELDS if k not in ("body_pos", "body_quat", "geom_size")}
                model_v = model_v.tree_replace(keep)
        else:
            model_v, in_axes = env.mjx_model, None

        def with_model(m):
            e = copy.copy(env); e._mjx_model = m            # what Playground's domain-randomisation vmap wrapper does (training/train.py:255,262)
            return e
        ax = in_axes if in_axes is not None else None
        reset = jax.jit(jax.vmap(lambda m, k: with_model(m).reset(k), in_axes=(ax, 0)))
        step = jax.jit(jax.vmap(lambda m, s, a: with_model(m).step(s, a), in_axes=(ax, 0, 0)))
        one = jax.jit(jax.vmap(lambda m, d, c: self.mjx.step(m, d.replace(ctrl=c)), in_axes=(ax, 0, 0)))
        fwd = jax.jit(jax.vmap(lambda m, d: self.mjx.forward(m, d), in_axes=(ax, 0)))
        state = reset(model_v, keys)
        return dict(env=env, model_v=model_v, batched=in_axes is not None, n=n, state=state, step=step, one=one, fwd=fwd, terrain=terrain, task=task)

    def rollout(self, h, steps: int, seed: int) -> None:
        rng = np.random.default_rng(seed)
        for _ in range(steps):
            a = np.tanh(rng.normal(size=(h["n"], 12)) * 0.6).astype(np.float32)
            h["state"] = h["step"](h["model_v"], h["state"], self.jp.asarray(a))

    def set_states(self, h, qpos: np.ndarray, qvel: np.ndarray) -> None:
        """overwrite the physics state of every env (TARGETED group): qacc_warmstart = 0, derived quantities recomputed by mjx.forward"""
        jp = self.jp
        d = h["state"].data
        d = d.replace(qpos=jp.asarray(qpos, dtype=d.qpos.dtype), qvel=jp.asarray(qvel, dtype=d.qvel.dtype), qacc_warmstart=jp.zeros_like(d.qacc_warmstart),
                      ctrl=jp.asarray(qpos[:, 7:], dtype=d.ctrl.dtype))
        h["state"] = h["state"].replace(data=h["fwd"](h["model_v"], d))

    def far_timer(self, h) -> None:
        info = dict(h["state"].info)
        info["steps_until_next_cmd"] = self.jp.full_like(info["steps_until_next_cmd"], FAR_TIMER)
        h["state"] = h["state"].replace(info=info)

    # ---- reading
    def _field(self, data, name: str):
        """a Data field by name; newer MJX keeps solver / contact arrays under data._impl"""
        for obj in (data, getattr(data, "_impl", None)):
            if obj is not None and hasattr(obj, name):
                return getattr(obj, name)
        return None

    def _np(self, x) -> np.ndarray:
        return np.asarray(self.jax.device_get(x))

    def read_inputs(self, h) -> Dict[str, np.ndarray]:
        s = h["state"]; d = s.data
        out = {"qpos": self._np(d.qpos), "qvel": self._np(d.qvel), "qacc_warmstart": self._np(d.qacc_warmstart)}
        for k in INFO_KEYS:
            out["info_" + k] = self._np(s.info[k])
        return out

    def read_dr(self, h) -> Dict[str, np.ndarray]:
        m, n = h["model_v"], h["n"]
        out = {}
        for k in DR_FIELDS:
            a = self._np(getattr(m, k))
            out[k] = a if h["batched"] else np.broadcast_to(a, (n,) + a.shape).copy()
        return out

    def _physics_out(self, d, full: bool = True) -> Dict[str, np.ndarray]:
        out = {"qpos": self._np(d.qpos), "qvel": self._np(d.qvel), "qacc_warmstart": self._np(d.qacc_warmstart)}
        if not full:
            return out
        for name in ("qacc", "sensordata", "actuator_force", "qfrc_bias", "qfrc_passive", "qfrc_actuator", "qfrc_constraint", "qfrc_smooth", "qacc_smooth",
                     "efc_force", "efc_D", "efc_aref", "efc_pos", "site_xpos", "site_xmat", "xpos", "xquat", "subtree_com"):
            v = self._field(d, name)
            if v is None:
                self.missing.append(name)
            else:
                out[name] = self._np(v)
        c = self._field(d, "contact")
        if c is None:
            self.missing.append("contact")
        else:
            geom = getattr(c, "geom", None)
            if geom is None:                                  # older MJX: geom1 / geom2
                geom = self.jp.stack([c.geom1, c.geom2], axis=-1)
            out["contact_geom"], out["contact_dist"] = self._np(geom), self._np(c.dist)
            for name in ("pos", "frame", "friction", "solref", "solimp", "includemargin"):
                if hasattr(c, name):
                    out["contact_" + name] = self._np(getattr(c, name))
        return out

    def mjx_step(self, h, ctrl: np.ndarray) -> Dict[str, np.ndarray]:
        d1 = h["one"](h["model_v"], h["state"].data, self.jp.asarray(ctrl, dtype=h["state"].data.ctrl.dtype))
        return self._physics_out(d1)

    def joystick_step(self, h, action: np.ndarray) -> Dict[str, np.ndarray]:
        s1 = h["step"](h["model_v"], h["state"], self.jp.asarray(action, dtype=np.float32))
        out = self._physics_out(s1.data, full=False)          # the solver-level fields of a control step's LAST substep add nothing to the one-mjx.step record
        out["obs_state"], out["obs_priv"] = self._np(s1.obs["state"]), self._np(s1.obs["privileged_state"])
        out["reward"], out["done"] = self._np(s1.reward), self._np(s1.done)
        keys = list(self._config().reward_config.scales.keys())
        out["metrics"] = np.stack([self._np(s1.metrics[f"reward/{k}"]) for k in keys] + [self._np(s1.metrics["swing_peak"])], axis=-1)
        out["metric_keys"] = np.array(keys + ["swing_peak"])
        for k in INFO_KEYS:
            out["info_" + k] = self._np(s1.info[k])
        out["scan_z"] = self._np(s1.info["heightscan"])[..., 2].reshape(h["n"], -1)
        return out

    # ---- model constants and id tables
    def geom_ids(self, h) -> Dict[str, np.ndarray]:
        env = h["env"]
        return {"feet_geom_id": self._np(env._feet_geom_id), "floor_geom_id": self._np(env._floor_geom_id)}      # go2/base.py:87-105

    def model_constants(self, h) -> Dict[str, np.ndarray]:
        env, mujoco = h["env"], self.mujoco
        m = env.mj_model
        out: Dict[str, Any] = {}
        for k in ("timestep", "gravity", "impratio", "tolerance", "ls_tolerance", "iterations", "ls_iterations", "cone", "jacobian", "solver", "integrator", "disableflags"):
            out["opt_" + k] = np.asarray(getattr(m.opt, k))
        out["stat_meaninertia"] = np.asarray(m.stat.meaninertia)
        for k in ("body_mass", "body_inertia", "body_ipos", "body_iquat", "body_pos", "body_quat", "body_invweight0", "body_parentid", "dof_invweight0", "dof_armature", "dof_damping",
                  "dof_frictionloss", "jnt_range", "jnt_axis", "jnt_solref", "jnt_solimp", "jnt_type", "qpos0", "actuator_gainprm", "actuator_biasprm", "actuator_ctrlrange",
                  "actuator_forcerange", "actuator_trnid", "geom_friction", "geom_solref", "geom_solimp", "geom_margin", "geom_gap", "geom_solmix", "geom_condim", "geom_size",
                  "geom_pos", "geom_quat", "geom_rbound", "geom_bodyid", "geom_type", "geom_contype", "geom_conaffinity", "geom_group", "site_pos", "site_bodyid", "sensor_adr",
                  "sensor_dim", "sensor_type"):
            out[k] = np.asarray(getattr(m, k))
        out["key_qpos"] = np.asarray(m.keyframe("home").qpos)
        for name in ("max_contact_points", "max_geom_pairs"):
            try:
                out["numeric_" + name] = np.asarray(m.numeric(name).data)
            except Exception:
                self.missing.append("numeric_" + name)
        out["imu_site_id"] = np.asarray(env._imu_site_id)
        out["feet_site_id"] = self._np(env._feet_site_id)
        out["nbody"], out["ngeom"] = np.asarray(m.nbody), np.asarray(m.ngeom)
        out["sensor_names"] = np.array([mujoco.mj_id2name(m, mujoco.mjtObj.mjOBJ_SENSOR, i) for i in range(m.nsensor)])
        return out


# ====================================================================================================================== recorder
def record_group(be, name: str, task: str, terrain: Optional[np.ndarray], n: int, seed: int, dr: bool, roll: int, cases=None) -> Dict[str, np.ndarray]:
    """one group of the fixture: n envs brought to a state (a roll-out of `roll` steps, or the crafted `cases`), then from THAT state one mjx.step
    and one Joystick.step, inputs and outputs recorded.  Keys: '<name>/in_*', '<name>/dr_*', '<name>/mjx_*', '<name>/step_*'."""
    h = be.make_batch(task, terrain, n, seed, dr)
    if cases is None:
        be.rollout(h, roll, seed + 1)
    else:
        be.set_states(h, np.stack([c["qpos"] for c in cases]), np.stack([c["qvel"] for c in cases]))
    be.far_timer(h)
    out: Dict[str, np.ndarray] = {}
    ins = be.read_inputs(h)
    rng = np.random.default_rng(seed + 2)
    action = np.tanh(rng.normal(size=(n, 12)) * 0.6) if cases is None else np.zeros((n, 12))
    key = be.model_constants(h)["key_qpos"]
    ctrl = np.asarray(key[7:])[None] + action * 0.5                        # motor_targets = default_pose + action * action_scale (joystick_pgtt.py:145)
    for k, v in ins.items():
        out[f"{name}/in_{k}"] = np.asarray(v)
    out[f"{name}/in_action"], out[f"{name}/in_ctrl"] = action, ctrl
    for k, v in be.read_dr(h).items():
        out[f"{name}/dr_{k}"] = np.asarray(v)
    for k, v in be.mjx_step(h, ctrl).items():
        out[f"{name}/mjx_{k}"] = np.asarray(v)
    for k, v in be.joystick_step(h, action).items():
        out[f"{name}/step_{k}"] = np.asarray(v)
    for k, v in be.geom_ids(h).items():
        out[f"{name}/ids_{k}"] = np.asarray(v)
    if terrain is not None:
        out[f"{name}/terrain"] = np.asarray(terrain, dtype=np.float32)
    if cases is not None:
        out[f"{name}/case_what"] = np.array([c["what"] for c in cases]); out[f"{name}/case_variant_wanted"] = np.array([c["variant"] for c in cases])
    out[f"{name}/task"] = np.array(task)
    return out, h


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--ref", default="/root/reference", help="checkout of NtagkasAlex/phase_guided_terrain_traversal (imported, never copied)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "mjx_step.npz"))
    ap.add_argument("--dry-run", action="store_true", help="stand-in modules backed by the repo's CPU oracle instead of jax / mujoco / mjx / the reference (tools/fake_mjx.py; plumbing check, pins nothing)")
    ap.add_argument("--envs-flat", type=int, default=96)
    ap.add_argument("--envs-level4", type=int, default=128)
    ap.add_argument("--roll", type=int, default=30, help="control steps of random actions before the recorded state")
    ap.add_argument("--level4", default=None, help="terrain matrix file (default: <ref>/terrains/level4.npy; dry run: the repo's copy)")
    a = ap.parse_args(argv)
    if a.dry_run and os.path.abspath(a.out) == os.path.join(ROOT, "tests", "golden", "mjx_step.npz"):
        raise SystemExit("--dry-run must not write tests/golden/mjx_step.npz (that name is reserved for vectors of the real MJX): pass --out")
    if a.dry_run:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import fake_mjx
        be = MjxBackend(None, standins=fake_mjx.install())
    else:
        be = MjxBackend(a.ref)
    lvl = a.level4 or (os.path.join(ROOT, "phase_guided_terrain_traversal_amd", "assets", "terrains", "level4.npy") if a.dry_run else os.path.join(a.ref, "terrains", "level4.npy"))
    level4 = np.load(lvl).astype(np.float32)
    data: Dict[str, np.ndarray] = {}
    g, h = record_group(be, "flat", "flat_terrain", None, a.envs_flat, 101, False, a.roll); data.update(g)
    g, h = record_group(be, "level4", "stairs", level4, a.envs_level4, 202, True, a.roll); data.update(g)
    for k, v in be.model_constants(h).items():
        data["model/" + k] = np.asarray(v)
    tt = targeted_terrain()
    cases = targeted_cases(np.asarray(data["model/key_qpos"], dtype=np.float64))
    # the reference's randomization_fn draws the variant at random (randomize.py:97-100): enough envs per case that the wanted one occurs is not
    # controllable from outside, so the TARGETED group is recorded once per variant on a ONE-variant terrain matrix
    for v in range(tt.shape[0]):
        cs = [c for c in cases if c["variant"] == v]
        g, _ = record_group(be, f"targeted{v}", "stairs", tt[v:v + 1], len(cs), 303 + v, False, 0, cases=cs); data.update(g)
    meta = {"format": FORMAT_VERSION, "backend": be.name, "dry_run": bool(a.dry_run), "versions": be.versions(), "missing": sorted(set(be.missing)),
            "groups": ["flat", "level4"] + [f"targeted{v}" for v in range(tt.shape[0])], "far_timer": FAR_TIMER, "feet_order": FEET, "info_keys": INFO_KEYS,
            "command": " ".join([os.path.basename(sys.argv[0])] + (argv if argv is not None else sys.argv[1:]))}
    data["meta"] = np.array(json.dumps(meta))
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    np.savez_compressed(a.out, **data)
    print(f"wrote {a.out}: {len(data)} arrays, {os.path.getsize(a.out) / 1e6:.2f} MB, backend {be.name}, missing fields: {meta['missing']}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
