"""Host-side model compiler: MJCF subset -> flat Go2 model constants (``PgttModel``).

The reference builds its model with MuJoCo's compiler (``mujoco.MjModel.from_xml_string``,
reference go2/base.py:53-55) and then patches a few fields (go2/base.py:57-61).  MuJoCo is not
available here, so this module is a from-scratch reader for exactly the MJCF subset the trained
model uses (go2/xmls/go2_mjx_feetonly.xml + scene_mjx_feetonly.xml / terrain_scene_mjx.xml):
``include``, ``compiler angle/autolimits``, ``option`` (+ ``flag``), ``custom/numeric``, nested
``default`` classes (joint / geom / site / general), ``body``/``inertial``/``freejoint``/``joint``/
``geom``/``site``, ``position`` actuators, the ``home`` keyframe, and the sensor *list* (the 16
sensors are hard-wired in the simulator; the compiler only checks that the file declares them).

Derived constants follow MuJoCo's compile-time ``set0`` at qpos0: joint-space inertia M0 (computed
here from body Jacobians, independent of the CRBA used by the simulator), ``dof_invweight0`` =
diag(M0^-1) (free joint: mean over the translational / rotational triplets), ``body_invweight0`` =
(tr(Jp M0^-1 Jp^T)/3, tr(Jr M0^-1 Jr^T)/3) at the body COM, ``stat.meaninertia`` = mean diag(M0).

All arithmetic is float64; the product casts to float32 when filling ``PgttModel``.
"""
from __future__ import annotations

import copy
import json
import os
import xml.etree.ElementTree as ET
from typing import Any, Dict, List, Optional

import numpy as np

# expected topology / names (reference go2/go2_constants.py:55-83, go2_mjx_feetonly.xml)
LEGS = ["FL", "FR", "RL", "RR"]                      # body-tree order
FEET_ORDER = ["FR", "FL", "RR", "RL"]                # FEET_SITES / FEET_GEOMS / actuator order
EXPECTED_SENSORS = [
    ("gyro", "gyro"), ("accelerometer", "accelerometer"), ("framequat", "orientation"),
    ("framepos", "global_position"), ("framelinvel", "global_linvel"),
    ("frameangvel", "global_angvel"), ("velocimeter", "local_linvel"), ("framezaxis", "upvector"),
    ("framepos", "FR_pos"), ("framepos", "FL_pos"), ("framepos", "RR_pos"), ("framepos", "RL_pos"),
    ("framelinvel", "FR_foot_global_linvel"), ("framelinvel", "FL_foot_global_linvel"),
    ("framelinvel", "RR_foot_global_linvel"), ("framelinvel", "RL_foot_global_linvel"),
]

_GEOM_DEFAULTS = dict(type="sphere", friction="1 0.005 0.0001", solref="0.02 1",
                      solimp="0.9 0.95 0.001 0.5 2", margin="0", gap="0", solmix="1",
                      condim="3", contype="1", conaffinity="1", group="0", pos="0 0 0",
                      quat="1 0 0 0", priority="0")
_JOINT_DEFAULTS = dict(type="hinge", axis="0 0 1", pos="0 0 0", damping="0", armature="0",
                       frictionloss="0", stiffness="0", margin="0", solreflimit="0.02 1",
                       solimplimit="0.9 0.95 0.001 0.5 2")
_GENERAL_DEFAULTS = dict(gainprm="1 0 0", biasprm="0 0 0", gear="1 0 0 0 0 0")
_SITE_DEFAULTS = dict(pos="0 0 0", quat="1 0 0 0", group="0")


def _vec(s: str, n: Optional[int] = None) -> np.ndarray:
    v = np.array([float(x) for x in s.split()], dtype=np.float64)
    if n is not None and v.size < n:
        v = np.concatenate([v, np.zeros(n - v.size)])
    return v


def _vec_d(s: str, default: str) -> np.ndarray:
    """Partial attribute: the given leading values override MuJoCo's defaults, the tail keeps them."""
    d = _vec(default)
    v = _vec(s)
    d[:v.size] = v[:d.size]
    return d


def _load_tree(path: str) -> ET.Element:
    """Parse an MJCF file, splicing <include file=.../> elements in place (MuJoCo semantics)."""
    root = ET.parse(path).getroot()
    base = os.path.dirname(path)

    def splice(parent: ET.Element):
        i = 0
        while i < len(parent):
            child = parent[i]
            if child.tag == "include":
                inc = _load_tree(os.path.join(base, child.attrib["file"]))
                parent.remove(child)
                for k, sub in enumerate(list(inc)):
                    parent.insert(i + k, sub)
                i += len(inc)
            else:
                splice(child)
                i += 1
    splice(root)
    return root


class _Defaults:
    """Nested <default class=...> resolution: child classes inherit every element of the parent."""

    def __init__(self, root: ET.Element):
        self.classes: Dict[str, Dict[str, Dict[str, str]]] = {"main": {}}
        for d in root.findall("default"):
            self._walk(d, "main", top=True)

    def _walk(self, node: ET.Element, parent: str, top: bool = False):
        name = node.attrib.get("class", "main" if top else None)
        if name is None:
            raise ValueError("nested <default> without class")
        cur = copy.deepcopy(self.classes.get(parent, {})) if name != "main" else self.classes["main"]
        for child in node:
            if child.tag == "default":
                continue
            cur.setdefault(child.tag, {}).update(child.attrib)
        self.classes[name] = cur
        for child in node:
            if child.tag == "default":
                self._walk(child, name)

    def get(self, cls: Optional[str], tag: str) -> Dict[str, str]:
        return dict(self.classes.get(cls or "main", {}).get(tag, {}))


def _quat_mul(a, b):
    return np.array([
        a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3],
        a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2],
        a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1],
        a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0]])


def _quat_to_mat(q):
    w, x, y, z = q
    return np.array([
        [w*w + x*x - y*y - z*z, 2*(x*y - w*z), 2*(x*z + w*y)],
        [2*(x*y + w*z), w*w - x*x + y*y - z*z, 2*(y*z - w*x)],
        [2*(x*z - w*y), 2*(y*z + w*x), w*w - x*x - y*y + z*z]])


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def compile_mjcf(path: str, sim_dt: float = 0.005, Kp: float = 40.0, Kd: float = 0.5) -> Dict[str, Any]:
    """Compile the Go2 feet-only MJCF at ``path`` into a dict of numpy arrays (PgttModel fields).

    ``sim_dt``/``Kp``/``Kd`` reproduce the post-compile overrides of reference go2/base.py:57-61
    (timestep, dof_damping[6:], gainprm[:,0], biasprm[:,1]).
    """
    root = _load_tree(path)
    comp = {}
    for c in root.findall("compiler"):
        comp.update(c.attrib)
    if comp.get("angle", "degree") != "radian":
        raise ValueError("only angle=radian is supported")
    dfl = _Defaults(root)

    # options: later <option> elements override earlier attributes (go2_mjx_feetonly.xml:4-6,17-19)
    opt = dict(timestep="0.002", gravity="0 0 -9.81", impratio="1", tolerance="1e-8",
               ls_tolerance="0.01", iterations="100", ls_iterations="50", cone="pyramidal",
               integrator="Euler", solver="Newton")
    flags = {}
    for o in root.findall("option"):
        opt.update(o.attrib)
        for f in o.findall("flag"):
            flags.update(f.attrib)
    if opt["cone"] != "pyramidal" or opt["integrator"] != "Euler" or opt["solver"] != "Newton":
        raise ValueError("simulator implements cone=pyramidal, integrator=Euler, solver=Newton only")
    if flags.get("eulerdamp", "enable") != "disable":
        raise ValueError("simulator implements eulerdamp=disable only")
    numeric = {n.attrib["name"]: float(n.attrib["data"]) for c in root.findall("custom")
               for n in c.findall("numeric")}

    # ---- walk the body tree (DFS pre-order = MuJoCo body ids)
    bodies: List[Dict[str, Any]] = []
    geoms: List[Dict[str, Any]] = []
    sites: List[Dict[str, Any]] = []
    joints: List[Dict[str, Any]] = []

    def elem_attrs(e: ET.Element, tag: str, childclass: Optional[str], base: Dict[str, str]):
        a = dict(base)
        a.update(dfl.get(e.attrib.get("class", childclass), tag))
        a.update(e.attrib)
        return a

    def add_geom(e, body_id, childclass):
        a = elem_attrs(e, "geom", childclass, _GEOM_DEFAULTS)
        geoms.append(dict(name=a.get("name"), body=body_id, type=a["type"],
                          size=_vec(a.get("size", "0 0 0"), 3), pos=_vec(a["pos"]), quat=_vec(a["quat"]),
                          friction=_vec_d(a["friction"], _GEOM_DEFAULTS["friction"]), solref=_vec_d(a["solref"], _GEOM_DEFAULTS["solref"]),
                          solimp=_vec_d(a["solimp"], _GEOM_DEFAULTS["solimp"]),
                          margin=float(a["margin"]), gap=float(a["gap"]), solmix=float(a["solmix"]),
                          condim=int(a["condim"]), contype=int(a["contype"]),
                          conaffinity=int(a["conaffinity"]), group=int(a["group"]),
                          priority=int(a["priority"])))

    def walk(e: ET.Element, parent: int, childclass: Optional[str]):
        bid = len(bodies)
        cc = e.attrib.get("childclass", childclass)
        b = dict(name=e.attrib.get("name", "world" if parent < 0 else None), parent=parent,
                 pos=_vec(e.attrib.get("pos", "0 0 0")), quat=_vec(e.attrib.get("quat", "1 0 0 0")),
                 ipos=np.zeros(3), iquat=np.array([1.0, 0, 0, 0]), mass=0.0, inertia=np.zeros(3),
                 joints=[])
        bodies.append(b)
        for c in e:
            if c.tag == "inertial":
                b["ipos"] = _vec(c.attrib["pos"])
                q = _vec(c.attrib.get("quat", "1 0 0 0"))
                b["iquat"] = q / np.linalg.norm(q)
                b["mass"] = float(c.attrib["mass"])
                b["inertia"] = _vec(c.attrib["diaginertia"])
            elif c.tag == "freejoint":
                joints.append(dict(name=c.attrib.get("name"), body=bid, type="free"))
                b["joints"].append(len(joints) - 1)
            elif c.tag == "joint":
                a = elem_attrs(c, "joint", cc, _JOINT_DEFAULTS)
                rng = _vec(a["range"]) if "range" in a else None
                joints.append(dict(name=a.get("name"), body=bid, type=a["type"], axis=_vec(a["axis"]),
                                   pos=_vec(a["pos"]), range=rng, damping=float(a["damping"]),
                                   armature=float(a["armature"]), frictionloss=float(a["frictionloss"]),
                                   stiffness=float(a["stiffness"]),
                                   solref=_vec_d(a["solreflimit"], _JOINT_DEFAULTS["solreflimit"]),
                                   solimp=_vec_d(a["solimplimit"], _JOINT_DEFAULTS["solimplimit"]),
                                   margin=float(a["margin"])))
                b["joints"].append(len(joints) - 1)
            elif c.tag == "geom":
                add_geom(c, bid, cc)
            elif c.tag == "site":
                a = elem_attrs(c, "site", cc, _SITE_DEFAULTS)
                sites.append(dict(name=a.get("name"), body=bid, pos=_vec(a["pos"]), quat=_vec(a["quat"])))
        for c in e:
            if c.tag == "body":
                walk(c, bid, cc)

    # several <worldbody> elements (robot file + scene file) merge into body 0; MuJoCo numbers the
    # world's own geoms first, so process direct geoms of every worldbody before the child bodies.
    world = ET.Element("body")
    for wb in root.findall("worldbody"):
        for c in wb:
            world.append(c)
    walk(world, -1, None)
    # geoms were appended in DFS order already (world geoms first because geoms precede child bodies
    # in `walk`); MuJoCo orders geoms by body id, which for a DFS walk is the same order.
    geoms.sort(key=lambda g: g["body"])  # stable

    # ---- topology checks
    names = [b["name"] for b in bodies]
    exp = ["world", "base"] + [f"{l}_{p}" for l in LEGS for p in ("hip", "thigh", "calf")]
    if names[:14] != exp:
        raise ValueError(f"unexpected body tree {names[:14]}")
    box_bodies = [i for i, b in enumerate(bodies) if i >= 14]
    for i in box_bodies:
        if not (bodies[i]["name"] or "").startswith("box_") or bodies[i]["parent"] != 0:
            raise ValueError("bodies after the robot must be static box_* placeholders")
    if joints[0]["type"] != "free" or len(joints) != 13 or any(j["type"] != "hinge" for j in joints[1:]):
        raise ValueError("expected one free joint + 12 hinges")
    for k, j in enumerate(joints[1:]):
        if j["body"] != 2 + k or np.any(j["pos"] != 0):
            raise ValueError("hinge k must belong to body 2+k with pos 0")

    # ---- geoms: floor, feet, boxes
    gname = {g["name"]: i for i, g in enumerate(geoms) if g["name"]}
    floor = geoms[gname["floor"]]
    if gname["floor"] != 0 or floor["type"] != "plane":
        raise ValueError("floor plane must be geom 0 (reference go2/randomize.py:21)")
    collide = lambda a, b: (a["contype"] & b["conaffinity"]) or (b["contype"] & a["conaffinity"])
    feet = [geoms[gname[l]] for l in LEGS]
    for l, g in zip(LEGS, feet):
        if g["type"] != "sphere" or g["body"] != 4 + 3 * LEGS.index(l):
            raise ValueError("foot geoms must be spheres on the calf bodies")
    for i, g in enumerate(geoms):
        if g["body"] in range(1, 14) and g["name"] not in LEGS and (collide(g, floor)):
            raise ValueError(f"geom {i} besides the feet can collide; only feet-only models are supported")
    boxes = [g for g in geoms if g["body"] >= 14]
    for g in boxes:
        if g["type"] != "box":
            raise ValueError("terrain placeholders must be boxes")
    foot_geom_ids = [gname[l] for l in LEGS]
    first_box_geom = min([i for i, g in enumerate(geoms) if g["body"] >= 14], default=-1)

    sname = {s["name"]: s for s in sites}
    imu = sname["imu"]
    if imu["body"] != 1 or np.any(imu["quat"] != np.array([1.0, 0, 0, 0])):
        raise ValueError("imu site must sit on the base with identity orientation")

    # ---- sensors (only verified, the simulator hard-wires them)
    sens = [(s.tag, s.attrib.get("name")) for sn in root.findall("sensor") for s in sn]
    if sens != EXPECTED_SENSORS:
        raise ValueError(f"unexpected sensor list {sens}")

    # ---- actuators
    jname = {j["name"]: i for i, j in enumerate(joints)}
    act = []
    for an in root.findall("actuator"):
        for a in an:
            if a.tag != "position":
                raise ValueError("only <position> actuators are supported")
            at = dict(_GENERAL_DEFAULTS)
            at.update(dfl.get(a.attrib.get("class"), "general"))
            at.update(a.attrib)
            gain = _vec(at["gainprm"], 3)
            bias = _vec(at["biasprm"], 3)
            if "kp" in at:
                gain[0] = float(at["kp"])
            bias[1] = -gain[0]            # mjs_setToPosition: biasprm[1] = -kp; biasprm[2] kept unless kv given
            if "kv" in at:
                bias[2] = -float(at["kv"])
            j = jname[at["joint"]]
            act.append(dict(name=at.get("name"), joint=j, dof=6 + (j - 1), gain=gain, bias=bias,
                            ctrlrange=_vec(at["ctrlrange"]), forcerange=_vec(at["forcerange"])))
    if len(act) != 12:
        raise ValueError("expected 12 actuators")
    exp_act = [f"{l}_{p}" for l in FEET_ORDER for p in ("hip", "thigh", "calf")]
    if [a["name"] for a in act] != exp_act:
        raise ValueError("actuators must be declared FR,FL,RR,RL (go2_mjx_feetonly.xml:219-230)")

    key = None
    for kf in root.findall("keyframe"):
        for k in kf:
            if k.attrib.get("name") == "home":
                key = _vec(k.attrib["qpos"])
    if key is None or key.size != 19:
        raise ValueError("keyframe 'home' with 19 qpos values required")

    # ---- assemble arrays for the 13 moving bodies (index 0 = base)
    mb = bodies[1:14]
    m: Dict[str, Any] = {}
    m["body_pos"] = np.array([b["pos"] for b in mb])
    m["body_quat"] = np.array([b["quat"] / np.linalg.norm(b["quat"]) for b in mb])
    m["body_ipos"] = np.array([b["ipos"] for b in mb])
    m["body_iquat"] = np.array([b["iquat"] for b in mb])
    m["body_mass"] = np.array([b["mass"] for b in mb])
    m["body_inertia"] = np.array([b["inertia"] for b in mb])
    hj = joints[1:]
    m["jnt_axis"] = np.array([j["axis"] / np.linalg.norm(j["axis"]) for j in hj])
    autolimits = comp.get("autolimits", "true") == "true"
    for j in hj:
        if j["range"] is None or not autolimits:
            raise ValueError("every hinge needs a range (autolimits)")
        if j["margin"] != 0 or j["frictionloss"] != 0 or j["stiffness"] != 0:
            raise ValueError("joint margin/frictionloss/stiffness must be 0")
    m["jnt_range"] = np.array([j["range"] for j in hj])
    m["jnt_solref"] = hj[0]["solref"].copy()
    m["jnt_solimp"] = hj[0]["solimp"].copy()
    qpos0 = np.zeros(19)
    qpos0[0:3] = bodies[1]["pos"]
    qpos0[3:7] = bodies[1]["quat"]
    m["qpos0"] = qpos0
    m["dof_armature"] = np.concatenate([np.zeros(6), [j["armature"] for j in hj]])
    m["dof_damping"] = np.concatenate([np.zeros(6), [Kd for _ in hj]])          # base.py:59
    m["act_dof"] = np.array([a["dof"] for a in act], dtype=np.int32)
    m["act_gain"] = np.array([Kp for _ in act])                                   # base.py:60
    bias = np.array([a["bias"] for a in act])
    bias[:, 1] = -Kp                                                               # base.py:61
    m["act_bias"] = bias
    m["act_ctrlrange"] = np.array([a["ctrlrange"] for a in act])
    m["act_forcerange"] = np.array([a["forcerange"] for a in act])
    m["foot_geom_pos"] = np.array([g["pos"] for g in feet])
    m["foot_radius"] = np.array([g["size"][0] for g in feet])
    m["foot_site_pos"] = np.array([sname[f"{l}_foot"]["pos"] for l in LEGS])
    m["imu_pos"] = imu["pos"].copy()
    for pre, g in (("foot", feet[0]), ("floor", floor), ("box", boxes[0] if boxes else floor)):
        m[f"{pre}_friction"] = g["friction"].copy()
        m[f"{pre}_solref"] = g["solref"].copy()
        m[f"{pre}_solimp"] = g["solimp"].copy()
        m[f"{pre}_margin"] = g["margin"]
        m[f"{pre}_gap"] = g["gap"]
        m[f"{pre}_solmix"] = g["solmix"]
        m[f"{pre}_condim"] = g["condim"]
    m["box_rbound"] = float(np.linalg.norm(boxes[0]["size"])) if boxes else float(np.sqrt(3.0))
    m["timestep"] = float(sim_dt)                                                  # base.py:57
    m["gravity"] = _vec(opt["gravity"])
    m["impratio"] = float(opt["impratio"])
    m["tolerance"] = float(opt["tolerance"])
    m["ls_tolerance"] = float(opt["ls_tolerance"])
    m["iterations"] = int(opt["iterations"])
    m["ls_iterations"] = int(opt["ls_iterations"])
    m["max_geom_pairs"] = int(numeric.get("max_geom_pairs", -1))
    m["max_contact_points"] = int(numeric.get("max_contact_points", -1))
    m["key_qpos"] = key
    # bookkeeping (not part of PgttModel, used by tests / host code)
    m["_nbox"] = len(boxes)
    m["_foot_geom_ids"] = foot_geom_ids
    m["_first_box_geom"] = first_box_geom
    m["_nbody"] = len(bodies)
    m["_ngeom"] = len(geoms)
    m["_body_names"] = names[:14]

    _derive_constants(m)
    return m


def kinematics_np(m: Dict[str, Any], qpos: np.ndarray):
    """float64 kinematics of the 13 moving bodies (used for derived constants and tests)."""
    xpos = np.zeros((13, 3)); xquat = np.zeros((13, 4))
    xpos[0] = qpos[0:3]
    xquat[0] = qpos[3:7] / np.linalg.norm(qpos[3:7])
    for b in range(1, 13):
        leg, k = divmod(b - 1, 3)
        parent = 0 if k == 0 else b - 1
        R = _quat_to_mat(xquat[parent])
        pos = xpos[parent] + R @ m["body_pos"][b]
        quat = _quat_mul(xquat[parent], m["body_quat"][b])
        ang = qpos[7 + b - 1] - m["qpos0"][7 + b - 1]
        ax = m["jnt_axis"][b - 1]
        qloc = np.concatenate([[np.cos(ang / 2)], ax * np.sin(ang / 2)])
        xquat[b] = _quat_mul(quat, qloc)
        xpos[b] = pos
    xmat = np.array([_quat_to_mat(q) for q in xquat])
    xipos = np.array([xpos[b] + xmat[b] @ m["body_ipos"][b] for b in range(13)])
    ximat = np.array([_quat_to_mat(_quat_mul(xquat[b], m["body_iquat"][b])) for b in range(13)])
    return xpos, xquat, xmat, xipos, ximat


def jacobians_np(m: Dict[str, Any], xpos, xmat, point, body: int):
    """(3,18) translational and rotational Jacobians of a world point rigidly attached to `body`."""
    jp = np.zeros((3, 18)); jr = np.zeros((3, 18))
    jp[:, 0:3] = np.eye(3)
    for i in range(3):                                   # free-joint rotation is body-frame
        ax = xmat[0][:, i]
        jr[:, 3 + i] = ax
        jp[:, 3 + i] = np.cross(ax, point - xpos[0])
    if body > 0:
        leg, k = divmod(body - 1, 3)
        for kk in range(k + 1):
            b = 1 + 3 * leg + kk
            ax = xmat[b] @ m["jnt_axis"][b - 1]
            jr[:, 6 + b - 1] = ax
            jp[:, 6 + b - 1] = np.cross(ax, point - xpos[b])
    return jp, jr


def mass_matrix_np(m: Dict[str, Any], qpos: np.ndarray, body_mass=None, body_ipos=None, armature=None):
    """Joint-space inertia from body Jacobians: M = sum_b m Jp^T Jp + Jr^T I_w Jr + diag(armature)."""
    mm = dict(m)
    if body_ipos is not None:
        mm["body_ipos"] = body_ipos
    mass = m["body_mass"] if body_mass is None else body_mass
    arm = m["dof_armature"] if armature is None else armature
    xpos, xquat, xmat, xipos, ximat = kinematics_np(mm, qpos)
    M = np.diag(arm).astype(np.float64)
    for b in range(13):
        jp, jr = jacobians_np(mm, xpos, xmat, xipos[b], b)
        Iw = ximat[b] @ np.diag(m["body_inertia"][b]) @ ximat[b].T
        M += mass[b] * jp.T @ jp + jr.T @ Iw @ jr
    return M


def _derive_constants(m: Dict[str, Any]) -> None:
    q0 = m["qpos0"]
    M0 = mass_matrix_np(m, q0)
    Minv = np.linalg.inv(M0)
    d = np.diag(Minv).copy()
    d[0:3] = d[0:3].mean()
    d[3:6] = d[3:6].mean()
    m["dof_invweight0"] = d
    xpos, xquat, xmat, xipos, ximat = kinematics_np(m, q0)
    biw = np.zeros((13, 2))
    for b in range(13):
        jp, jr = jacobians_np(m, xpos, xmat, xipos[b], b)
        biw[b, 0] = np.trace(jp @ Minv @ jp.T) / 3
        biw[b, 1] = np.trace(jr @ Minv @ jr.T) / 3
    m["body_invweight0"] = biw
    m["meaninertia"] = float(np.mean(np.diag(M0)))
    m["_M0"] = M0


# ---------------------------------------------------------------- (de)serialisation of the shipped asset
_ARRAY_FIELDS = ["body_pos", "body_quat", "body_ipos", "body_iquat", "body_mass", "body_inertia",
                 "body_invweight0", "jnt_axis", "jnt_range", "jnt_solref", "jnt_solimp", "qpos0",
                 "dof_armature", "dof_damping", "dof_invweight0", "act_dof", "act_gain", "act_bias",
                 "act_ctrlrange", "act_forcerange", "foot_geom_pos", "foot_radius", "foot_site_pos",
                 "imu_pos", "foot_friction", "foot_solref", "foot_solimp", "floor_friction",
                 "floor_solref", "floor_solimp", "box_friction", "box_solref", "box_solimp", "gravity",
                 "key_qpos"]
_SCALAR_FIELDS = ["foot_margin", "foot_gap", "foot_solmix", "floor_margin", "floor_gap", "floor_solmix",
                  "box_margin", "box_gap", "box_solmix", "box_rbound", "foot_condim", "floor_condim",
                  "box_condim", "timestep", "impratio", "tolerance", "ls_tolerance", "meaninertia",
                  "iterations", "ls_iterations", "max_geom_pairs", "max_contact_points"]


def model_to_json(m: Dict[str, Any]) -> str:
    out = {k: np.asarray(m[k]).tolist() for k in _ARRAY_FIELDS}
    out.update({k: m[k] for k in _SCALAR_FIELDS})
    out["_nbox"] = m["_nbox"]
    return json.dumps(out, indent=1)


def model_from_json(text: str) -> Dict[str, Any]:
    raw = json.loads(text)
    m: Dict[str, Any] = {}
    for k in _ARRAY_FIELDS:
        m[k] = np.asarray(raw[k], dtype=np.int32 if k == "act_dof" else np.float64)
    for k in _SCALAR_FIELDS:
        m[k] = raw[k]
    m["_nbox"] = raw.get("_nbox", 0)
    return m


_ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


def with_bias_velocity(model: Dict[str, Any], kv: float) -> Dict[str, Any]:
    """Copy of a compiled model with the actuators' velocity bias biasprm[:, 2] set to `kv`.  The shipped constants keep the
    default class's -0.5 (go2_mjx_feetonly.xml:27); whether MuJoCo's <position> shortcut keeps or clears it is a parser detail
    that cannot be checked here, so the value is a switch with a test on either side."""
    out = dict(model)
    b = np.array(model["act_bias"], dtype=np.float64, copy=True)
    b[:, 2] = kv
    out["act_bias"] = b
    return out


def asset_path(task: str = "stairs") -> str:
    """the shipped compiled-model asset of a task (the counterpart of go2_constants.task_to_xml, go2/go2_constants.py:45-52)"""
    return os.path.join(_ASSET_DIR, {"flat_terrain": "go2_flat_terrain.json", "stairs": "go2_stairs.json"}[task])


def load_model(task: str = "stairs") -> Dict[str, Any]:
    """Load the shipped, pre-compiled model constants (assets/go2_<task>.json).

    ``task`` follows reference go2/go2_constants.py:45-52: "flat_terrain" (plane only) or "stairs"
    (plane + 100 placeholder boxes filled by the terrain table).
    """
    fn = {"flat_terrain": "go2_flat_terrain.json", "stairs": "go2_stairs.json"}[task]
    with open(os.path.join(_ASSET_DIR, fn)) as f:
        return model_from_json(f.read())
