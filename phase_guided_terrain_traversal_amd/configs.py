"""Task configuration: mirror of the reference's go2/configs.py:6-79 ``default_config()`` plus the
overrides training/train.py:127-129 applies before training (command range, gait frequency).

Plain nested dicts (the reference uses ml_collections.ConfigDict, which is not available here); keys and
values are the reference's.  ``scan_*`` come from go2/go2_constants.py:90-94 and go2/heightmap.py:38.
"""
from __future__ import annotations

import copy
from typing import Any, Dict


def default_config() -> Dict[str, Any]:
    return dict(
        ctrl_dt=0.02, sim_dt=0.005, episode_length=1000, vel_percentage=0.65, Kp=40.0, Kd=0.5,
        action_repeat=1, action_scale=0.5, history_len=2, history_update_steps=5,
        soft_joint_pos_limit_factor=0.95,
        noise_config=dict(level=1.0, scales=dict(joint_pos=0.03, joint_vel=1.5, gyro=0.2, gravity=0.05,
                                                 linvel=0.1, heightscan=0.01)),
        reward_config=dict(
            scales=dict(tracking_lin_vel=1.0, tracking_ang_vel=0.5, lin_vel_z=-1.0, ang_vel_xy=-0.05,
                        orientation=-0.2, dof_pos_limits=-1.0, pose=-1.0, termination=-1.0,
                        stand_still=-0.0, torques=-0.0002, action_rate=-0.01, energy=-0.0005,
                        feet_clearance=-0.0, feet_height=-0.0, feet_slip=-0.0, feet_air_time=0.0,
                        feet_phase=0.5, feet_swing=0.0, body_height=-0.0, contact=2.0, center=-0.0),
            tracking_sigma=0.2, swing_height=-0.2, base_feet_distance=-0.3, phase_sigma=0.05),
        command_config=dict(u_max=[1.5, 0.8, 1.2], u_min=[-1.5, -0.8, -1.2], b=[0.9, 0.25, 0.5]),
        gait_freq=[2, 6],
        heighmap_size=(13, 9),
        scan_dist_x=0.1, scan_dist_y=0.1, scan_z_offset=0.6,
        autoreset=0,
        method="pgtt",
    )


def baseline_config() -> Dict[str, Any]:
    """go2/configs.py:82-152 ``baseline_config()``: the comparison task go2/joystick.py is trained with."""
    cfg = default_config()
    cfg["reward_config"] = dict(
        scales=dict(tracking_lin_vel=1.0, tracking_ang_vel=0.5, lin_vel_z=-2.0, ang_vel_xy=-0.05,
                    orientation=-0.2, dof_pos_limits=-1.0, pose=-0.2, termination=-1.0,
                    stand_still=-0.5, torques=-0.0002, action_rate=-0.005, energy=-0.0005,
                    feet_clearance=-1.0, feet_height=-0.0, feet_slip=-0.1, feet_air_time=0.1,
                    feet_phase=0.0, feet_swing=0.0, body_height=-0.0, contact=0.0, center=-0.0),
        tracking_sigma=0.25, swing_height=-0.2, base_feet_distance=-0.3, phase_sigma=0.05)
    cfg["method"] = "baseline"
    return cfg


def training_config(method: str = "pgtt") -> Dict[str, Any]:
    """default_config() / baseline_config() with the overrides of training/train.py:120-129."""
    cfg = default_config() if method == "pgtt" else baseline_config()
    cfg["command_config"]["u_max"] = [0.6, 0.6, 1.0]
    cfg["command_config"]["u_min"] = [-0.6, -0.6, -1.0]
    cfg["gait_freq"] = [1, 3]
    return cfg


def evaluation_config(method: str = "pgtt") -> Dict[str, Any]:
    """the evaluator's env of training/evaluate.py:120-129: default_config() / baseline_config() with the NARROWER command range
    u_max = [0.4, 0.4, 0.7], u_min = -u_max and gait_freq = [1, 3] (survivor counts are quoted on these commands, not on training's +-0.6 / 1.0)."""
    cfg = default_config() if method == "pgtt" else baseline_config()
    cfg["command_config"]["u_max"] = [0.4, 0.4, 0.7]
    cfg["command_config"]["u_min"] = [-0.4, -0.4, -0.7]
    cfg["gait_freq"] = [1, 3]
    return cfg


def with_overrides(cfg: Dict[str, Any], **kw) -> Dict[str, Any]:
    out = copy.deepcopy(cfg)
    for k, v in kw.items():
        node = out
        parts = k.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return out
