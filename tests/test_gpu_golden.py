"""HIP kernels against the reference's OWN outputs, no oracle in between: the fixtures tests/golden/task_step*.npz and
task_reset.npz were produced by executing go2/joystick_pgtt.py / go2/joystick.py (Joystick.step :141-231, Joystick.reset :50-131)
with jax.random stubbed to fixed draws and fake physics outputs (tools/gen_golden.py).  They are replayed through the C ABI
with libpgtt's test hooks (pgtt_set_test_overrides: fixed uniform draws, scan heights preset in buf.scan_z)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from phase_guided_terrain_traversal_amd import abi, configs, mjcf

TOL = 2e-4      # buffers are fp32; the reference ran in float64 numpy


def _load_step_cases(g, env, method):
    n = g.ncases
    S = np.zeros((abi.NSTATE, n), np.float32); I = np.zeros((abi.NISTATE, n), np.int32)
    F = np.zeros((abi.NFRAME, n), np.float32); Z = np.zeros((n, abi.NSCAN), np.float32); A = np.zeros((n, 12), np.float32)
    key = np.asarray(env.model["key_qpos"], dtype=np.float64)
    for i in range(n):
        k = lambda name: g[f"c{i}_{name}"]
        S[abi.S_QPOS:abi.S_QPOS + 19, i] = k("qpos"); S[abi.S_QVEL:abi.S_QVEL + 18, i] = k("qvel")
        S[abi.S_CMD:abi.S_CMD + 3, i] = k("in_command"); S[abi.S_PHASE:abi.S_PHASE + 4, i] = k("in_phase")
        S[abi.S_PHASE_DT, i] = k("in_phase_dt"); S[abi.S_GAIT_FREQ, i] = k("in_gait_freq")
        S[abi.S_LAST_ACT:abi.S_LAST_ACT + 12, i] = k("in_last_act"); S[abi.S_LAST_LAST_ACT:abi.S_LAST_LAST_ACT + 12, i] = k("in_last_last_act")
        S[abi.S_AIR_TIME:abi.S_AIR_TIME + 4, i] = k("in_feet_air_time"); S[abi.S_SWING_PEAK:abi.S_SWING_PEAK + 4, i] = k("in_swing_peak")
        S[abi.S_HMAX:abi.S_HMAX + 4, i] = k("in_H_max"); S[abi.S_HMIN:abi.S_HMIN + 4, i] = k("in_H_min")
        # info["motor_targets"] of THIS step (joystick_pgtt.py:145,149): what the physics kernel leaves in the row
        S[abi.S_MOTOR_TARGETS:abi.S_MOTOR_TARGETS + 12, i] = key[7:] + k("action") * env.config["action_scale"]
        S[abi.S_QERR_HIST:abi.S_QERR_HIST + 24, i] = k("in_qpos_error_history"); S[abi.S_QVEL_HIST:abi.S_QVEL_HIST + 24, i] = k("in_qvel_history")
        S[abi.S_LAST_CONTACT:abi.S_LAST_CONTACT + 4, i] = k("in_last_contact")
        I[abi.I_STEP, i] = int(k("in_step")); I[abi.I_STEPS_UNTIL_CMD, i] = int(k("in_steps_until_next_cmd"))
        s = k("sensordata")          # sensor layout of go2_mjx_feetonly.xml:258-274
        F[abi.F_GYRO:abi.F_GYRO + 3, i] = s[0:3]; F[abi.F_ACCEL:abi.F_ACCEL + 3, i] = s[3:6]
        F[abi.F_GLOBAL_LINVEL:abi.F_GLOBAL_LINVEL + 3, i] = s[13:16]; F[abi.F_GLOBAL_ANGVEL:abi.F_GLOBAL_ANGVEL + 3, i] = s[16:19]
        F[abi.F_LOCAL_LINVEL:abi.F_LOCAL_LINVEL + 3, i] = s[19:22]; F[abi.F_UPVECTOR:abi.F_UPVECTOR + 3, i] = s[22:25]
        F[abi.F_GRAVITY:abi.F_GRAVITY + 3, i] = -k("site_imu_mat")[2]             # imu_xmat^T (0, 0, -1), go2/base.py:129-131
        F[abi.F_FEET_POS:abi.F_FEET_POS + 12, i] = s[25:37]; F[abi.F_FEET_VEL:abi.F_FEET_VEL + 12, i] = s[37:49]
        F[abi.F_ACT_FORCE:abi.F_ACT_FORCE + 12, i] = k("actuator_force"); F[abi.F_CONTACT:abi.F_CONTACT + 4, i] = k("contact")
        F[abi.F_FOOT_SITE_Z:abi.F_FOOT_SITE_Z + 4, i] = k("site_foot_z")
        Z[i] = k("scan_z"); A[i] = k("action")
    return S, I, F, Z, A


_STEP_FILES = {"synthetic": "task_step", "rollout": "task_step_rollout", "draws": "task_step_draws"}


@pytest.mark.parametrize("source", ["synthetic", "rollout", "draws"])
@pytest.mark.parametrize("observe", ["fused", "split"])
@pytest.mark.parametrize("method", ["pgtt", "baseline"])
def test_observe_kernel_against_reference_step(golden_dir, method, observe, source):
    """observe_kernel (scan statistics, observation rows, 21 rewards, bookkeeping) on the reference's own Joystick.step vectors: 14 synthetic states per task
    and the closed-loop roll-out records (240 / 120 consecutive steps; tools/gen_golden.py).  `draws` (round 6): the same two kinds recorded with every
    uniform draw pinned to 0.2 / 0.4 / 0.7 - noise terms (2u - 1) * scale != 0 in the five noisy blocks (joystick_pgtt.py:242-285, configs.py:19-29) and
    sample_command's w = 1 branch with z = [1,1,1] / [1,0,1] (joystick_pgtt.py:603-611); one launch per pinned value, each case compared under its own"""
    from conftest import GoldenCases
    from phase_guided_terrain_traversal_amd.env import Joystick
    g = GoldenCases(os.path.join(golden_dir, _STEP_FILES[source] + ("" if method == "pgtt" else "_baseline") + ".npz"))
    n = g.ncases
    env = Joystick("flat_terrain", configs.training_config(method), num_envs=n, device="cuda:0", observe_form=observe, test_hooks=True)
    env.reset(seed=0)          # allocates / initialises everything; the rows the step reads are then overwritten
    S, I, F, Z, A = _load_step_cases(g, env, method)
    fracs = np.array([float(g[f"c{i}_frac"]) for i in range(n)])
    od, pd = abi.obs_dims(method)
    npos = nchanged = nnoisy = 0
    for f in np.unique(fracs):
        env.buffers["state"].copy_(torch.from_numpy(S)); env.buffers["istate"].copy_(torch.from_numpy(I))
        env.buffers["frame"].copy_(torch.from_numpy(F)); env.buffers["scan_z"].copy_(torch.from_numpy(Z))
        env.set_test_overrides(rng_value=float(f), scan_preset=True)
        env.observe(torch.from_numpy(A).cuda())
        torch.cuda.synchronize()
        b = {k: v.cpu().numpy().astype(np.float64) for k, v in env.buffers.items()}
        assert b["obs_state"].shape == (n, od) and b["obs_priv"].shape == (n, pd)
        for i in np.nonzero(fracs == f)[0]:
            k = lambda name: g[f"c{i}_{name}"]
            St, It = b["state"][:, i], b["istate"][:, i]
            assert np.abs(b["obs_state"][i] - k("obs")).max() < TOL, i
            assert np.abs(b["obs_priv"][i] - k("priv")).max() < TOL, i
            assert abs(b["reward"][i] - k("reward")) < TOL, i
            npos += k("reward") > 0
            assert b["done"][i] == k("done"), i
            assert np.abs(b["metrics"][:, i] - k("metrics")).max() < TOL * max(1.0, np.abs(k("metrics")).max()), i
            for off, cnt, name in ((abi.S_CMD, 3, "command"), (abi.S_PHASE, 4, "phase"), (abi.S_LAST_ACT, 12, "last_act"),
                                   (abi.S_LAST_LAST_ACT, 12, "last_last_act"), (abi.S_AIR_TIME, 4, "feet_air_time"), (abi.S_SWING_PEAK, 4, "swing_peak"),
                                   (abi.S_HMAX, 4, "H_max"), (abi.S_HMIN, 4, "H_min"), (abi.S_MOTOR_TARGETS, 12, "motor_targets"),
                                   (abi.S_QERR_HIST, 24, "qpos_error_history"), (abi.S_QVEL_HIST, 24, "qvel_history"), (abi.S_LAST_CONTACT, 4, "last_contact")):
                assert np.abs(St[off:off + cnt] - k("out_" + name)).max() < TOL, (i, name)
            assert It[abi.I_STEP] == int(k("out_step")) and It[abi.I_STEPS_UNTIL_CMD] == int(k("out_steps_until_next_cmd")), i
            # what the case exercises, counted on the KERNEL's outputs: a command that changed, a gyro row that carries noise
            nchanged += bool(np.abs(St[abi.S_CMD:abi.S_CMD + 3] - S[abi.S_CMD:abi.S_CMD + 3, i]).max() > 1e-3)
            nnoisy += bool(np.abs(b["obs_state"][i, :3] - F[abi.F_GYRO:abi.F_GYRO + 3, i]).min() > 0.03)       # |2u - 1| * 0.2 = 0.12 / 0.04 / 0.08
    assert npos >= 2           # the fixtures include un-clipped positive totals
    if source == "draws":      # the point of these records: the stochastic branches are taken, by the kernel, with the reference's numbers
        assert nchanged >= 20 and nnoisy == n, (nchanged, nnoisy)
    else:
        assert nchanged == 0 and nnoisy == 0
    env.close()


def test_reset_kernels_against_reference_reset(golden_dir):
    """pgtt_reset (reset_pose -> forward -> scan lift -> forward -> observe<RESET>) on the reference's own Joystick.reset vectors:
    spawn offset, yaw and quaternion product order (tilted keyframe), initial velocity, lift by the highest scan point on a
    0.12 m slab, command / gait-frequency / exponential-timer draws, info initial values, first observation"""
    from phase_guided_terrain_traversal_amd.env import Joystick
    g = np.load(os.path.join(golden_dir, "task_reset.npz"))
    slab = np.zeros((1, 100, 10), dtype=np.float32)
    slab[0, :, 3] = 1.0
    slab[0, 0, :3] = [0.0, 0.0, 0.06]; slab[0, 0, 7:] = [6.0, 6.0, 0.06]
    slab[0, 1:, :3] = [50.0, 50.0, -1.0]; slab[0, 1:, 7:] = 0.01
    for i in range(int(g["ncases"])):
        k = lambda name: g[f"r{i}_{name}"]
        method, f, top = str(k("method")), float(k("frac")), float(k("top"))
        task = "stairs" if top else "flat_terrain"
        model = dict(mjcf.load_model(task)); model["key_qpos"] = np.asarray(k("init_q"), dtype=np.float64)
        env = Joystick(task, configs.training_config(method), num_envs=3, terrain=slab if top else None, device="cuda:0", model=model, test_hooks=True)
        env.set_test_overrides(rng_value=f)
        env.reset(seed=3)
        torch.cuda.synchronize()
        S = env.buffers["state"].cpu().numpy().astype(np.float64); I = env.buffers["istate"].cpu().numpy()
        for e in range(3):
            assert np.abs(S[abi.S_QPOS:abi.S_QPOS + 19, e] - k("qpos")).max() < 3e-6, (i, e)
            assert np.abs(S[abi.S_QVEL:abi.S_QVEL + 18, e] - k("qvel")).max() < 3e-6, (i, e)
            assert np.abs(S[abi.S_CMD:abi.S_CMD + 3, e] - k("info_command")).max() < 3e-6, i
            assert abs(S[abi.S_GAIT_FREQ, e] - k("info_gait_freq")) < 3e-6 and abs(S[abi.S_PHASE_DT, e] - k("info_phase_dt")) < 3e-6
            assert np.abs(S[abi.S_PHASE:abi.S_PHASE + 4, e] - k("info_phase")).max() < 1e-6
            assert I[abi.I_STEPS_UNTIL_CMD, e] == int(k("info_steps_until_next_cmd")) and I[abi.I_STEP, e] == 0
            for off, cnt, name in ((abi.S_LAST_ACT, 12, "last_act"), (abi.S_LAST_LAST_ACT, 12, "last_last_act"), (abi.S_AIR_TIME, 4, "feet_air_time"),
                                   (abi.S_SWING_PEAK, 4, "swing_peak"), (abi.S_HMAX, 4, "H_max"), (abi.S_HMIN, 4, "H_min"), (abi.S_MOTOR_TARGETS, 12, "motor_targets"),
                                   (abi.S_QERR_HIST, 24, "qpos_error_history"), (abi.S_QVEL_HIST, 24, "qvel_history"), (abi.S_LAST_CONTACT, 4, "last_contact")):
                assert np.abs(S[off:off + cnt, e] - k("info_" + name)).max() < 1e-6, (i, name)
        assert float(env.buffers["reward"].abs().sum()) == 0.0 and float(env.buffers["done"].abs().sum()) == 0.0
        env.close()
