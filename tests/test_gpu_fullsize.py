"""Full-size (N = 4096, BASELINE configs[1..3]) size-independent properties of the HIP path: determinism, shard
invariance, masked reset, Episode/AutoReset semantics, analytic height scan, finiteness along a rollout."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from phase_guided_terrain_traversal_amd import abi, configs, mjcf

ASSETS = os.path.join(os.path.dirname(mjcf.__file__), "assets", "terrains")
N = 4096


def make(n=N, off=0, level="level4", autoreset=True, cfg=None, variant=None, **kw):
    from phase_guided_terrain_traversal_amd.env import Joystick
    terrain = np.load(os.path.join(ASSETS, level + ".npy"))
    if variant is None:
        variant = np.random.Generator(np.random.Philox(key=[2, 0])).integers(0, terrain.shape[0], N).astype(np.int32)[off:off + n]
    return Joystick("stairs", cfg or configs.training_config(), num_envs=n, terrain=terrain, device="cuda:0",
                    autoreset=autoreset, env_id_offset=off, variant=torch.from_numpy(variant), **kw), terrain, variant


def actions(k, n=N, off=0):
    g = np.random.Generator(np.random.Philox(key=[7, k]))
    return torch.from_numpy(np.tanh(g.normal(size=(N, 12)) * 0.6).astype(np.float32)[off:off + n]).cuda()


def snapshot(env):
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in env.buffers.items()}


def test_determinism_and_shard_invariance():
    a, _, _ = make()
    b, _, _ = make()
    h0, _, _ = make(n=N // 2, off=0)
    h1, _, _ = make(n=N // 2, off=N // 2)
    for e in (a, b, h0, h1):
        e.reset(seed=11)
    for k in range(12):
        a.step(actions(k)); b.step(actions(k))
        h0.step(actions(k, N // 2, 0)); h1.step(actions(k, N // 2, N // 2))
    sa, sb, s0, s1 = snapshot(a), snapshot(b), snapshot(h0), snapshot(h1)
    for key in ("state", "istate", "obs_state", "obs_priv", "reward", "done", "metrics", "scan_z", "frame"):
        assert torch.equal(sa[key], sb[key]), key                               # bitwise run-to-run determinism
        if sa[key].dim() == 2 and sa[key].shape[0] == N:                         # AoS [N][k]
            merged = torch.cat([s0[key], s1[key]], 0)
        elif sa[key].dim() == 1:
            merged = torch.cat([s0[key], s1[key]], 0)
        else:                                                                    # SoA [rows][N]
            merged = torch.cat([s0[key], s1[key]], 1)
        assert torch.equal(sa[key], merged), key                                 # 2 shards == 1 batch, bit for bit
    for e in (a, b, h0, h1):
        e.close()


def test_step_equals_physics_then_observe_and_sees_caller_edits():
    """pgtt_step hands the physics outputs to its observe launch through the library's env-major hand-over record; pgtt_physics +
    pgtt_observe called on their own read the caller-visible rows.  Same bits either way - and a row the caller edits between the two calls is
    what the observation then shows (the record must not shadow it)."""
    n = 1000                                                 # ragged: the last physics wave is partly empty
    a_env, _, _ = make(n=n, level="level13", interval_sums=True)
    b_env, _, _ = make(n=n, level="level13", interval_sums=True)
    a_env.reset(seed=5); b_env.reset(seed=5)
    for k in range(30):
        act = actions(k, n)
        a_env.step(act)
        b_env.physics(act); b_env.observe(act)
    torch.cuda.synchronize()
    sa, sb = snapshot(a_env), snapshot(b_env)
    for key in sa:
        assert torch.equal(sa[key].contiguous().view(torch.uint8), sb[key].contiguous().view(torch.uint8)), key
    act = actions(30, n)
    b_env.physics(act)
    b_env.buffers["frame"][0].fill_(0.75)                    # PGTT_F_GYRO + 0 = observation row 0 (noise: +- noise_level * scale, 0.2 at most)
    b_env.observe(act)
    torch.cuda.synchronize()
    live = b_env.buffers["done"] == 0                        # a finished episode shows its first observation instead
    obs0 = b_env.buffers["obs_state"][:, 0][live]
    assert int(live.sum()) > n // 2 and float((obs0 - 0.75).abs().max()) < 0.25
    a_env.close(); b_env.close()


def test_masked_reset_leaves_other_envs_untouched():
    env, _, _ = make()
    env.reset(seed=3)
    for k in range(5):
        env.step(actions(k))
    before = snapshot(env)
    mask = torch.zeros(N, dtype=torch.uint8); mask[::3] = 1
    env.reset(seed=4, mask=mask)
    after = snapshot(env)
    keep = (mask == 0).cuda()
    for key in ("state", "istate"):
        assert torch.equal(before[key][:, keep], after[key][:, keep]), key
    assert torch.equal(before["obs_state"][keep], after["obs_state"][keep])
    m = mask.bool().cuda()
    assert (after["istate"][abi.I_STEP][m] == 0).all() and (after["done"][m] == 0).all()
    assert not torch.equal(before["state"][:3, m], after["state"][:3, m])
    env.close()


@pytest.mark.parametrize("observe", ["fused", "split"])
def test_episode_and_autoreset_semantics(observe):
    cfg = configs.with_overrides(configs.training_config(), episode_length=7)
    env, _, _ = make(n=512, cfg=cfg, variant=np.zeros(512, dtype=np.int32), observe_form=observe)
    env.reset(seed=5)
    first_state = env.buffers["first_state"].clone(); first_obs = env.buffers["first_obs"].clone()
    assert torch.equal(first_state, env.buffers["state"][:abi.S_CMD])
    lengths = []
    for k in range(15):
        obs, reward, done, info = env.step(actions(k, 512))
        torch.cuda.synchronize()
        d = done.bool()
        ep = env.buffers["istate"][abi.I_EP_STEPS]
        if k == 6:                                                               # truncation at episode_length
            assert d.all() and (ep == 7).all()
        if d.any():                                                              # AutoReset-to-first-state
            assert torch.equal(env.buffers["state"][:abi.S_CMD][:, d], first_state[:, d])
            assert torch.equal(obs["state"][d], first_obs[d][:, :abi.OBS])
            assert torch.equal(obs["privileged_state"][d], first_obs[d][:, abi.OBS:])
        if k == 7:                                                               # counters restart after a done
            assert (ep == 1).all()
            # the step after a done contributes nothing to the running episode sums (x += v; x *= 1 - prev_done)
            assert (env.buffers["ep_metrics"][abi.NMETRIC + 1] == 0).all()
        if k == 9:
            assert (env.buffers["ep_metrics"][abi.NMETRIC + 1] == 2).all()
        lengths.append(int(ep.max()))
    # task info (phase clock, step counter) is NOT reset by AutoReset
    assert (env.buffers["istate"][abi.I_STEP] == 15).all()
    env.close()


@pytest.mark.parametrize("observe", ["fused", "split"])
def test_wrapper_contract_row_by_row_on_a_hand_built_trajectory(observe, golden_dir):
    """The reference states its wrapper contract in two places only: `episode_length` + `wrap_for_brax_training` (training/train.py:255,262) and the
    evaluator keys training/evaluate.py:203-215 reads (`eval/episode_reward`, `eval/episode_reward/tracking_*`, `eval/avg_episode_length` = means over the
    envs of the Episode wrapper's running sums at the end of each env's first episode).  SURVEY 8b (ii) / (iii) [UPSTREAM-RECALL] spell the semantics
    out.  Here a trajectory with KNOWN episode boundaries goes through pgtt_step with episode_length = 7: envs 0..15 are turned on their backs before
    control step 2 (termination there), envs 16..31 before control step 6 (termination AND the length limit in the same step), the rest run into the
    limit at step 6.  A twin handle without the wrappers is put on the wrapped env's state before every step, so that it shows what the bare
    `Joystick.step` returns from the same state.  Asserted row by row: `done`, the truncation flag (1 - termination at the limit), the running episode
    sums against a host restatement of `x = (x + v) * (1 - prev_done)`, the restore of the first reset's physics rows and observations, EVERY task
    `info` row surviving the restore, and evaluate.py's aggregation of the first episodes."""
    n, L = 64, 7
    cfg = configs.with_overrides(configs.training_config(), episode_length=L)
    A, _, var = make(n=n, cfg=cfg, variant=np.zeros(n, dtype=np.int32), observe_form=observe)
    B, _, _ = make(n=n, cfg=cfg, variant=np.zeros(n, dtype=np.int32), observe_form=observe, autoreset=False)
    # the metric rows carry the reference's names (reference-held: the key set of Joystick.reset's metrics dict, tests/golden/task_reset.npz)
    ref_keys = [str(k) for k in np.load(os.path.join(golden_dir, "task_reset.npz"), allow_pickle=False)["r0_metrics_keys"]]
    assert sorted(["reward/" + k for k in abi.REWARD_KEYS] + ["swing_peak"]) == ref_keys and abi.NMETRIC == len(ref_keys)
    i_lin, i_ang = abi.REWARD_KEYS.index("tracking_lin_vel"), abi.REWARD_KEYS.index("tracking_ang_vel")     # what evaluate.py:213-214 reads
    A.reset(seed=5); B.reset(seed=5)
    torch.cuda.synchronize()
    first_state, first_obs = A.buffers["first_state"].clone(), A.buffers["first_obs"].clone()
    E = np.zeros((abi.NMETRIC + 2, n), np.float32)                         # host restatement of the Episode wrapper's sums
    prev_done = np.zeros(n, bool)
    first = np.ones(n, bool); ev_ret = np.zeros(n, np.float32); ev_len = np.zeros(n, np.float32); ev_terms = np.zeros((abi.NMETRIC, n), np.float32)
    ep_at_first_done = np.zeros((abi.NMETRIC + 2, n), np.float32)
    ep_steps = np.zeros(n, np.int64)
    expect_done = {2: set(range(0, 16)), 6: set(range(16, n)), 9: set(range(0, 16)), 13: set(range(16, n))}
    for k in range(15):
        if k in (2, 6):                                                        # hand-built boundary: these robots are on their backs now
            idx = torch.arange(0, 16) if k == 2 else torch.arange(16, 32)
            A.buffers["state"][abi.S_QPOS + 3:abi.S_QPOS + 7, idx] = torch.tensor([0.0, 1.0, 0.0, 0.0], device="cuda:0")[:, None]
        for key in ("state", "istate", "scan_z"):                              # the twin steps from the SAME state, without Episode / AutoReset
            B.buffers[key].copy_(A.buffers[key])
        a = actions(k, n)
        obs, reward, done, info = A.step(a)
        obsB, rewardB, doneB, infoB = B.step(a)
        torch.cuda.synchronize()
        gA = {kk: v.cpu().numpy() for kk, v in A.buffers.items()}; gB = {kk: v.cpu().numpy() for kk, v in B.buffers.items()}
        term = gB["done"] != 0                                                  # the bare env's done = termination
        ep_steps = np.where(prev_done, 0, ep_steps) + 1
        d = term | (ep_steps >= L)
        assert np.array_equal(gA["done"] != 0, d), k                            # EpisodeWrapper: done |= steps >= episode_length
        assert set(np.nonzero(d)[0]) == expect_done.get(k, set()), (k, np.nonzero(d)[0])
        assert np.array_equal(gA["istate"][abi.I_EP_STEPS], ep_steps), k
        trunc = (ep_steps >= L) & ~term                                         # truncation = 1 - termination at the limit
        if k == 2:
            assert term[:16].all() and not trunc.any()
        if k == 6:
            assert term[16:32].all() and not trunc[16:32].any() and trunc[32:].all() and (gA["frame"][abi.F_UPVECTOR + 2, 16:32] < 0).all()
        # reward / metrics of the step are the bare env's, bit for bit
        assert np.array_equal(gA["reward"], gB["reward"]) and np.array_equal(gA["metrics"], gB["metrics"]), k
        # running episode sums: x = (x + v) * (1 - prev_done), rows = 22 metrics, sum_reward, length
        keep = np.where(prev_done, np.float32(0), np.float32(1))
        E[:abi.NMETRIC] = (E[:abi.NMETRIC] + gA["metrics"]) * keep
        E[abi.NMETRIC] = (E[abi.NMETRIC] + gA["reward"]) * keep
        E[abi.NMETRIC + 1] = (E[abi.NMETRIC + 1] + np.float32(1)) * keep
        for r in range(abi.NMETRIC + 2):
            assert np.array_equal(gA["ep_metrics"][r], E[r]), (k, r)
        # AutoReset: physics rows and observations of a done env are the FIRST reset's, the others the bare env's
        for r in range(abi.S_CMD):
            assert np.array_equal(gA["state"][r], np.where(d, first_state[r].cpu().numpy(), gB["state"][r])), (k, r)
        fo = first_obs.cpu().numpy()
        assert np.array_equal(gA["obs_state"], np.where(d[:, None], fo[:, :abi.OBS], gB["obs_state"])), k
        assert np.array_equal(gA["obs_priv"], np.where(d[:, None], fo[:, abi.OBS:], gB["obs_priv"])), k
        # ... while EVERY task info row survives the restore: command, phase, gait frequency, last actions, air time, swing peak, H_max / H_min,
        # motor targets, both histories, last contact - and the task's own counters
        for r in range(abi.S_CMD, abi.NSTATE):
            assert np.array_equal(gA["state"][r], gB["state"][r]), (k, r)
        for r in (abi.I_STEP, abi.I_STEPS_UNTIL_CMD, abi.I_RNG_CTR):
            assert np.array_equal(gA["istate"][r], gB["istate"][r]), (k, r)
        # the evaluator's view (training/evaluate.py:203-215 through brax's Evaluator): sums over each env's FIRST episode
        w = first.astype(np.float32)
        ev_ret += gA["reward"] * w; ev_len += w; ev_terms += gA["metrics"] * w
        newly = first & d
        ep_at_first_done[:, newly] = gA["ep_metrics"][:, newly]
        first &= ~d
        prev_done = d
    assert not first.any()
    # at the step an episode ends the wrapper's sums ARE the evaluator's: eval/episode_reward, eval/episode_reward/<term>, eval/avg_episode_length
    assert np.array_equal(ep_at_first_done[abi.NMETRIC], ev_ret) and np.array_equal(ep_at_first_done[abi.NMETRIC + 1], ev_len)
    for r in (i_lin, i_ang):
        assert np.array_equal(ep_at_first_done[r], ev_terms[r])
    assert ev_len.mean() == (16 * 3 + 48 * 7) / 64.0 and (ev_len[:16] == 3).all() and (ev_len[16:] == 7).all()
    assert (A.buffers["istate"][abi.I_STEP] == 15).all()                       # the task's step counter never restarts
    A.close(); B.close()


@pytest.mark.parametrize("method", ["pgtt", "baseline"])
def test_task_layer_is_the_oracles_on_the_devices_own_physics_every_env_step(method):
    """A statement about the task layer that needs no W: along a rollout on level4 with observation noise on, the DEVICE's physics outputs of EVERY
    env-step (qpos, qvel, the 65-row sensor frame, contact flags, the 117 scan heights) go through the oracle's task layer (`pgtt_oracle_task_post_ex`: the
    code path tests/test_golden_task.py holds to the reference's own vectors) with the env's Philox key, from the device's own `info` rows before the
    step - and observations (171 + 215; noise draws included), reward, done, the 22 metrics, every `info` row after the step (command resampling, phase,
    air time, swing peak, H_max / H_min, histories, last contact, the counters) must agree to fp32 rounding.  No solver sits in between, so nothing is
    amplified: the ~22 % of env-steps outside W are covered like the rest."""
    import ctypes as C
    from oracle import oracle
    from test_golden_task import PostIn
    n, steps, seed = 256, 60, 11
    cfg = configs.training_config(method)
    env, terrain, variant = make(n=n, cfg=cfg, autoreset=False)
    cs, ms = abi.config_struct(dict(env.config)), abi.model_struct(env.model)
    od, pd = abi.obs_dims(method)
    L = oracle.lib()
    L.pgtt_oracle_task_post_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint32]
    hb = oracle.HostBuffers(1, method=method)
    env.reset(seed)
    worst = {}
    resampled = 0
    for k in range(steps):
        torch.cuda.synchronize()
        S0, I0 = env.buffers["state"].cpu().numpy(), env.buffers["istate"].cpu().numpy()
        a = actions(k, n)
        env.step(a)
        torch.cuda.synchronize()
        g = {kk: v.cpu().numpy() for kk, v in env.buffers.items()}
        act = a.cpu().numpy()
        Fr = g["frame"]
        for e in range(n):
            hb["state"][:, 0] = S0[:, e]; hb["istate"][:, 0] = I0[:, e]
            pin = PostIn()
            q = g["state"][:19, e].astype(np.float64)
            np.ctypeslib.as_array(pin.qpos)[:] = q
            np.ctypeslib.as_array(pin.qvel)[:] = g["state"][19:37, e]
            sd = np.zeros(49)
            sd[0:3] = Fr[abi.F_GYRO:abi.F_GYRO + 3, e]; sd[3:6] = Fr[abi.F_ACCEL:abi.F_ACCEL + 3, e]; sd[6:10] = q[3:7]
            sd[13:16] = Fr[abi.F_GLOBAL_LINVEL:abi.F_GLOBAL_LINVEL + 3, e]; sd[16:19] = Fr[abi.F_GLOBAL_ANGVEL:abi.F_GLOBAL_ANGVEL + 3, e]
            sd[19:22] = Fr[abi.F_LOCAL_LINVEL:abi.F_LOCAL_LINVEL + 3, e]; sd[22:25] = Fr[abi.F_UPVECTOR:abi.F_UPVECTOR + 3, e]
            sd[25:37] = Fr[abi.F_FEET_POS:abi.F_FEET_POS + 12, e]; sd[37:49] = Fr[abi.F_FEET_VEL:abi.F_FEET_VEL + 12, e]
            np.ctypeslib.as_array(pin.sensordata)[:] = sd
            mat = np.zeros(9); mat[6:9] = -Fr[abi.F_GRAVITY:abi.F_GRAVITY + 3, e]          # the task layer reads the IMU frame's third row only (gravity)
            np.ctypeslib.as_array(pin.site_imu_mat)[:] = mat
            np.ctypeslib.as_array(pin.site_foot_z)[:] = Fr[abi.F_FOOT_SITE_Z:abi.F_FOOT_SITE_Z + 4, e]
            np.ctypeslib.as_array(pin.actuator_force)[:] = Fr[abi.F_ACT_FORCE:abi.F_ACT_FORCE + 12, e]
            np.ctypeslib.as_array(pin.action)[:] = act[e]
            np.ctypeslib.as_array(pin.scan_z)[:] = g["scan_z"][e]
            np.ctypeslib.as_array(pin.contact)[:] = Fr[abi.F_CONTACT:abi.F_CONTACT + 4, e].astype(np.int32)
            b = hb.struct()
            L.pgtt_oracle_task_post_ex(C.byref(cs), C.byref(ms), C.byref(b), C.byref(pin), 0, C.c_uint64(seed), C.c_uint32(e))
            errs = dict(obs=np.abs(hb["obs_state"][0] - g["obs_state"][e]).max(), priv=np.abs(hb["obs_priv"][0] - g["obs_priv"][e]).max(),
                        reward=abs(float(hb["reward"][0]) - float(g["reward"][e])), metrics=(np.abs(hb["metrics"][:, 0] - g["metrics"][:, e]) / (1 + np.abs(hb["metrics"][:, 0]))).max(),
                        info=np.abs(hb["state"][abi.S_CMD:, 0] - g["state"][abi.S_CMD:, e]).max())
            for kk, v in errs.items():
                worst[kk] = max(worst.get(kk, 0.0), float(v))
            assert errs["obs"] < 1e-5 and errs["priv"] < 2e-5 and errs["reward"] < 1e-6 and errs["metrics"] < 1e-5 and errs["info"] < 1e-6, (k, e, errs)      # measured: 1.9e-6 / 1.9e-6 / 4e-9 / 1e-7 / 6e-8
            assert hb["done"][0] == g["done"][e] and np.array_equal(hb["istate"][:3, 0], g["istate"][:3, e]), (k, e)
            resampled += int(not np.array_equal(S0[abi.S_CMD:abi.S_CMD + 3, e], g["state"][abi.S_CMD:abi.S_CMD + 3, e]))
    assert resampled >= 10                                      # sample_command's resampling branch was met (timer ~ Exp(5 s))
    print(f"\n[{method}] {n * steps} env-steps, worst |device - oracle task layer on the device's physics|:", {kk: f"{v:.2e}" for kk, v in worst.items()}, "commands resampled:", resampled)
    assert od == g["obs_state"].shape[1] and pd == g["obs_priv"].shape[1]
    env.close()


def _tops(boxes, pts):
    """analytic terrain height under (x, y) for yaw-only boxes resting on z = 0"""
    top = np.zeros(len(pts))
    act = boxes[:, 0] < 50
    for b in boxes[act]:
        ang = 2 * np.arctan2(b[6], b[3])
        dx, dy = pts[:, 0] - b[0], pts[:, 1] - b[1]
        lx, ly = np.cos(ang) * dx + np.sin(ang) * dy, -np.sin(ang) * dx + np.cos(ang) * dy
        inside = (np.abs(lx) <= b[7]) & (np.abs(ly) <= b[8])
        top = np.where(inside, np.maximum(top, b[2] + b[9]), top)
    return top


def test_scan_equals_box_tops_full_size():
    env, terrain, variant = make(autoreset=False)
    env.reset(seed=9)
    for k in range(10):
        env.step(actions(k))
    torch.cuda.synchronize()
    S = env.buffers["state"].cpu().numpy(); z = env.buffers["scan_z"].cpu().numpy()
    q = S[3:7]; yaw = np.arctan2(2 * (q[0] * q[3] + q[1] * q[2]), 1 - 2 * (q[2] ** 2 + q[3] ** 2))
    r, c = np.meshgrid(np.arange(13), np.arange(9), indexing="ij")
    ox, oy = ((6 - r) * 0.1).ravel(), ((4 - c) * 0.1).ravel()
    bad = 0
    for e in range(0, N, 37):                                                    # 111 envs x 117 rays, all variants
        px = S[0, e] + ox * np.cos(yaw[e]) - oy * np.sin(yaw[e]); py = S[1, e] + ox * np.sin(yaw[e]) + oy * np.cos(yaw[e])
        tops = _tops(terrain[variant[e]], np.stack([px, py], 1))
        diff = np.abs(z[e] - tops)
        # rays within 1e-4 of a box edge may legitimately fall on either side
        edge = diff > 1e-4
        if edge.any():
            for dx, dy in ((2e-4, 0), (-2e-4, 0), (0, 2e-4), (0, -2e-4)):
                t2 = _tops(terrain[variant[e]], np.stack([px + dx, py + dy], 1))
                edge &= np.abs(z[e] - t2) > 1e-4
        bad += int(edge.sum())
    assert bad == 0
    env.close()


def test_rollout_stays_finite_and_bounded():
    env, _, _ = make(level="level13")
    env.reset(seed=1)
    for k in range(150):
        obs, reward, done, info = env.step(actions(k))
    torch.cuda.synchronize()
    for key in ("state", "obs_state", "obs_priv", "reward", "metrics", "frame"):
        assert torch.isfinite(env.buffers[key]).all(), key
    assert float(env.buffers["state"][2].max()) < 2.0 and float(env.buffers["state"][19:37].abs().max()) < 200.0
    assert (reward >= 0).all() and (reward <= 1e4).all()
    q = env.buffers["state"][3:7]
    assert torch.allclose((q * q).sum(0), torch.ones(N, device="cuda"), atol=1e-5)
    env.close()


def test_kernel_timing_api():
    """pgtt_enable_timing / pgtt_last_kernel_ms / pgtt_kernel_ms_mean: ring of HIP events, every n-th step"""
    env, _, _ = make(n=512)
    env.reset(seed=3)
    env.enable_timing(True)
    for k in range(70):                      # more steps than ring slots (64): slots are harvested on re-use
        env.step(actions(k, 512))
    p, o = env.last_kernel_ms()
    pm, om, cnt = env.kernel_ms_mean()
    assert cnt == 70 and 0.0 < pm < 50.0 and 0.0 < om < 50.0 and 0.0 < p < 50.0 and 0.0 < o < 50.0
    env.enable_timing(8)
    for k in range(33):
        env.step(actions(k, 512))
    pm, om, cnt = env.kernel_ms_mean()
    assert cnt == 4 and pm > 0.0                # steps 4, 12, 20, 28 (mid-period: never the first step after a synchronisation)
    env.enable_timing(False)
    env.step(actions(0, 512))
    with pytest.raises(Exception):
        env.kernel_ms_mean()
    env.close()


def test_scan_on_tilted_boxes_matches_the_oracle():
    """boxes turned about all three axes go through the generic six-face ray test of observe_kernel (upright boxes, i.e.
    every shipped terrain, take its two-face form); both against the oracle's restatement of mjx _ray_box"""
    from oracle import oracle
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(5)
    nb = 24
    boxes = np.zeros((1, 100, 10), np.float32)
    boxes[0, :, 0] = 1.0e4 + np.arange(100)          # unused placeholders far away, like the shipped files
    boxes[0, :, 3] = 1.0
    for b in range(nb):
        upright = b % 3 == 0
        eul = rng.uniform(-0.5, 0.5, 3) * ([0, 0, 1] if upright else [1, 1, 1])
        q = Rotation.from_euler("xyz", eul).as_quat()                      # x y z w
        boxes[0, b, :3] = [rng.uniform(-1.2, 1.2), rng.uniform(-1.2, 1.2), rng.uniform(0.0, 0.15)]
        boxes[0, b, 3:7] = [q[3], q[0], q[1], q[2]]
        boxes[0, b, 7:10] = [rng.uniform(0.1, 0.5), rng.uniform(0.1, 0.5), rng.uniform(0.02, 0.12)]
    n = 64
    from phase_guided_terrain_traversal_amd.env import Joystick
    cfg = configs.training_config()
    env = Joystick("stairs", cfg, num_envs=n, terrain=boxes, device="cuda:0", variant=torch.zeros(n, dtype=torch.int32))
    env.reset(seed=3)
    z = env.scan().cpu().numpy()
    S = env.buffers["state"].cpu().numpy()
    cs = abi.config_struct(env.config)
    worst = 0.0
    for e in range(n):
        q = S[3:7, e].astype(np.float64)
        yaw = np.arctan2(2 * (q[0] * q[3] + q[1] * q[2]), 1 - 2 * (q[2] ** 2 + q[3] ** 2))
        ref = oracle.scan(cs, boxes[0], S[:3, e].astype(np.float64), float(yaw))[..., 2].ravel()
        d = np.abs(z[e] - ref)
        # a ray within 1e-4 of a box edge may fall on either side in fp32: tolerated on at most a few rays
        assert (d > 2e-4).sum() <= 2, (e, d.max())
        worst = max(worst, float(np.median(d)))
    assert worst < 1e-5
    env.close()


def test_configs3_wfc_dr_8192_full_size():
    """BASELINE configs[3] at its full size: 8192 envs (4097..8192 envs run the oct lane layout: 8 envs per wave),
    WFC-generated terrain + full randomize.py DR.  Size-independent properties: bitwise run-to-run determinism, two
    4096-env shards == one batch (DR and RNG streams are keyed by the global env id), finiteness along a rollout with
    AutoReset, scan == analytic box tops of the env's own variant."""
    from phase_guided_terrain_traversal_amd.env import Joystick
    from phase_guided_terrain_traversal_amd.randomize import domain_randomize
    from phase_guided_terrain_traversal_amd.terrain_gen import create_random_matrix
    n = 8192
    terrain = create_random_matrix(100, 100, 5, 0.05, 0.13, seed=3)
    model = mjcf.load_model("stairs")

    def build(cnt, off, layout=None):
        out = domain_randomize(model, cnt, seed=3, terrain=terrain, env_id_offset=off, total_envs=n)
        return Joystick("stairs", configs.training_config(), num_envs=cnt, terrain=terrain, device="cuda:0", autoreset=True, env_id_offset=off, layout=layout,
                        variant=torch.from_numpy(out["variant"]), params=torch.from_numpy(out["params"]), box_friction=torch.from_numpy(out["box_friction"])), out

    def acts(k, cnt, off):
        g = np.random.Generator(np.random.Philox(key=[9, k]))
        return torch.from_numpy(np.tanh(g.normal(size=(n, 12)) * 0.6).astype(np.float32)[off:off + cnt]).cuda()

    h0, _ = build(n // 2, 0, "oct"); h1, _ = build(n // 2, n // 2, "oct")       # the shards pin the layout the full batch selects by itself (PgttConfig.lane_layout)
    a, dr = build(n, 0); b, _ = build(n, 0)
    for e in (a, b, h0, h1):
        e.reset(seed=5)
    for k in range(40):
        a.step(acts(k, n, 0)); b.step(acts(k, n, 0)); h0.step(acts(k, n // 2, 0)); h1.step(acts(k, n // 2, n // 2))
    sa, sb, s0, s1 = snapshot(a), snapshot(b), snapshot(h0), snapshot(h1)
    for key in ("state", "istate", "obs_state", "obs_priv", "reward", "done", "metrics", "scan_z", "frame"):
        assert torch.equal(sa[key], sb[key]), key
        merged = torch.cat([s0[key], s1[key]], 0 if (sa[key].dim() == 1 or sa[key].shape[0] == n) else 1)
        assert torch.equal(sa[key], merged), key
        assert torch.isfinite(sa[key].float()).all(), key
    assert float(sa["done"].sum()) >= 0 and int(sa["istate"][abi.I_STEP].min()) == 40
    # scan of the last step against the analytic tops of each env's own variant
    S = sa["state"].cpu().numpy(); z = sa["scan_z"].cpu().numpy(); variant = dr["variant"]
    q = S[3:7]; yaw = np.arctan2(2 * (q[0] * q[3] + q[1] * q[2]), 1 - 2 * (q[2] ** 2 + q[3] ** 2))
    r, c = np.meshgrid(np.arange(13), np.arange(9), indexing="ij")
    ox, oy = ((6 - r) * 0.1).ravel(), ((4 - c) * 0.1).ravel()
    bad = 0
    done_now = sa["done"].cpu().numpy() != 0          # a finished episode's qpos was replaced by the first state after the scan
    for e in range(0, n, 61):
        if done_now[e]:
            continue
        px = S[0, e] + ox * np.cos(yaw[e]) - oy * np.sin(yaw[e]); py = S[1, e] + ox * np.sin(yaw[e]) + oy * np.cos(yaw[e])
        edge = np.abs(z[e] - _tops(terrain[variant[e]], np.stack([px, py], 1))) > 1e-4
        if edge.any():
            for dx, dy in ((2e-4, 0), (-2e-4, 0), (0, 2e-4), (0, -2e-4)):
                edge &= np.abs(z[e] - _tops(terrain[variant[e]], np.stack([px + dx, py + dy], 1))) > 1e-4
        bad += int(edge.sum())
    assert bad == 0
    for e in (a, b, h0, h1):
        e.close()


def test_rccl_metric_allreduce_single_rank():
    """the one collective of the path through RCCL itself (backend "nccl" on ROCm): a one-rank process group on this box's GPU, the
    fused 25-float all-reduce issued from the bench loop's MetricReducer after real env steps (the 8-GPU run is the driver's)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys, json
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from phase_guided_terrain_traversal_amd import abi, configs
from phase_guided_terrain_traversal_amd.distributed import MetricReducer, init_from_env
from phase_guided_terrain_traversal_amd.env import Joystick
rank, local, world = init_from_env("nccl", force=True)
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
torch.cuda.set_device(local)
env = Joystick("flat_terrain", configs.training_config(), num_envs=256, device="cuda:0", autoreset=True)
env.reset(seed=1)
red = MetricReducer(torch.device("cuda", 0))
tot = 0.0
for k in range(20):
    obs, reward, done, info = env.step(torch.zeros(256, 12, device="cuda"))
    red.accumulate_block(env.step_block); tot += float(reward.sum())
out = red.reduce()
torch.cuda.synchronize()
print(json.dumps({"env_steps": float(out["env_steps"]), "reward_mean": float(out["reward_mean"]), "expect": tot / (20 * 256)}))
dist.destroy_process_group()
""" % root
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    import json
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["env_steps"] == 20 * 256 and abs(d["reward_mean"] - d["expect"]) < 1e-5


@pytest.mark.parametrize("observe", ["fused", "split"])
def test_interval_sums_equal_the_per_step_sums(observe):
    """buffers["interval_sums"]: the running per-env sums of [22 metrics; reward; done] kept by the step kernels are the sums of
    the per-step outputs (same additions in the same order: bitwise), AutoReset on, until the caller clears the block"""
    env, _, _ = make(n=1024, level="level13", observe_form=observe, interval_sums=True)
    env.reset(seed=2)
    ref = torch.zeros_like(env.buffers["interval_sums"])
    for k in range(25):
        env.step(actions(k, 1024))
        ref[:abi.NMETRIC] += env.buffers["metrics"]; ref[abi.NMETRIC] += env.buffers["reward"]; ref[abi.NMETRIC + 1] += env.buffers["done"]
    torch.cuda.synchronize()
    assert torch.equal(env.buffers["interval_sums"], ref) and float(ref[abi.NMETRIC + 1].sum()) > 0
    from phase_guided_terrain_traversal_amd.distributed import MetricReducer
    # the rank-local reduction as ONE launch of the library (pgtt_interval_reduce: sum over the envs, block cleared, accumulator added to)
    # against the torch form of the same reduction (GEMV + fill), on a copy of the block; a ragged env count takes the scalar loop
    want = ref.double().sum(1)
    acc = torch.full((abi.NMETRIC + 3,), 2.0, dtype=torch.float32, device="cuda")
    env.interval_reduce(acc, 7.0, accumulate=True)
    torch.cuda.synchronize()
    assert float(env.buffers["interval_sums"].abs().sum()) == 0.0 and float(acc[abi.NMETRIC + 2]) == 9.0
    assert torch.allclose(acc[:abi.NMETRIC + 2].double() - 2.0, want, rtol=2e-5, atol=1e-3)
    env.buffers["interval_sums"].copy_(ref)
    env.interval_reduce(acc, 5.0)                                  # overwrite form
    assert float(acc[abi.NMETRIC + 2]) == 5.0 and torch.allclose(acc[:abi.NMETRIC + 2].double(), want, rtol=2e-5, atol=1e-3)
    env.buffers["interval_sums"].copy_(ref)
    out = MetricReducer(torch.device("cuda", 0)).reduce_block(env.buffers["interval_sums"], 25.0 * 1024)
    assert float(env.buffers["interval_sums"].abs().sum()) == 0.0 and float(out["env_steps"]) == 25 * 1024
    assert abs(float(out["reward_mean"]) - float(ref[abi.NMETRIC].sum()) / (25 * 1024)) < 1e-6
    env.buffers["interval_sums"].copy_(ref)
    out2 = MetricReducer(torch.device("cuda", 0)).reduce_env(env, 25.0 * 1024)
    assert float(env.buffers["interval_sums"].abs().sum()) == 0.0 and float(out2["env_steps"]) == 25 * 1024
    assert torch.allclose(out2["metrics_mean"], out["metrics_mean"], rtol=2e-5, atol=1e-6) and abs(float(out2["reward_mean"]) - float(out["reward_mean"])) < 1e-6
    env.close()
    if observe == "fused":
        env, _, _ = make(n=1001, level="level13", interval_sums=True)
        env.reset(seed=2)
        for k in range(3):
            env.step(actions(k, 1001))
        want = env.buffers["interval_sums"].double().sum(1)
        acc = torch.zeros(abi.NMETRIC + 3, dtype=torch.float32, device="cuda")
        env.interval_reduce(acc, 3.0 * 1001)
        assert float(env.buffers["interval_sums"].abs().sum()) == 0.0 and torch.allclose(acc[:abi.NMETRIC + 2].double(), want, rtol=2e-5, atol=1e-3)
        assert float(acc[abi.NMETRIC + 2]) == 3003.0
        env.close()


def test_configs4_one_rank_of_the_curriculum_shard():
    """BASELINE configs[4] (32768 envs, level1..10 curriculum over 8 GPUs) is 8 x this: rank r owns the global env ids
    [4096 r, 4096 (r + 1)) on its stage of the reference's level files (bench.py --workload curriculum).  Every stage at its full
    per-GPU size, with the global env-id offset of that rank: finite rollout with AutoReset, integer bookkeeping, the interval
    sums reduced through the trainer-side reducer, and the reference-compatible env properties."""
    from phase_guided_terrain_traversal_amd.distributed import MetricReducer, shard_range
    stages = [1, 2, 3, 4, 7, 10, 13]
    red = MetricReducer(torch.device("cuda", 0))
    for r, lev in enumerate(stages):
        lo, hi = shard_range(32768, r, 8)
        assert (lo, hi) == (4096 * r, 4096 * (r + 1))
        terrain = np.load(os.path.join(ASSETS, f"level{lev}.npy"))
        variant = np.random.Generator(np.random.Philox(key=[2, 0])).integers(0, terrain.shape[0], 32768).astype(np.int32)[lo:hi]
        env, _, _ = make(n=N, off=lo, level=f"level{lev}", variant=variant, interval_sums=True)
        assert env.observation_size == {"state": 171, "privileged_state": 215} and env.action_size == 12 and abs(env.dt - 0.02) < 1e-9
        assert env.xml_path.endswith("go2_stairs.json") and env.mj_model["_nbox"] == 100
        env.reset(seed=7)
        for k in range(20):
            obs, reward, done, info = env.step(actions(k))
        out = red.reduce_block(env.buffers["interval_sums"], 20.0 * N)
        torch.cuda.synchronize()
        assert float(out["env_steps"]) == 20 * N and torch.isfinite(out["metrics_mean"]).all() and 0.0 <= float(out["reward_mean"]) < 1.0
        for key in ("state", "obs_state", "obs_priv", "frame"):
            assert torch.isfinite(env.buffers[key]).all(), (lev, key)
        assert (env.buffers["istate"][abi.I_STEP] == 20).all()
        env.close()


def test_largest_single_gpu_batch_32768_envs():
    """the largest single-GPU configuration of BASELINE.json's list (32768 envs = configs[4] on one device; bench.py's "single-GPU saturation" row):
    automatic layout (quad beyond 8192 envs: two rounds of 1024 waves), AutoReset, 25 control steps - finite, counters exact, and the first 4096 envs
    equal, bit for bit, a 4096-env run pinned to the same layout (results do not depend on the batch size within a layout)"""
    from phase_guided_terrain_traversal_amd.env import Joystick
    from phase_guided_terrain_traversal_amd.randomize import domain_randomize
    big = 32768
    terrain = np.load(os.path.join(ASSETS, "level4.npy"))
    variant = domain_randomize(mjcf.load_model("stairs"), big, seed=2, terrain=terrain, enable=False)["variant"]
    assert len(np.unique(variant)) == terrain.shape[0]
    acts = [torch.from_numpy(np.tanh(np.random.Generator(np.random.Philox(key=[7, k])).normal(size=(big, 12)) * 0.6).astype(np.float32)).cuda() for k in range(25)]
    a = Joystick("stairs", configs.training_config(), num_envs=big, terrain=terrain, device="cuda:0", autoreset=True, variant=torch.from_numpy(variant))
    b = Joystick("stairs", configs.training_config(), num_envs=N, terrain=terrain, device="cuda:0", autoreset=True, variant=torch.from_numpy(variant[:N]), layout="quad")
    a.reset(seed=5); b.reset(seed=5)
    for k in range(25):
        a.step(acts[k]); b.step(acts[k][:N])
    sa, sb = snapshot(a), snapshot(b)
    for key in ("state", "obs_state", "obs_priv", "frame", "reward", "metrics", "scan_z"):
        assert torch.isfinite(sa[key]).all(), key
    assert (sa["istate"][abi.I_STEP] == 25).all()
    for key in ("state", "istate", "frame", "metrics"):
        assert torch.equal(sa[key][:, :N], sb[key]), key
    for key in ("obs_state", "obs_priv", "scan_z", "reward", "done"):
        assert torch.equal(sa[key][:N], sb[key]), key
    assert float(sa["done"].sum()) > 0                                         # some robots fell under random actions: AutoReset ran
    a.close(); b.close()
