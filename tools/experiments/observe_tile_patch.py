p='/root/repo/phase_guided_terrain_traversal_amd/csrc/pgtt_kernels.hip.h'; s=open(p).read()
def rep(a,b,first=False):
    global s
    assert s.count(a)>=1, a[:80]
    if not first: assert s.count(a)==1, (s.count(a), a[:80])
    s=s.replace(a,b,1)

rep('''template <int OMODE, bool HAS_TERRAIN>
// four waves per SIMD (128 VGPRs): the kernel is latency-bound, a launch lasts as long as the resident waves of a SIMD take in turn
__global__ __launch_bounds__(64, 4) void observe_kernel(KArgs a, const float* __restrict__ action) {
  const int e = xcd_block(blockIdx.x, gridDim.x), lane = threadIdx.x, N = a.N;''','''// WPB = waves (= envs) per workgroup.  WPB = 1: every row of the SoA state a wave touches is its own 64-byte request (lane r reads
// buffer[r][e]: ~650 requests per env, and the texture-address unit of a CU handles them one by one).  WPB = 16 (the step form,
// OBS_STEP): the 16 envs of a workgroup are neighbours in every row, so the workgroup moves the rows it reads (state, sensor frame,
// counters, the running episode / interval sums) and the rows it writes through an LDS TILE [row][env] with full 64-byte segments -
// 16 x fewer requests - and each wave works on its env's column.  Same arithmetic per env in both forms (bit-identical results).
constexpr int kObsTileRows = PGTT_NSTATE + PGTT_NFRAME + PGTT_NISTATE + 1 + 3 * (PGTT_NMETRIC + 2);
enum { OT_STATE = 0, OT_FRAME = PGTT_NSTATE, OT_ISTATE = OT_FRAME + PGTT_NFRAME, OT_DONE = OT_ISTATE + PGTT_NISTATE, OT_EPM = OT_DONE + 1,
       OT_IVL = OT_EPM + PGTT_NMETRIC + 2, OT_OUT = OT_IVL + PGTT_NMETRIC + 2 };
template <int OMODE, bool HAS_TERRAIN, int WPB>
// four waves per SIMD (128 VGPRs): the kernel is latency-bound, a launch lasts as long as the resident waves of a SIMD take in turn
__global__ __launch_bounds__(64 * WPB, 4) void observe_kernel(KArgs a, const float* __restrict__ action) {
  static_assert(WPB == 1 || OMODE == OBS_STEP, "the tile form is the step's");
  constexpr bool TILE = WPB > 1;
  constexpr int TS = WPB + 1;                       // tile row stride: odd, a wave's column walk hits every LDS bank
  const int wv = TILE ? (int)(threadIdx.x >> 6) : 0, lane = threadIdx.x & 63, N = a.N;
  const int blk = xcd_block(blockIdx.x, gridDim.x);
  const int e_raw = blk * WPB + wv;
  const bool env_ok = e_raw < N;                    // tile form with a ragged last workgroup: surplus waves run env N - 1 and store nothing
  const int e = env_ok ? e_raw : N - 1;
  // LDS traffic inside ONE wave needs no s_barrier (a wave's LDS instructions execute in order); with several waves per workgroup a
  // __syncthreads() here would also stall on the other envs
  auto wsync = [&]() { if constexpr (TILE) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } else __syncthreads(); };''')

rep('''  __shared__ float sh_st[PGTT_NSTATE];
  __shared__ float sh_fr[PGTT_NFRAME];
  __shared__ float sh_scan[128];
  __shared__ float sh_obs[PGTT_OBS + PGTT_PRIV + 2];
  __shared__ float sh_act[12];

  for (int r = lane; r < PGTT_NSTATE; r += 64) sh_st[r] = S[r * (long)N + e];
  for (int r = lane; r < PGTT_NFRAME; r += 64) sh_fr[r] = a.buf.frame[r * (long)N + e];
  if (OMODE == OBS_STEP && lane < 12) sh_act[lane] = action[(long)e * 12 + lane];
  __syncthreads();''','''  __shared__ float sh_st_[WPB][PGTT_NSTATE];
  __shared__ float sh_fr_[WPB][PGTT_NFRAME];
  __shared__ float sh_scan_[WPB][128];
  __shared__ float sh_obs_[WPB][PGTT_OBS + PGTT_PRIV + 2];
  __shared__ float sh_act_[WPB][12];
  __shared__ float tile[TILE ? kObsTileRows * TS : 1];
  float* const sh_st = sh_st_[wv]; float* const sh_fr = sh_fr_[wv]; float* const sh_scan = sh_scan_[wv];
  float* const sh_obs = sh_obs_[wv]; float* const sh_act = sh_act_[wv];
  // where a persistent row of this env lives: its global SoA slot, or its column of the tile
  auto SROW = [&](int row) -> float* { if constexpr (TILE) return &tile[(OT_STATE + row) * TS + wv]; else return &S[row * (long)N + e]; };
  auto IROW = [&](int row) -> int* { if constexpr (TILE) return reinterpret_cast<int*>(&tile[(OT_ISTATE + row) * TS + wv]); else return &I[row * (long)N + e]; };

  if constexpr (TILE) {
    // rows in: [row][16 envs] segments of 64 bytes, consecutive threads -> consecutive envs of one row
    const int e0 = blk * WPB;
    for (int idx = threadIdx.x; idx < OT_OUT * WPB; idx += 64 * WPB) {
      const int r = idx / WPB, c = idx - r * WPB;
      const long col = min(e0 + c, N - 1);
      float v = 0.f;
      if (r < OT_FRAME) v = S[r * (long)N + col];
      else if (r < OT_ISTATE) v = a.buf.frame[(r - OT_FRAME) * (long)N + col];
      else if (r < OT_DONE) v = __int_as_float(I[(r - OT_ISTATE) * (long)N + col]);
      else if (r == OT_DONE) v = a.buf.done[col];
      else if (r < OT_IVL) { if (a.buf.ep_metrics) v = a.buf.ep_metrics[(r - OT_EPM) * (long)N + col]; }
      else { if (a.buf.interval_sums) v = a.buf.interval_sums[(r - OT_IVL) * (long)N + col]; }
      tile[r * TS + c] = v;
    }
    if (lane < 12) sh_act[lane] = action[(long)e * 12 + lane];
    __syncthreads();
    for (int r = lane; r < PGTT_NSTATE; r += 64) sh_st[r] = tile[(OT_STATE + r) * TS + wv];
    for (int r = lane; r < PGTT_NFRAME; r += 64) sh_fr[r] = tile[(OT_FRAME + r) * TS + wv];
  } else {
    for (int r = lane; r < PGTT_NSTATE; r += 64) sh_st[r] = S[r * (long)N + e];
    for (int r = lane; r < PGTT_NFRAME; r += 64) sh_fr[r] = a.buf.frame[r * (long)N + e];
    if (OMODE == OBS_STEP && lane < 12) sh_act[lane] = action[(long)e * 12 + lane];
  }
  wsync();''')

rep('''    if (idx < PGTT_NSCAN) { sh_scan[idx] = z[h]; a.buf.scan_z[(long)e * PGTT_NSCAN + idx] = z[h]; }''','''    if (idx < PGTT_NSCAN) { sh_scan[idx] = z[h]; if (env_ok) a.buf.scan_z[(long)e * PGTT_NSCAN + idx] = z[h]; }''')

rep('''  const unsigned ep = (unsigned)I[PGTT_I_RNG_CTR * (long)N + e];
  int step_ctr = I[PGTT_I_STEP * (long)N + e];
  int timer = I[PGTT_I_STEPS_UNTIL_CMD * (long)N + e];
  int ep_steps = I[PGTT_I_EP_STEPS * (long)N + e];''','''  const unsigned ep = (unsigned)*IROW(PGTT_I_RNG_CTR);
  int step_ctr = *IROW(PGTT_I_STEP);
  int timer = *IROW(PGTT_I_STEPS_UNTIL_CMD);
  int ep_steps = *IROW(PGTT_I_EP_STEPS);''')
rep('''    step_ctr = 0; ep_steps = 0;
    __syncthreads();
    // info arrays that _get_obs reads
    if (lane < 12) { sh_st[PGTT_S_LAST_ACT + lane] = 0.f; sh_st[PGTT_S_LAST_LAST_ACT + lane] = 0.f; sh_st[PGTT_S_MOTOR_TARGETS + lane] = 0.f; }
    if (lane < 24) { sh_st[PGTT_S_QERR_HIST + lane] = 0.f; sh_st[PGTT_S_QVEL_HIST + lane] = 0.f; }
    __syncthreads();''','''    step_ctr = 0; ep_steps = 0;
    wsync();
    // info arrays that _get_obs reads
    if (lane < 12) { sh_st[PGTT_S_LAST_ACT + lane] = 0.f; sh_st[PGTT_S_LAST_LAST_ACT + lane] = 0.f; sh_st[PGTT_S_MOTOR_TARGETS + lane] = 0.f; }
    if (lane < 24) { sh_st[PGTT_S_QERR_HIST + lane] = 0.f; sh_st[PGTT_S_QVEL_HIST + lane] = 0.f; }
    wsync();''')
rep('''    prev_done = cfg->autoreset && a.buf.done[e] != 0.f;''','''    prev_done = cfg->autoreset && (TILE ? tile[OT_DONE * TS + wv] : a.buf.done[e]) != 0.f;''', first=True)
rep('''  __shared__ unsigned sh_rng[64 * 4];''','''  __shared__ unsigned sh_rng_[WPB][64 * 4];
  unsigned* const sh_rng = sh_rng_[wv];''')
rep('''    sh_rng[4 * lane + 0] = c0; sh_rng[4 * lane + 1] = c1; sh_rng[4 * lane + 2] = c2; sh_rng[4 * lane + 3] = c3;
  }
  __syncthreads();''','''    sh_rng[4 * lane + 0] = c0; sh_rng[4 * lane + 1] = c1; sh_rng[4 * lane + 2] = c2; sh_rng[4 * lane + 3] = c3;
  }
  wsync();''')
rep('''  __syncthreads();
  for (int i = lane; i < OBSD; i += 64) sh_obs[OBSD + i] = sh_obs[i];     // privileged = state || extras
  __syncthreads();''','''  wsync();
  for (int i = lane; i < OBSD; i += 64) sh_obs[OBSD + i] = sh_obs[i];     // privileged = state || extras
  wsync();''')
rep('''        float* p = a.buf.ep_metrics + k * (long)N + e;
        *p = (*p + add) * keep;''','''        float* p = TILE ? &tile[(OT_EPM + k) * TS + wv] : a.buf.ep_metrics + k * (long)N + e;
        *p = (*p + add) * keep;''', first=True)
i0=s.index("  if (lane < 3) S[(PGTT_S_CMD + lane) * (long)N + e] = sel4(lane, cmd[0], cmd[1], cmd[2], 0.f);")
i1=s.index("  const bool restore = OMODE == OBS_STEP && cfg->autoreset && wdone && a.buf.first_state && a.buf.first_obs;")
new='''  if (lane < 3) *SROW(PGTT_S_CMD + lane) = sel4(lane, cmd[0], cmd[1], cmd[2], 0.f);
  if (lane < 4) {
    *SROW(PGTT_S_PHASE + lane) = sel4(lane, phase[0], phase[1], phase[2], phase[3]);
    *SROW(PGTT_S_AIR_TIME + lane) = sel4(lane, air[0], air[1], air[2], air[3]);
    *SROW(PGTT_S_SWING_PEAK + lane) = sel4(lane, peak[0], peak[1], peak[2], peak[3]);
    *SROW(PGTT_S_HMAX + lane) = sel4(lane, hmax[0], hmax[1], hmax[2], hmax[3]);
    *SROW(PGTT_S_HMIN + lane) = sel4(lane, hmin[0], hmin[1], hmin[2], hmin[3]);
    *SROW(PGTT_S_LAST_CONTACT + lane) = sel4(lane, last_contact[0], last_contact[1], last_contact[2], last_contact[3]);
  }
  if (lane < 24) { *SROW(PGTT_S_QVEL_HIST + lane) = hist_v; *SROW(PGTT_S_QERR_HIST + lane) = hist_q; }
  if (lane < 12) {
    if (OMODE == OBS_STEP) {
      *SROW(PGTT_S_LAST_LAST_ACT + lane) = sh_st[PGTT_S_LAST_ACT + lane];
      *SROW(PGTT_S_LAST_ACT + lane) = act_i;
    } else {
      *SROW(PGTT_S_LAST_LAST_ACT + lane) = 0.f; *SROW(PGTT_S_LAST_ACT + lane) = 0.f;
      *SROW(PGTT_S_MOTOR_TARGETS + lane) = 0.f;
    }
  }
  if (lane == 0) {
    *SROW(PGTT_S_PHASE_DT) = phase_dt; *SROW(PGTT_S_GAIT_FREQ) = gait_freq;
    *IROW(PGTT_I_STEP) = step_ctr; *IROW(PGTT_I_STEPS_UNTIL_CMD) = timer;
    *IROW(PGTT_I_RNG_CTR) = (int)(ep + 1u); *IROW(PGTT_I_EP_STEPS) = ep_steps;
    if constexpr (TILE) { tile[(OT_OUT + PGTT_NMETRIC) * TS + wv] = reward; tile[(OT_OUT + PGTT_NMETRIC + 1) * TS + wv] = wdone ? 1.f : 0.f; }
    else { a.buf.reward[e] = reward; a.buf.done[e] = wdone ? 1.f : 0.f; }
  }
  for (int k = lane; k < PGTT_NMETRIC; k += 64) {
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < PGTT_NMETRIC; j++) if (j == k) v = metrics[j];
    if constexpr (TILE) {
      tile[(OT_OUT + k) * TS + wv] = v;
      tile[(OT_IVL + k) * TS + wv] += v;              // stored only when the caller bound interval_sums
    } else {
      a.buf.metrics[k * (long)N + e] = v;
      if (OMODE == OBS_STEP && a.buf.interval_sums) a.buf.interval_sums[k * (long)N + e] += v;
    }
  }
  if constexpr (TILE) { if (lane < 2) tile[(OT_IVL + PGTT_NMETRIC + lane) * TS + wv] += lane == 0 ? reward : (wdone ? 1.f : 0.f); }
  else if (OMODE == OBS_STEP && a.buf.interval_sums && lane < 2)
    a.buf.interval_sums[(PGTT_NMETRIC + lane) * (long)N + e] += lane == 0 ? reward : (wdone ? 1.f : 0.f);
'''
s=s[:i0]+new+s[i1:]
rep('''  if (restore) {
    for (int r = lane; r < PGTT_S_CMD; r += 64) S[r * (long)N + e] = a.buf.first_state[r * (long)N + e];
    const float* fo = a.buf.first_obs + (long)e * (OBSD + PRIVD);
    for (int i = lane; i < OBSD; i += 64) a.buf.obs_state[(long)e * OBSD + i] = fo[i];
    for (int i = lane; i < PRIVD; i += 64) a.buf.obs_priv[(long)e * PRIVD + i] = fo[OBSD + i];
  } else {
    for (int i = lane; i < OBSD; i += 64) a.buf.obs_state[(long)e * OBSD + i] = sh_obs[i];
    for (int i = lane; i < PRIVD; i += 64) a.buf.obs_priv[(long)e * PRIVD + i] = sh_obs[OBSD + i];
  }''','''  if (restore && env_ok) {
    // AutoReset-to-first-state (rare): qpos / qvel / warm-start rows straight to their SoA slots (the tile's write-back below covers
    // the rows from PGTT_S_CMD on only), observations from the rows captured at reset
    for (int r = lane; r < PGTT_S_CMD; r += 64) S[r * (long)N + e] = a.buf.first_state[r * (long)N + e];
    const float* fo = a.buf.first_obs + (long)e * (OBSD + PRIVD);
    for (int i = lane; i < OBSD; i += 64) a.buf.obs_state[(long)e * OBSD + i] = fo[i];
    for (int i = lane; i < PRIVD; i += 64) a.buf.obs_priv[(long)e * PRIVD + i] = fo[OBSD + i];
  } else if (env_ok) {
    for (int i = lane; i < OBSD; i += 64) a.buf.obs_state[(long)e * OBSD + i] = sh_obs[i];
    for (int i = lane; i < PRIVD; i += 64) a.buf.obs_priv[(long)e * PRIVD + i] = sh_obs[OBSD + i];
  }
  if constexpr (TILE) {
    // rows out: everything this kernel writes except the observation / scan rows (contiguous per env, stored above)
    __syncthreads();
    const int e0 = blk * WPB;
    for (int idx = threadIdx.x; idx < (kObsTileRows - PGTT_S_CMD) * WPB; idx += 64 * WPB) {
      const int r = PGTT_S_CMD + idx / WPB, c = idx % WPB;
      const long col = e0 + c;
      if (col >= N) continue;
      const float v = tile[r * TS + c];
      if (r < OT_FRAME) S[r * (long)N + col] = v;
      else if (r < OT_ISTATE) { }                                      // the sensor frame is read-only here
      else if (r < OT_DONE) I[(r - OT_ISTATE) * (long)N + col] = __float_as_int(v);
      else if (r == OT_DONE) { }
      else if (r < OT_IVL) { if (a.buf.ep_metrics && cfg->autoreset) a.buf.ep_metrics[(r - OT_EPM) * (long)N + col] = v; }
      else if (r < OT_OUT) { if (a.buf.interval_sums) a.buf.interval_sums[(r - OT_IVL) * (long)N + col] = v; }
      else if (r < OT_OUT + PGTT_NMETRIC) a.buf.metrics[(r - OT_OUT) * (long)N + col] = v;
      else if (r == OT_OUT + PGTT_NMETRIC) a.buf.reward[col] = v;
      else a.buf.done[col] = v;
    }
  }''')
open(p,'w').write(s)
print("ok")
