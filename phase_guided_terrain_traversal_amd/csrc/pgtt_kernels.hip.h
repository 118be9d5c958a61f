// pgtt_kernels.hip.h — __global__ kernels of libpgtt.so (gfx950).
//
//   physics_kernel  : one env per 4 or 16 lanes (lane = leg [x sub-lane]; 16 or 4 envs per 64-thread block, see
//                     pgtt_physics_quad.hip.h), DPP reductions.  MODE_STEP = mjx_env.step (n_substeps x
//                     {forward, Euler}) + sensor frame + contact flags; MODE_FORWARD = one mjx.forward
//                     (reset path).  Reference: go2/joystick_pgtt.py:146-148, :72, :78.
//   observe_kernel  : one env per WAVE.  13x9 height scan with the terrain variant's boxes read through
//                     wave-uniform addresses, quadrant statistics by wave reductions, the 171/215-dim
//                     observation rows assembled in LDS and stored coalesced, 21 rewards, bookkeeping,
//                     and (optionally) the Episode/AutoReset wrapper semantics.
//                     Reference: go2/joystick_pgtt.py:156-231, :238-370, go2/heightmap.py:25-67.
//   reset_pose_kernel: pose / velocity sampling of Joystick.reset (go2/joystick_pgtt.py:51-70).
#pragma once
#include "pgtt_physics_quad.hip.h"

namespace pgtt {

struct KArgs {
  const PgttModel* model;
  const PgttConfig* cfg;
  const TerrainBox* terrain;   // [T][B]
  const float4* cull;          // [T][B] (px, py, hx, hy) of the same boxes: the scan's cull reads 1.6 KB per variant instead of strided pieces of 8 KB
  const uint4* grid;           // [T][kGridG * kGridG]: boxes whose grown world AABB touches the cell (bit b of the 128 = box b)
  float grid_E, grid_inv;      // the grid covers [-E, E]^2, cell (ix, iy) = floor((x + E) * inv), clamped
  int T, B;
  PgttBuffers buf;
  int N;
  unsigned long long seed;
  long long env_off;
  const unsigned char* mask;
  float yaw_override;          // NaN = use the base yaw
  int write_qpos;              // MODE_FORWARD: store the (quaternion-normalised) qpos
  // test hooks (pgtt_set_test_overrides; both off in normal operation): with rng_fix != NaN every uniform draw returns rng_fix
  // (the reference-generated fixtures tests/golden/task_*.npz were produced with jax.random stubbed that way), and with
  // scan_preset != 0 the step's observe kernel takes the 117 scan heights from buf.scan_z instead of casting rays (the
  // fixtures hold scan values, not terrains)
  float rng_fix;
  int scan_preset;
  // Hand-over record of a control step, env-major [N][kHandover]: what the physics kernel computes and the observe kernel of the SAME
  // pgtt_step reads (qpos, qvel, motor targets, sensor frame).  The caller-visible rows stay the SoA [row][N] buffers, written as before;
  // but a wave that reads ITS env's 114 values out of them makes 114 requests for 128-byte lines, and the observe launch spends its first
  // ~5 us doing that (one request in ~23 ns per env, measured by leaving rows out).  From the record they are two coalesced loads.
  // handover_w: the physics launch writes it (every MODE_STEP launch does); handover_r: the observe launch may read it (pgtt_step only -
  // between pgtt_physics and pgtt_observe called on their own the caller may have edited the rows).
  float* handover_w;
  const float* handover_r;
#if defined(PGTT_TRACE) || defined(PGTT_TIME)
  float* trace;                // debugging builds only: per-iteration solver record of env 0
#endif
};

// ------------------------------------------------------------------ Philox4x32-10 (independent of the oracle's C)
PG_INL void philox4x32_10(unsigned k0, unsigned k1, unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    unsigned h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    unsigned h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    unsigned n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
PG_INL float rng_uniform(unsigned long long seed, unsigned env, unsigned epoch, unsigned stream, int idx) {
  unsigned c0 = env, c1 = epoch, c2 = stream, c3 = (unsigned)(idx >> 2);
  philox4x32_10((unsigned)seed, (unsigned)(seed >> 32), c0, c1, c2, c3);
  unsigned w = (idx & 3) == 0 ? c0 : ((idx & 3) == 1 ? c1 : ((idx & 3) == 2 ? c2 : c3));
  return (float)(w >> 8) * (1.0f / 16777216.0f);
}
PG_INL float rng_uniform(unsigned long long seed, unsigned env, unsigned epoch, unsigned stream, int idx, float fix) {
  const float u = rng_uniform(seed, env, epoch, stream, idx);
  return fix == fix ? fix : u;
}
PG_INL int exp_timer(unsigned long long seed, unsigned env, unsigned epoch, unsigned stream, float ctrl_dt, float fix) {
  double u = (double)rng_uniform(seed, env, epoch, stream, 0, fix);
  double t = -log1p(-u) * 5.0;
  return (int)rint(t / (double)ctrl_dt);
}

enum { MODE_STEP = 0, MODE_FORWARD = 1 };
constexpr int kHandover = 128;                 // floats per env (512 bytes: four lines)
enum { HO_QPOS = 0, HO_QVEL = 19, HO_MOTOR = 37, HO_FRAME = 49, HO_END = HO_FRAME + PGTT_NFRAME };
static_assert(HO_END <= kHandover, "hand-over record");

// Workgroup i is dispatched to XCD i % 8 and every XCD has its own L2.  Rows of the SoA state are contiguous over envs,
// so neighbouring envs share 128-byte lines: give each XCD a CONTIGUOUS range of logical blocks (MI355X_MICROARCH.md,
// "XCD-aware launches").  Identity when the grid is not a multiple of 8.
PG_INL int xcd_block(int bid, int nblocks) {
  if (nblocks & 7) return bid;
  return (bid & 7) * (nblocks >> 3) + (bid >> 3);
}

// ------------------------------------------------------------------ physics: one env per 4 * SUBS lanes (layouts: see pgtt_physics_quad.hip.h)
// SUBS is part of the kernel's name only (the layout itself is the translation unit's PG_SUBS).
template <int MODE, bool HAS_DR, bool HAS_TERRAIN, int SUBS>
__global__ __launch_bounds__(64) void physics_kernel(KArgs a, const float* __restrict__ action) {
  static_assert(SUBS == kSubs, "one lane layout per translation unit");
  const int N = a.N;
#ifdef PGTT_TIME
  const long long t0_cyc = __builtin_readcyclecounter(), t0_real = wall_clock64();
#endif
  const int l = lane_leg();                            // leg FL,FR,RL,RR
  const int blk = xcd_block(blockIdx.x, gridDim.x);
  int e = blk * kEnvsPerWave + lane_env();
  bool valid = e < N;
  if (!valid) e = N - 1;                               // keep whole quads running (DPP), suppress the stores
  if (MODE != MODE_STEP && a.mask && !a.mask[e]) valid = false;
  // Hex layout: the model constants (sizeof(PgttModel) = 2.5 KB, read ~100 times per substep through uniform or per-leg addresses) are staged
  // in LDS once per launch: with one wave per SIMD every wait for a vector-memory round trip is exposed, and an LDS read returns in about
  // half the time of an L1 hit (80 global loads of the step kernel became LDS reads: bit-identical, level4 168.0 -> 166.7 us, flat 118.1 ->
  // 115.9 us at 4096 envs).  Not in the oct layout, where the change costs 52 B of scratch per lane and 0.5 - 0.8 %.
  // Order of the prologue: the variant index first (two dependent round trips hang on it: index -> box records), then every other load of
  // the launch - model image, per-env model, state rows, action - and ONE barrier behind all the staging stores; the prologue reads the
  // model through its global pointer (gm), everything after the barrier through `m`.
  __shared__ unsigned sh_model[kSubs == 4 ? (sizeof(PgttModel) + 3) / 4 : 1];
  const PgttModel* __restrict__ gm = a.model;
  // a label outside [0, T) would index past the terrain tables: clamped (v_med3, identity for a valid label; pgtt_reset reports such labels as PGTT_E_ARG)
  const int variant = (HAS_TERRAIN && a.buf.variant) ? min(max(a.buf.variant[e], 0), a.T - 1) : 0;
  constexpr int kModelWords = (int)((sizeof(PgttModel) + 3) / 4), kModelTrips = (kModelWords + 63) / 64;
  unsigned mw[kSubs == 4 ? kModelTrips : 1];
  if (kSubs == 4) {
#pragma unroll
    for (int t = 0; t < kModelTrips; t++) { const int i = t * 64 + (int)threadIdx.x; mw[t] = reinterpret_cast<const unsigned*>(a.model)[i < kModelWords ? i : 0]; }
  }
  const PgttConfig* __restrict__ cfg = a.cfg;
  float* __restrict__ S = a.buf.state;
  // Base-body rows are stored by ALL four lanes of the quad (same address, bit-identical value): the kernel has no
  // region in which only part of a quad is active while replicated state is live ("quad invariants").
  const bool lead = valid;

  QEnvModel em;
  qload_env_model<HAS_DR>(gm, a.buf.params, N, e, l, em);
  QSim s;
#ifdef PGTT_TIME
  s.tlast = t0_cyc;
#endif
#pragma unroll
  for (int i = 0; i < 7; i++) s.qb[i] = PG_ROW(S, PGTT_S_QPOS + i, N, e);
#pragma unroll
  for (int i = 0; i < 6; i++) { s.vb[i] = PG_ROW(S, PGTT_S_QVEL + i, N, e); s.wb[i] = PG_ROW(S, PGTT_S_QWARM + i, N, e); }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int j = 3 * l + k, ac = 3 * (l ^ 1) + k;     // joint index, actuator index driving it
    s.ql[k] = PG_ROW(S, PGTT_S_QPOS + 7 + j, N, e);
    s.vl[k] = PG_ROW(S, PGTT_S_QVEL + 6 + j, N, e);
    s.wl[k] = PG_ROW(S, PGTT_S_QWARM + 6 + j, N, e);
    if (MODE == MODE_STEP) s.ctrl[k] = gm->key_qpos[7 + ac] + PG_REC(action, e, 12, ac) * cfg->action_scale;
    else s.ctrl[k] = PG_ROW(S, PGTT_S_QPOS + 7 + ac, N, e);          // mjx_env.init(ctrl = qpos[7:])
  }
  const TerrainBox* boxes = nullptr;
  const uint4* grid_v = nullptr;
  unsigned box0 = 0u, cell0 = 0u;       // PG_ADDR32: first box / grid cell of the env's variant as 32-bit element indices from the tables' bases
  int nbox = 0;
  // LDS staging of the env's terrain variant: centre + bounding radius of its <=100 boxes (read 2 x 4 substeps
  // by the broad phase), and a per-lane column for the broad-phase keys of the own foot
  __shared__ float4 sh_box[HAS_TERRAIN && kBoxLds ? PGTT_MAX_BOX * kEnvsPerWave : 1];      // (cx, cy, cz, hx)
  __shared__ float2 sh_box2[HAS_TERRAIN && kBoxLds ? PGTT_MAX_BOX * kEnvsPerWave : 1];     // (hy, hz)
  __shared__ float sh_con[HAS_TERRAIN ? kMaxB * kSlotFields * kSlotCols : 1];
  const int quad = lane_env();                         // env within the wave
  const BoxSlots slots{sh_con, lane_col()};
  if (HAS_TERRAIN) {
#if PG_ADDR32
    boxes = a.terrain; grid_v = a.grid;
    box0 = (unsigned)variant * (unsigned)a.B; cell0 = (unsigned)variant * (unsigned)(kGridG * kGridG);
#else
    boxes = a.terrain + (long)variant * a.B;
    grid_v = a.grid + (long)variant * (kGridG * kGridG);
#endif
    nbox = a.B;
    for (int b = lane_in_env(); kBoxLds && b < nbox; b += 4 * kSubs) {
      const TerrainBox* tb = PG_ADDR32 ? &pg_at(boxes, box0 + (unsigned)b) : boxes + b;
      sh_box[b * kEnvsPerWave + quad] = make_float4(tb->px, tb->py, tb->pz, tb->hx);
      sh_box2[b * kEnvsPerWave + quad] = make_float2(tb->hy, tb->hz);
    }
    slots.clear_all();
  }
  const PgttModel* __restrict__ m = gm;
  if (kSubs == 4) {
#pragma unroll
    for (int t = 0; t < kModelTrips; t++) { const int i = t * 64 + (int)threadIdx.x; if (i < kModelWords) sh_model[i] = mw[t]; }
    m = reinterpret_cast<const PgttModel*>(sh_model);
  }
  if (HAS_TERRAIN || kSubs == 4) __syncthreads();
  s.niter = 0; s.niter_max = 0; s.pen_overflow = false;
  QPhysics ph(m, em, s, l);
  QSolver sol(m, s, slots);
  sol.lds_slots = HAS_TERRAIN && nbox > 0;
#ifdef PGTT_TRACE
  if (valid && e == 0 && a.trace && lane_sub() == 0) s.tr = a.trace + l;
#endif
  const int nsub = MODE == MODE_STEP ? cfg->n_substeps : 1;
  const float dt = m->timestep;
  PG_TICK(s, 15);          // launch prologue: state rows, per-env model, LDS staging of the terrain variant
  for (int sub = 0; sub < nsub; sub++) {
    PG_TICK(s, 9);
    ph.kinematics();
    PG_TICK(s, 0);
    if (HAS_TERRAIN) ph.collide(boxes, box0, nbox, sh_box, sh_box2, slots, quad, grid_v, cell0, a.grid_E, a.grid_inv); else s.nbox = 0;
    PG_TICK(s, 16);
    ph.inertia();
    PG_TICK(s, 0);
    ph.velocity_stage();
    PG_TICK(s, 1);
    ph.constraint_stage(HAS_TERRAIN && boxes != nullptr && nbox > 0, a.buf.box_friction, N, e, slots);
    PG_TICK(s, 2);
    // ---- sensors of the last forward (pre-integration state), written BEFORE the solve; the accelerometer is an affine map of
    //      qacc[0:6]: its constant part is kept across the solve (3 values), the 3 x 6 matrix is formed after it from frames that are
    //      still live (R0, cdr, imu, com) - 18 registers less across the Newton loop of a kernel that spills to scratch
    float acc0[3];
    const bool last = sub == nsub - 1;
    if (last) {
      float* __restrict__ Fr = a.buf.frame;
      int ee = e; asm volatile("" : "+v"(ee));     // re-form the row addresses here (see the final stores)
      V3 w = s.cvel0.a, vl = s.cvel0.l;
      V3 dif = s.imu - s.com;
      V3 gyro = mtmul(s.R0, w);
      V3 glin = vl - cross(dif, w);
      V3 llin = mtmul(s.R0, glin);
      S6 cacc{v3(0, 0, 0), v3(-m->gravity[0], -m->gravity[1], -m->gravity[2])};
#pragma unroll
      for (int k = 0; k < 3; k++) cacc = cacc + s.cddr[k] * s.vb[3 + k];
      V3 a0 = mtmul(s.R0, cacc.l - cross(dif, cacc.a)) + cross(gyro, llin);
      acc0[0] = a0.x; acc0[1] = a0.y; acc0[2] = a0.z;
      float* __restrict__ Ho = &PG_REC(a.handover_w, ee, kHandover, HO_FRAME);      // MODE_STEP: the same values, env-major, for this step's observe launch
      auto put1 = [&](int row, float v) { PG_ROW(Fr, row, N, ee) = v; if (MODE == MODE_STEP) Ho[row] = v; };
      auto put3 = [&](int row, V3 v) { put1(row, v.x); put1(row + 1, v.y); put1(row + 2, v.z); };
      if (lead) {
        put3(PGTT_F_GYRO, gyro); put3(PGTT_F_GLOBAL_LINVEL, glin); put3(PGTT_F_GLOBAL_ANGVEL, w); put3(PGTT_F_LOCAL_LINVEL, llin);
        put3(PGTT_F_UPVECTOR, v3(s.R0.m[2], s.R0.m[5], s.R0.m[8]));
        put3(PGTT_F_GRAVITY, v3(-s.R0.m[6], -s.R0.m[7], -s.R0.m[8]));
      }
      // own foot: sensor order FR,FL,RR,RL = leg ^ 1
      const int f = l ^ 1;
      bool touching = s.con0.dist < 0.f;
      if (HAS_TERRAIN) {
#pragma unroll
        for (int k = 0; k < kMaxB; k++) touching |= (k < s.nbox) & (slots.at(k, 0) < 0.f);
      }
      // box-contact slot numbering of the debug record: own contacts follow those of the lower legs
      const int n0 = quad_bcast<0>(s.nbox), n1 = quad_bcast<1>(s.nbox), n2 = quad_bcast<2>(s.nbox), n3 = quad_bcast<3>(s.nbox);
      const int off = l == 0 ? 0 : (l == 1 ? n0 : (l == 2 ? n0 + n1 : n0 + n1 + n2)), total = n0 + n1 + n2 + n3;
      if (valid) {
        put3(PGTT_F_FEET_POS + 3 * f, mtmul(s.R0, s.sitef - s.imu));
        S6 cv = s.cvell[2];
        put3(PGTT_F_FEET_VEL + 3 * f, cv.l - cross(s.sitef - s.com, cv.a));
        put1(PGTT_F_CONTACT + f, touching ? 1.0f : 0.0f);
        put1(PGTT_F_FOOT_SITE_Z + f, s.sitef.z);
#pragma unroll
        for (int k = 0; k < 3; k++) put1(PGTT_F_ACT_FORCE + 3 * f + k, s.act_force[k]);
        if (a.buf.dbg_contact && a.buf.dbg_dist) {
          int* dc = a.buf.dbg_contact + (long)ee * 16; float* dd = a.buf.dbg_dist + (long)ee * 8;
          dc[2 * l] = l; dc[2 * l + 1] = -1; dd[l] = s.con0.dist;
          if (HAS_TERRAIN) {
#pragma unroll
            for (int k = 0; k < kMaxB; k++) if (k < s.nbox && off + k < 4) {
              dc[2 * (4 + off + k)] = l; dc[2 * (4 + off + k) + 1] = __float_as_int(slots.at(k, 20)); dd[4 + off + k] = slots.at(k, 0); }
          }
#pragma unroll
          for (int k = 0; k < 4; k++) if (k >= total) { dc[2 * (4 + k)] = -1; dc[2 * (4 + k) + 1] = -2; dd[4 + k] = 1.0f; }
        }
      }
    }
    sol.solve();
    const int pen_ovf = last ? quad_sum_i(sub_sum_i(s.pen_overflow ? 1 : 0)) : 0;      // over the lanes of the env, all lanes active
    if (last && lead) {
      float* __restrict__ Fr = a.buf.frame;
      int ee = e; asm volatile("" : "+v"(ee));
      float accA[3][6];
      const V3 dif = s.imu - s.com;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        V3 ct = mtmul(s.R0, v3(k == 0, k == 1, k == 2));
        V3 cr = mtmul(s.R0, s.cdr[k].l - cross(dif, s.cdr[k].a));
        accA[0][k] = ct.x; accA[1][k] = ct.y; accA[2][k] = ct.z;
        accA[0][3 + k] = cr.x; accA[1][3 + k] = cr.y; accA[2][3 + k] = cr.z;
      }
#pragma unroll
      for (int r = 0; r < 3; r++) {
        float v = acc0[r];
#pragma unroll
        for (int k = 0; k < 6; k++) v += accA[r][k] * s.qacc_b[k];
        PG_ROW(Fr, PGTT_F_ACCEL + r, N, ee) = v;
        if (MODE == MODE_STEP) PG_REC(a.handover_w, ee, kHandover, HO_FRAME + PGTT_F_ACCEL + r) = v;
      }
      if (a.buf.dbg_niter) a.buf.dbg_niter[ee] = s.niter_max | (pen_ovf > 0 ? PGTT_DBG_PEN_OVERFLOW : 0);
    }
    if (MODE == MODE_STEP) {
      // ---- semi-implicit Euler (eulerdamp disabled)
#pragma unroll
      for (int i = 0; i < 6; i++) s.vb[i] = s.vb[i] + s.qacc_b[i] * dt;
#pragma unroll
      for (int k = 0; k < 3; k++) s.vl[k] = s.vl[k] + s.qacc_l[k] * dt;
#pragma unroll
      for (int i = 0; i < 3; i++) s.qb[i] = s.qb[i] + dt * s.vb[i];
      V3 wv = v3(s.vb[3], s.vb[4], s.vb[5]);
      float nn = normalize3(wv);
      float sn, cs; sincosf(0.5f * (dt * nn), &sn, &cs);
      Q4 q2 = qmul(Q4{s.qb[3], s.qb[4], s.qb[5], s.qb[6]}, Q4{cs, wv.x * sn, wv.y * sn, wv.z * sn});
      normalize4(q2);
      s.qb[3] = q2.w; s.qb[4] = q2.x; s.qb[5] = q2.y; s.qb[6] = q2.z;
#pragma unroll
      for (int k = 0; k < 3; k++) s.ql[k] = s.ql[k] + dt * s.vl[k];
    }
  }
#ifdef PGTT_TIME
  // stage ticks of the wave that owns env PGTT_TIME (e.g. -DPGTT_TIME=0): 0 position 1 velocity 2 constraint 3 sensors
  // 4 solver init x3 5 first gradient 6 line search 7 update_constraint 8 update_gradient 9 rest 10 #iterations
  if (e == PGTT_TIME && a.trace) {
    for (int i = 0; i < 20; i++) a.trace[i] = s.cyc[i];
    for (int i = 20; i < 28; i++) a.trace[4 + i] = s.cyc[i];      // line-search sub-stages at [24..31]
    // whole-kernel span of this wave in shader-clock ticks and in ticks of the constant 100 MHz clock (gives the shader clock rate)
    a.trace[20] = (float)(__builtin_readcyclecounter() - t0_cyc); a.trace[21] = (float)(wall_clock64() - t0_real);
  }
  if (a.trace) {      // per-wave totals: [32 + block] ticks of the whole kernel, [32 + 4096 + block] sum over substeps of nslots
    float tot = 0.f;
    for (int i = 0; i < 10; i++) tot += i == 9 ? 0.f : s.cyc[i];
    for (int i = 11; i < 17; i++) tot += s.cyc[i];
    a.trace[32 + blockIdx.x] = tot; a.trace[32 + 4096 + blockIdx.x] = s.cyc[17];
    // placement and timeline of the wave (blocks < 1024): HW_ID, XCC_ID, start / end on the constant 100 MHz clock
    if (blockIdx.x < 1024) {
      unsigned* tw = (unsigned*)(a.trace + 32 + 8192 + 4 * blockIdx.x);
      tw[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4); tw[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
      tw[2] = (unsigned)t0_real; tw[3] = (unsigned)wall_clock64();
      for (int i = 0; i < 20; i++) a.trace[16384 + 24 * blockIdx.x + i] = s.cyc[i];      // stage ticks of every wave
    }
  }
#endif
  if (!valid) return;
#ifdef PGTT_EFFORT
  if (a.buf.dbg_contact) { a.buf.dbg_contact[(long)e * 16 + 14] = (int)(unsigned)s.eff; a.buf.dbg_contact[(long)e * 16 + 15] = (int)(unsigned)(s.eff >> 32);
    a.buf.dbg_contact[(long)e * 16 + 12] = s.eff_hess; a.buf.dbg_contact[(long)e * 16 + 13] = s.eff_hess_same; }       // the same for every env of the wave
#endif
  // The compiler would otherwise keep the ~50 row addresses formed for the loads at the top alive (spilled to scratch)
  // until these stores: an opaque copy of the env index makes it re-form them here (one mad each).
  asm volatile("" : "+v"(e));
#if PG_ADDR32
  int lq = l; asm volatile("" : "+v"(lq));      // ... and of the leg index: the per-leg row offsets (3 l + k) N + e are formed again as well
#else
  const int lq = l;
#endif
  if (MODE == MODE_STEP || a.write_qpos) {
#pragma unroll
    for (int i = 0; i < 7; i++) PG_ROW(S, PGTT_S_QPOS + i, N, e) = s.qb[i];
#pragma unroll
    for (int k = 0; k < 3; k++) PG_ROW(S, PGTT_S_QPOS + 7 + 3 * lq + k, N, e) = s.ql[k];
  }
  if (MODE == MODE_STEP) {
#pragma unroll
    for (int i = 0; i < 6; i++) PG_ROW(S, PGTT_S_QVEL + i, N, e) = s.vb[i];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      PG_ROW(S, PGTT_S_QVEL + 6 + 3 * lq + k, N, e) = s.vl[k];
      PG_ROW(S, PGTT_S_MOTOR_TARGETS + 3 * (lq ^ 1) + k, N, e) = s.ctrl[k];
    }
    float* __restrict__ Ho = &PG_REC(a.handover_w, e, kHandover, 0);
#pragma unroll
    for (int i = 0; i < 7; i++) Ho[HO_QPOS + i] = s.qb[i];
#pragma unroll
    for (int i = 0; i < 6; i++) Ho[HO_QVEL + i] = s.vb[i];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      Ho[HO_QPOS + 7 + 3 * lq + k] = s.ql[k]; Ho[HO_QVEL + 6 + 3 * lq + k] = s.vl[k]; Ho[HO_MOTOR + 3 * (lq ^ 1) + k] = s.ctrl[k];
    }
  }
#pragma unroll
  for (int i = 0; i < 6; i++) PG_ROW(S, PGTT_S_QWARM + i, N, e) = s.wb[i];
#pragma unroll
  for (int k = 0; k < 3; k++) PG_ROW(S, PGTT_S_QWARM + 6 + 3 * lq + k, N, e) = s.wl[k];
}

// ------------------------------------------------------------------ reset: pose sampling (go2/joystick_pgtt.py:51-70)
template <int UNUSED>
__global__ __launch_bounds__(64) void reset_pose_kernel(KArgs a) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  const int N = a.N;
  if (e >= N) return;
  if (a.mask && !a.mask[e]) return;
  const PgttModel* __restrict__ m = a.model;
  float* __restrict__ S = a.buf.state;
  const unsigned id = (unsigned)(a.env_off + e);
  const unsigned ep = (unsigned)a.buf.istate[PGTT_I_RNG_CTR * (long)N + e];
  float qpos[19];
#pragma unroll
  for (int i = 0; i < 19; i++) qpos[i] = m->key_qpos[i];
  qpos[0] += rng_uniform(a.seed, id, ep, PGTT_RS_RESET_XY, 0, a.rng_fix) * 1.0f + -0.5f;
  qpos[1] += rng_uniform(a.seed, id, ep, PGTT_RS_RESET_XY, 1, a.rng_fix) * 1.0f + -0.5f;
  float yaw = rng_uniform(a.seed, id, ep, PGTT_RS_RESET_YAW, 0, a.rng_fix) * 6.28f + -3.14f;
  float sn, cs; sincosf(0.5f * yaw, &sn, &cs);
  Q4 q = qmul(Q4{qpos[3], qpos[4], qpos[5], qpos[6]}, Q4{cs, 0.f * sn, 0.f * sn, 1.f * sn});
  qpos[3] = q.w; qpos[4] = q.x; qpos[5] = q.y; qpos[6] = q.z;
#pragma unroll
  for (int i = 0; i < 19; i++) S[(PGTT_S_QPOS + i) * (long)N + e] = qpos[i];
#pragma unroll
  for (int i = 0; i < 18; i++) {
    S[(PGTT_S_QVEL + i) * (long)N + e] = i < 6 ? rng_uniform(a.seed, id, ep, PGTT_RS_RESET_VEL, i, a.rng_fix) * 0.2f + -0.1f : 0.f;
    S[(PGTT_S_QWARM + i) * (long)N + e] = 0.f;
  }
}

// ------------------------------------------------------------------ observe: one env per wave
// OBS_STEP_OBS: the scan + observation half of a step (rewards / bookkeeping are done by task_kernel, one env per LANE)
enum { OBS_STEP = 0, OBS_SCAN_LIFT = 1, OBS_RESET = 2, OBS_SCAN_ONLY = 3, OBS_STEP_OBS = 4 };

// wave-wide max / min (all 64 lanes active): DPP butterflies inside the 16-lane rows, then the four row results through
// scalar registers - no LDS crossbar (ds_bpermute) round trips
PG_INL float wave_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x124>(v)); v = fmaxf(v, dpp_f<0x128>(v));
  const int i = __float_as_int(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(i, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(i, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(i, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(i, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
PG_INL float wave_min(float v) {
  v = fminf(v, dpp_f<0xB1>(v)); v = fminf(v, dpp_f<0x4E>(v)); v = fminf(v, dpp_f<0x124>(v)); v = fminf(v, dpp_f<0x128>(v));
  const int i = __float_as_int(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(i, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(i, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(i, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(i, 48));
  return fminf(fminf(r0, r1), fminf(r2, r3));
}

// vertical ray (0,0,-1) from world point p against one terrain box; generic mjx _ray_box in the box frame.  The ray
// direction in the box frame lv = -R[2][:] is the same for every ray of a box: its reciprocals are formed once per box
// (v_rcp_f32, 1 ulp) and the six face parameters are (side - lp) * (1 / lv) instead of six divisions per ray; an
// axis-parallel component gives 1 / (+-0) = +-inf and the same +-inf / NaN face parameters as the division.
PG_INL float mul_unfused(float x, float y) {
#pragma clang fp contract(off)
  return x * y;
}
struct RayBox { float il[3]; bool upright; };
PG_INL RayBox ray_box_prepare(const TerrainBox& tb) {
  // upright: the box z axis is along the world z axis (every shipped / generated terrain: boxes are only turned about z).  The ray
  // then runs along the box z axis: lv = (-0, -0, -m22), the four side faces get the parameters (side - lp) * rcp(-0) = +-inf / NaN
  // and hit points lp + x * (-0) = NaN, so they are never valid, while on the two z faces p0 = lp0 + x * (-0) = lp0 exactly - the generic
  // test reduces, bit for bit, to its two z faces.  m22 itself need not be 1: the tables hold float32 quaternions, and a box turned by
  // 90 degrees (0.70710677, 0, 0, 0.70710677) has m22 = w^2 + z^2 = 0.99999994 (half of the boxes of the level files).
  return RayBox{{__builtin_amdgcn_rcpf(-tb.m20), __builtin_amdgcn_rcpf(-tb.m21), __builtin_amdgcn_rcpf(-tb.m22)},
                tb.m20 == 0.f && tb.m21 == 0.f && tb.m22 != 0.f};
}
PG_INL float ray_box_down(const TerrainBox& tb, const RayBox& rb, V3 p) {
  V3 rel = p - v3(tb.px, tb.py, tb.pz);
  float lp[3] = {tb.m00 * rel.x + tb.m10 * rel.y + tb.m20 * rel.z, tb.m01 * rel.x + tb.m11 * rel.y + tb.m21 * rel.z,
                 tb.m02 * rel.x + tb.m12 * rel.y + tb.m22 * rel.z};
  float lv[3] = {-tb.m20, -tb.m21, -tb.m22};
  float sz[3] = {tb.sx, tb.sy, tb.sz};
  float best = INFINITY;
  if (rb.upright) {           // wave-uniform (the box record sits in scalar registers)
    const bool inside = (fabsf(lp[0]) <= sz[0]) & (fabsf(lp[1]) <= sz[1]);      // p0 = lp[0] + x * (-0) = lp[0], p1 likewise
    const float xt = (sz[2] - lp[2]) * rb.il[2], xb = (-sz[2] - lp[2]) * rb.il[2];
    best = (inside & (xt >= 0.f)) ? xt : best;
    best = (inside & (xb >= 0.f) & (xb < best)) ? xb : best;
    return best;
  }
#pragma unroll
  for (int f = 0; f < 6; f++) {
    const int ax = f % 3, i0 = ax == 0 ? 1 : 0, i1 = ax == 2 ? 1 : 2;
    float side = f < 3 ? sz[ax] : -sz[ax];
    float x = (side - lp[ax]) * rb.il[ax];
    float p0 = lp[i0] + x * lv[i0], p1 = lp[i1] + x * lv[i1];
    const bool valid = (fabsf(p0) <= sz[i0]) & (fabsf(p1) <= sz[i1]) & (x >= 0.f);     // `&`: selects, not branches
    best = (valid & (x < best)) ? x : best;
  }
  return best;
}

PG_INL float cubic_hermite(float t, float p0, float p1, float m0, float m1) {
  float t2 = t * t, t3 = t2 * t;
  return (2 * t3 - 3 * t2 + 1) * p0 + (t3 - 2 * t2 + t) * m0 + (-2 * t3 + 3 * t2) * p1 + (t3 - t2) * m1;
}
PG_INL float gait_get_z(float phi, float swing_height, float swing_min) {
  const float T_swing = (float)(2 * M_PI * (1 - 0.5) / 2), T_peak = (float)(2 * M_PI * (1 + 0.5) / 2), T_stance = (float)(2 * M_PI * 0.5);
  if (phi <= T_stance) return swing_min;
  if (phi <= T_peak) return cubic_hermite((phi - T_stance) / T_swing, swing_min, swing_height, T_swing * 0.f, T_swing * 0.f);
  return cubic_hermite((phi - T_peak) / T_swing, swing_height, swing_min, T_swing * 0.f, T_swing * 0.f);
}

// fmodf(x, y) for 0 <= x < 2 y - the phase clock: phase in [0, 2 pi) plus an increment below 2 pi.  There fmod is x or x - y, and x - y is exact
// (Sterbenz: y <= x <= 2 y), so the select returns fmodf's bits at 3 instructions instead of ~32 per foot; anything else takes the library call
PG_INL float fmod_once(float x, float y) {
  if (__builtin_expect(__ballot(!((x >= 0.f) & (x < 2.0f * y))) != 0ull, 0)) return fmodf(x, y);
  return x >= y ? x - y : x;
}

// Rewards, termination and bookkeeping of one control step (joystick_pgtt.py:193-227 / joystick.py): shared by the fused
// observe kernel (sh_* in LDS) and by task_kernel (sh_* = per-lane arrays, all indices compile-time constants).
struct TaskScalars {
  float cmd[3], phase[4], air[4], peak[4], hmax[4], last_contact[4], contact[4], first_contact[4];
  float phase_dt; int step_ctr, timer;
  bool done; float reward; float metrics[PGTT_NMETRIC];
};
// sums over the 16 lanes of a DPP row / the 4 lanes of a quad, the same bits in every lane (symmetric butterflies, plain adds)
PG_INL float row16_sum(float x) {
#pragma clang fp contract(off)
  x = x + dpp_f<0xB1>(x); x = x + dpp_f<0x4E>(x); x = x + dpp_f<0x128>(x); x = x + dpp_f<0x124>(x);
  return x;
}
PG_INL float quad4_sum(float x) {
#pragma clang fp contract(off)
  x = x + dpp_f<0xB1>(x); x = x + dpp_f<0x4E>(x);
  return x;
}
PG_INL float quad4_min(float x) { x = fminf(x, dpp_f<0xB1>(x)); x = fminf(x, dpp_f<0x4E>(x)); return x; }
// WAVE = true (fused observe kernel, one env per wave, every lane holds the same per-env scalars): the 12 joint terms and the
// 4 foot terms are evaluated by the lanes of a row / a quad in parallel (lane & 15 = joint, lane & 3 = foot) and summed
// with DPP butterflies - a fraction of the instructions of the serial loops and of the ~130 VGPRs their unrolled bodies keep
// alive.  WAVE = false (task_kernel, one env per lane): the serial loops.
template <bool WAVE>
PG_INL void task_rewards(const float* sh_st, const float* sh_fr, const float* sh_act, const PgttConfig* __restrict__ cfg,
                         const PgttModel* __restrict__ m, bool baseline, unsigned long long seed, unsigned id, unsigned ep, float dt,
                         float rng_fix, TaskScalars& t) {
  float (&cmd)[3] = t.cmd; float (&phase)[4] = t.phase; float (&air)[4] = t.air; float (&peak)[4] = t.peak; float (&hmax)[4] = t.hmax;
  float (&last_contact)[4] = t.last_contact; float (&contact)[4] = t.contact; float (&first_contact)[4] = t.first_contact;
  float (&metrics)[PGTT_NMETRIC] = t.metrics;
  const float phase_dt = t.phase_dt; int& step_ctr = t.step_ctr; int& timer = t.timer; bool& done = t.done; float& reward = t.reward;
  struct { unsigned long long seed; } a{seed};
    done = sh_fr[PGTT_F_UPVECTOR + 2] < 0.f;
    float rew[PGTT_NREW];
    const float cmd_norm = sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1] + cmd[2] * cmd[2]);
    // the three exponentials (both tracking terms, feet_phase): WAVE = true evaluates them in ONE expf expansion, lanes 0 / 1 / 2
    float xlin, xang;
    {
      float e0 = cmd[0] - sh_fr[PGTT_F_LOCAL_LINVEL], e1 = cmd[1] - sh_fr[PGTT_F_LOCAL_LINVEL + 1];
      xlin = -(e0 * e0 + e1 * e1) / cfg->tracking_sigma;
      float ea = cmd[2] - sh_fr[PGTT_F_GYRO + 2];
      xang = -(ea * ea) / cfg->tracking_sigma;
    }
    rew[PGTT_R_LIN_VEL_Z] = sh_fr[PGTT_F_GLOBAL_LINVEL + 2] * sh_fr[PGTT_F_GLOBAL_LINVEL + 2];
    rew[PGTT_R_ANG_VEL_XY] = sh_fr[PGTT_F_GLOBAL_ANGVEL] * sh_fr[PGTT_F_GLOBAL_ANGVEL] + sh_fr[PGTT_F_GLOBAL_ANGVEL + 1] * sh_fr[PGTT_F_GLOBAL_ANGVEL + 1];
    rew[PGTT_R_ORIENTATION] = sh_fr[PGTT_F_UPVECTOR] * sh_fr[PGTT_F_UPVECTOR] + sh_fr[PGTT_F_UPVECTOR + 1] * sh_fr[PGTT_F_UPVECTOR + 1];
    {
      float sa = 0.f, lim = 0.f, pose = 0.f, s2 = 0.f, s1 = 0.f, en = 0.f, ar = 0.f;
      auto joint = [&](int i, float on) {
        float q = sh_st[PGTT_S_QPOS + 7 + i], dq = q - m->key_qpos[7 + i];
        sa += on * fabsf(dq);
        pose += on * ((dq * dq) * ((i % 3) == 0 ? 1.0f : 0.1f));
        float lo = m->jnt_range[i][0] * cfg->soft_joint_pos_limit_factor, hi = m->jnt_range[i][1] * cfg->soft_joint_pos_limit_factor;
        lim += on * (-fminf(q - lo, 0.f) + fmaxf(q - hi, 0.f));
        float f = sh_fr[PGTT_F_ACT_FORCE + i];
        s2 += on * (f * f); s1 += on * fabsf(f);
        en += on * (fabsf(sh_st[PGTT_S_QVEL + 6 + i]) * fabsf(f));
        float da = sh_act[i] - sh_st[PGTT_S_LAST_ACT + i]; ar += on * (da * da);
      };
      if constexpr (WAVE) {
        const int j = (int)(threadIdx.x & 15);
        joint(j < 12 ? j : 0, j < 12 ? 1.0f : 0.0f);
        sa = row16_sum(sa); lim = row16_sum(lim); pose = row16_sum(pose); s2 = row16_sum(s2); s1 = row16_sum(s1); en = row16_sum(en); ar = row16_sum(ar);
      } else {
#pragma unroll
        for (int i = 0; i < 12; i++) joint(i, 1.0f);
      }
      rew[PGTT_R_STAND_STILL] = sa * (cmd_norm < 0.01f ? 1.f : 0.f);
      rew[PGTT_R_POSE] = pose; rew[PGTT_R_DOF_POS_LIMITS] = lim;
      rew[PGTT_R_TORQUES] = sqrtf(s2) + s1; rew[PGTT_R_ENERGY] = en; rew[PGTT_R_ACTION_RATE] = ar;
    }
    rew[PGTT_R_TERMINATION] = done ? 1.f : 0.f;
    {
      float slip = 0.f, clear = 0.f, perr = 0.f, swing = 0.f, airr = 0.f, con = 0.f, center = 0.f, fh = 0.f, minfoot = INFINITY;
      auto foot = [&](int f, float contact_f, float hmax_f, float phase_f, float air_f, float first_f, float peak_f) {
        float vx = sh_fr[PGTT_F_FEET_VEL + 3 * f], vy = sh_fr[PGTT_F_FEET_VEL + 3 * f + 1];
        float v2 = vx * vx + vy * vy;
        slip += v2 * contact_f;
        float px = sh_fr[PGTT_F_FEET_POS + 3 * f], py = sh_fr[PGTT_F_FEET_POS + 3 * f + 1], pz = sh_fr[PGTT_F_FEET_POS + 3 * f + 2];
        const float clr = baseline ? sh_fr[PGTT_F_FOOT_SITE_Z + f] - (hmax_f - cfg->base_feet_distance + cfg->swing_height)   // joystick.py:569-572
                                   : pz - (hmax_f + cfg->swing_height);                                                  // joystick_pgtt.py:576-578
        clear += fabsf(clr) * sqrtf(sqrtf(v2));
        float rz = gait_get_z(phase_f, hmax_f + cfg->swing_height, cfg->base_feet_distance);
        perr += (pz - rz) * (pz - rz);
        bool swing_mask = phase_f / (float)(2 * M_PI) >= 0.5f;
        swing += ((pz - cfg->swing_height) * (pz - cfg->swing_height)) * (swing_mask ? 1.f : 0.f);
        con += (swing_mask && contact_f != 0.f) ? 1.f : 0.f;
        airr += (air_f - (baseline ? 0.5f : 0.1f)) * first_f;        // joystick.py:591 / joystick_pgtt.py:597
        center += px * px + py * py;
        float er = peak_f / cfg->swing_height - 1.0f;
        fh += (er * er) * first_f;
        minfoot = fminf(minfoot, sh_fr[PGTT_F_FOOT_SITE_Z + f]);
      };
      if constexpr (WAVE) {
        const int f = (int)(threadIdx.x & 3);
        foot(f, sel4(f, contact[0], contact[1], contact[2], contact[3]), sel4(f, hmax[0], hmax[1], hmax[2], hmax[3]),
             sel4(f, phase[0], phase[1], phase[2], phase[3]), sel4(f, air[0], air[1], air[2], air[3]),
             sel4(f, first_contact[0], first_contact[1], first_contact[2], first_contact[3]), sel4(f, peak[0], peak[1], peak[2], peak[3]));
        slip = quad4_sum(slip); clear = quad4_sum(clear); perr = quad4_sum(perr); swing = quad4_sum(swing); airr = quad4_sum(airr);
        con = quad4_sum(con); center = quad4_sum(center); fh = quad4_sum(fh); minfoot = quad4_min(minfoot);
      } else {
#pragma unroll
        for (int f = 0; f < 4; f++) foot(f, contact[f], hmax[f], phase[f], air[f], first_contact[f], peak[f]);
      }
      float moving = cmd_norm > 0.01f ? 1.f : 0.f;
      rew[PGTT_R_FEET_SLIP] = slip * moving; rew[PGTT_R_FEET_CLEARANCE] = clear;
      const float xph = -perr / cfg->phase_sigma;
      if constexpr (WAVE) {
        const int ln = (int)(threadIdx.x & 63);
        const int ex = __float_as_int(expf(ln == 0 ? xlin : (ln == 1 ? xang : xph)));
        rew[PGTT_R_TRACKING_LIN_VEL] = __int_as_float(__builtin_amdgcn_readlane(ex, 0));
        rew[PGTT_R_TRACKING_ANG_VEL] = __int_as_float(__builtin_amdgcn_readlane(ex, 1));
        rew[PGTT_R_FEET_PHASE] = __int_as_float(__builtin_amdgcn_readlane(ex, 2));
      } else {
        rew[PGTT_R_TRACKING_LIN_VEL] = expf(xlin); rew[PGTT_R_TRACKING_ANG_VEL] = expf(xang); rew[PGTT_R_FEET_PHASE] = expf(xph);
      }
      rew[PGTT_R_FEET_SWING] = swing;
      rew[PGTT_R_FEET_AIR_TIME] = airr * moving; rew[PGTT_R_CONTACT] = -con; rew[PGTT_R_CENTER] = center;
      rew[PGTT_R_FEET_HEIGHT] = fh * moving;
      float bh = sh_st[PGTT_S_QPOS + 2] - minfoot - 0.27f;
      rew[PGTT_R_BODY_HEIGHT] = bh * bh;
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < PGTT_NREW; k++) { metrics[k] = rew[k] * cfg->reward_scale[k]; sum += metrics[k]; }
    reward = fminf(fmaxf(sum * dt, 0.f), 10000.f);
    // bookkeeping (joystick_pgtt.py:205-227)
    step_ctr += 1;
#pragma unroll
    for (int f = 0; f < 4; f++) phase[f] = fmod_once(phase[f] + phase_dt, (float)(2 * M_PI));
    timer -= 1;
    if constexpr (WAVE) {
      // the waves that resample (1 - 2 % of a launch) are among the last to leave it: their four counter blocks (command y / z / w, timer)
      // in ONE Philox pass, lanes 0..3, instead of four - same words as rng_uniform(.., stream, i) = word i of block 0 of the stream
      if (done || timer <= 0) {
        const int ln = (int)(threadIdx.x & 63);
        unsigned c0 = id, c1 = ep, c2 = (unsigned)(ln == 0 ? PGTT_RS_CMD_Y : (ln == 1 ? PGTT_RS_CMD_Z : (ln == 2 ? PGTT_RS_CMD_W : PGTT_RS_TIMER))), c3 = 0u;
        philox4x32_10((unsigned)a.seed, (unsigned)(a.seed >> 32), c0, c1, c2, c3);
        auto uni = [&](unsigned w, int src) {
          const float u = (float)((unsigned)__builtin_amdgcn_readlane((int)w, src) >> 8) * (1.0f / 16777216.0f);
          return rng_fix == rng_fix ? rng_fix : u;
        };
        if (timer <= 0) {
          const unsigned w[3] = {c0, c1, c2};
#pragma unroll
          for (int i = 0; i < 3; i++) {
            float y = uni(w[i], 0) * (cfg->cmd_u_max[i] - cfg->cmd_u_min[i]) + cfg->cmd_u_min[i];
            float zb = uni(w[i], 1) < cfg->cmd_b[i] ? 1.f : 0.f;
            float wb = uni(w[i], 2) < 0.5f ? 1.f : 0.f;
            cmd[i] = cmd[i] - wb * (cmd[i] - y * zb);
          }
        }
        const double t = -log1p(-(double)uni(c0, 3)) * 5.0;           // exp_timer
        timer = (int)rint(t / (double)dt);
      }
    } else {
      if (timer <= 0) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
          float y = rng_uniform(a.seed, id, ep, PGTT_RS_CMD_Y, i, rng_fix) * (cfg->cmd_u_max[i] - cfg->cmd_u_min[i]) + cfg->cmd_u_min[i];
          float zb = rng_uniform(a.seed, id, ep, PGTT_RS_CMD_Z, i, rng_fix) < cfg->cmd_b[i] ? 1.f : 0.f;
          float wb = rng_uniform(a.seed, id, ep, PGTT_RS_CMD_W, i, rng_fix) < 0.5f ? 1.f : 0.f;
          cmd[i] = cmd[i] - wb * (cmd[i] - y * zb);
        }
      }
      if (done || timer <= 0) timer = exp_timer(a.seed, id, ep, PGTT_RS_TIMER, dt, rng_fix);
    }
    float sp = 0.f;
#pragma unroll
    for (int f = 0; f < 4; f++) {
      float nc = contact[f] != 0.f ? 0.f : 1.f;
      air[f] *= nc; peak[f] *= nc; last_contact[f] = contact[f]; sp += peak[f];
    }
    metrics[PGTT_NREW] = sp / 4;
}

#ifdef PGTT_TIME
// -DPGTT_TIME=<env> builds (tools/gpu_observe_time.py): phase ticks of the observe wave of that env at a.trace[60000 + i]
#define PG_OTICK(i) do { if (OMODE == OBS_STEP && a.trace) { long long t_ = __builtin_readcyclecounter(); if (lane == 0) { if (e == PGTT_TIME) a.trace[60000 + (i)] = (float)(t_ - ot0_); \
    if (blockIdx.x < 512) a.trace[61000 + 8 * blockIdx.x + (i)] = (float)(t_ - ot0_); } } } while (0)        /* [61000 + 8 block + i]: the boundaries of the first 512 waves (inside trace segment 0, clear of the physics records) */
#elif defined(PGTT_OBS_STOP)
// -DPGTT_OBS_STOP builds (tools/gpu_observe_instr.py): the step's observe wave leaves at phase boundary i when the test-hook integer says so
#define PG_OTICK(i) do { if (OMODE == OBS_STEP && a.scan_preset == 100 + (i)) return; } while (0)
#define PG_SCAN_PRESET(a) ((a).scan_preset == 1)
#else
#define PG_OTICK(i) ((void)0)
#endif
#ifndef PG_SCAN_PRESET
#define PG_SCAN_PRESET(a) ((a).scan_preset != 0)
#endif
// ---- observation rows as a table.  Every row of the state observation is (source value [- zmin]) [+ noise] [- offset]; which source, which
// word of the noise draws, which noise scale and which offset is a function of the row number alone.  The kernel used to find them through a
// chain of row-range tests that every lane of every pass walked (the vector ALU is what the four waves of a SIMD share); now lane io reads
// descriptor io - byte offsets into ONE LDS array that holds the state rows, the sensor frame, the scan and the per-env scalars - and the
// pass is four LDS reads and ~20 vector instructions.  Layout of the array (floats):
enum { OL_ST = 0, OL_FR = OL_ST + PGTT_NSTATE, OL_SCAN = OL_FR + PGTT_NFRAME, OL_DRV = OL_SCAN + 128,
       OD_PHASE = 0 /* cos x4, sin x4 */, OD_GAIT = 8, OD_CMD = 9, OD_ZERO = 12, OD_LASTC = 13, OD_AIR = 17,
       OD_SCALE = 21 /* 0, gyro, gravity, joint pos, joint vel, scan */, OD_OFFS = 27 /* 0, key_qpos[7..18] */, OD_END = 40, OL_END = OL_DRV + OD_END };
constexpr int kObsRowSlots = 192, kPrivSlots = 64;
struct ObsRowTab { unsigned row[2][kObsRowSlots]; unsigned short priv[kPrivSlots]; unsigned short quad[64][4]; };
// descriptor: bits 0..10 byte offset of the source, 11..20 byte offset of the noise word in sh_rng, 21..25 byte offset of the scale in
// OD_SCALE (20 = scan: the row is taken relative to zmin), 26..31 byte offset of the offset in OD_OFFS
constexpr unsigned obs_row_desc(int i) {       // i = row in the PGTT layout (joystick_pgtt.py:336-349)
  unsigned src = 0, word = 0, scale = 0, offs = 0;
  if (i < 3) { src = OL_FR + PGTT_F_GYRO + i; word = i; scale = 1; }
  else if (i < 6) { src = OL_FR + PGTT_F_GRAVITY + i - 3; word = 4 + i - 3; scale = 2; }
  else if (i < 18) { src = OL_ST + PGTT_S_QPOS + 7 + i - 6; word = 8 + i - 6; scale = 3; offs = 1 + i - 6; }
  else if (i < 30) { src = OL_ST + PGTT_S_QVEL + 6 + i - 18; word = 20 + i - 18; scale = 4; }
  else if (i < 38) { src = OL_DRV + OD_PHASE + i - 30; }
  else if (i < 38 + PGTT_NSCAN) { src = OL_SCAN + i - 38; word = 32 + i - 38; scale = 5; }
  else if (i == 38 + PGTT_NSCAN) { src = OL_DRV + OD_GAIT; }
  else if (i < 39 + PGTT_NSCAN + 12) { src = OL_ST + PGTT_S_LAST_ACT + i - (39 + PGTT_NSCAN); }
  else if (i < 39 + PGTT_NSCAN + 15) { src = OL_DRV + OD_CMD + i - (51 + PGTT_NSCAN); }
  else { src = OL_DRV + OD_ZERO; }
  return (src * 4u) | ((word * 4u) << 11) | ((scale * 4u) << 21) | ((offs * 4u) << 26);
}
constexpr unsigned short obs_priv_src(int i) {    // the 44 privileged extras (joystick_pgtt.py:355-365)
  int src = OL_DRV + OD_ZERO;
  if (i < 3) src = OL_FR + PGTT_F_LOCAL_LINVEL + i;
  else if (i < 6) src = OL_FR + PGTT_F_ACCEL + i - 3;
  else if (i < 9) src = OL_FR + PGTT_F_GLOBAL_ANGVEL + i - 6;
  else if (i < 21) src = OL_FR + PGTT_F_ACT_FORCE + i - 9;
  else if (i < 25) src = OL_DRV + OD_LASTC + i - 21;
  else if (i < 37) src = OL_FR + PGTT_F_FEET_VEL + i - 25;
  else if (i < 41) src = OL_DRV + OD_AIR + i - 37;
  return (unsigned short)(src * 4);
}
// Quadrant statistics (joystick_pgtt.py:169-190, n = 6 on both axes of the 13 x 9 grid: top / back = rows 0..5 / 7..12, right / left =
// columns 7..8 / 0..5): the 16 lanes of row q of the wave cover quadrant q (0 top right, 1 top left, 2 back right, 3 back left), three
// cells each; a quadrant has 12 or 36 cells, the surplus slots repeat cells of the same quadrant (max and min do not mind).
constexpr unsigned short obs_quad_cell(int lane, int j) {
  const int q = lane >> 4, s = (lane & 15) + 16 * j;
  const bool left = (q & 1) != 0, back = (q & 2) != 0;
  const int ncol = left ? 6 : 2, ncell = 6 * ncol, k = s % ncell;
  const int r = (back ? 7 : 0) + k / ncol, c = (left ? 0 : 7) + k % ncol;
  return (unsigned short)((OL_SCAN + r * PGTT_SCAN_W + c) * 4);
}
constexpr ObsRowTab make_obs_row_tab() {
  ObsRowTab t{};
  for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) t.quad[l][j] = obs_quad_cell(l, j < 3 ? j : 0);
  for (int io = 0; io < kObsRowSlots; io++) {
    t.row[0][io] = obs_row_desc(io);
    // the baseline layout (go2/joystick.py) drops rows 30..37 (phase) and 38 + NSCAN (gait_freq)
    t.row[1][io] = obs_row_desc(io < 30 ? io : (io < 30 + PGTT_NSCAN ? io + 8 : io + 9));
  }
  for (int i = 0; i < kPrivSlots; i++) t.priv[i] = obs_priv_src(i);
  return t;
}
static_assert(OL_END * 4 < 2048 && (8 + (PGTT_NSCAN + 3) / 4) * 16 <= 1024 && PGTT_OBS <= kObsRowSlots && PGTT_PRIV - PGTT_OBS <= kPrivSlots, "descriptor fields");
__device__ const ObsRowTab kObsRowTab = make_obs_row_tab();

template <int OMODE, bool HAS_TERRAIN>
// four waves per SIMD (128 VGPRs): the kernel is latency-bound, a launch lasts as long as the resident waves of a SIMD take in turn
__global__ __launch_bounds__(64, 4) void observe_kernel(KArgs a, const float* __restrict__ action) {
  const int e = xcd_block(blockIdx.x, gridDim.x), lane = threadIdx.x, N = a.N;
#ifdef PGTT_TIME
  const long long ot0_ = __builtin_readcyclecounter();
  const unsigned ow0_ = (unsigned)wall_clock64();
#endif
  if (OMODE != OBS_STEP && OMODE != OBS_STEP_OBS && a.mask && !a.mask[e]) return;
  const PgttModel* __restrict__ m = a.model;
  const PgttConfig* __restrict__ cfg = a.cfg;
  float* __restrict__ S = a.buf.state;
  int* __restrict__ I = a.buf.istate;

  __shared__ float sh_src[OL_END];
  float* const sh_st = sh_src + OL_ST; float* const sh_fr = sh_src + OL_FR; float* const sh_scan = sh_src + OL_SCAN; float* const sh_drv = sh_src + OL_DRV;
  __shared__ float sh_obs[PGTT_OBS + PGTT_PRIV + 2];
  __shared__ float sh_act[12];

  // The cull data of the env's variant do not depend on the state: lane j requests (centre x, y, world-AABB half extents x, y) of boxes j and
  // j + 64 BEFORE the state rows are waited for, so the two round trips overlap.  They come from the compact table (16 contiguous bytes per
  // box: 13 lines per variant) - the same four numbers out of the 80-byte records cost 160 line requests per wave, more than all its rows
  const TerrainBox* __restrict__ boxes = nullptr;
  float4 brec[2];
  if (HAS_TERRAIN) {
    const int v = a.buf.variant ? min(max(a.buf.variant[e], 0), a.T - 1) : 0;      // clamped like in physics_kernel
    boxes = a.terrain + (long)v * a.B;
#pragma unroll
    for (int h = 0; h < 2; h++) brec[h] = a.cull[(long)v * a.B + min(lane + 64 * h, a.B - 1)];
  }
  if (OMODE == OBS_STEP && a.handover_r) {
    // this step's physics launch left qpos, qvel, the motor targets and the sensor frame env-major: two coalesced loads.  Of the other rows
    // the step reads PGTT_S_CMD .. PGTT_NSTATE - 1 without H_max / H_min (formed anew from this step's scan) and the older halves of the two
    // histories - not the warm start either
    const float* __restrict__ Hr = a.handover_r + (long)e * kHandover;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int j = lane + 64 * h;
      if (j < HO_END) {
        const float v = Hr[j];
        if (j < HO_MOTOR) sh_st[j] = v;                     // qpos, qvel: rows 0 .. 36 in the same order
        else if (j < HO_FRAME) sh_st[PGTT_S_MOTOR_TARGETS + j - HO_MOTOR] = v;
        else sh_fr[j - HO_FRAME] = v;
      }
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int r = PGTT_S_CMD + lane + 64 * h;
      const bool need = r < PGTT_NSTATE && !(r >= PGTT_S_HMAX && r < PGTT_S_HMIN + 4) && !(r >= PGTT_S_MOTOR_TARGETS && r < PGTT_S_MOTOR_TARGETS + 12) &&
                        !(r >= PGTT_S_QERR_HIST + 12 && r < PGTT_S_QERR_HIST + 24) && !(r >= PGTT_S_QVEL_HIST + 12 && r < PGTT_S_QVEL_HIST + 24);     // the older half of a history only ever leaves
      if (need) sh_st[r] = S[r * (long)N + e];
    }
  } else {
    for (int r = lane; r < PGTT_NSTATE; r += 64) sh_st[r] = S[r * (long)N + e];
    for (int r = lane; r < PGTT_NFRAME; r += 64) sh_fr[r] = a.buf.frame[r * (long)N + e];
  }
  if (OMODE == OBS_STEP && lane < 12) sh_act[lane] = action[(long)e * 12 + lane];
  // row descriptors of the three passes over the observation and of the privileged extras (constants: requested with the state rows)
  const bool baseline = cfg->method == PGTT_METHOD_BASELINE;
  unsigned rdesc[3]; unsigned psrc;
#pragma unroll
  for (int it = 0; it < 3; it++) rdesc[it] = kObsRowTab.row[baseline ? 1 : 0][lane + 64 * it];
  psrc = kObsRowTab.priv[lane];
  const uint2 qcell = *reinterpret_cast<const uint2*>(kObsRowTab.quad[lane]);      // byte offsets of this lane's three scan cells
  // the running sums this step adds to (rows of this env, touched by this wave only) are requested here, a launch ahead of their use:
  // at the end of the wave nothing is left to hide a round trip behind
  float epm_old = 0.f, ivs_old = 0.f;
  if (OMODE == OBS_STEP && lane < PGTT_NMETRIC + 2) {
    if (a.buf.ep_metrics) epm_old = a.buf.ep_metrics[lane * (long)N + e];
    if (a.buf.interval_sums) ivs_old = a.buf.interval_sums[lane * (long)N + e];
  }
  __syncthreads();
  PG_OTICK(0);

  // ---------------- height scan (heightmap.py:25-67)
  const float bx = sh_st[PGTT_S_QPOS + 0], by = sh_st[PGTT_S_QPOS + 1], bz = sh_st[PGTT_S_QPOS + 2];
  float yaw;
  if (OMODE == OBS_STEP || OMODE == OBS_STEP_OBS || (OMODE == OBS_SCAN_ONLY && a.yaw_override != a.yaw_override)) {
    float qw = sh_st[PGTT_S_QPOS + 3], qx = sh_st[PGTT_S_QPOS + 4], qy = sh_st[PGTT_S_QPOS + 5], qz = sh_st[PGTT_S_QPOS + 6];
    float qn = sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
    qw /= qn; qx /= qn; qy /= qn; qz /= qn;
    yaw = atan2f(2.0f * (qw * qz + qx * qy), 1.0f - 2.0f * (qy * qy + qz * qz));
  } else if (OMODE == OBS_SCAN_ONLY) {
    yaw = a.yaw_override;
  } else {
    yaw = 0.f;
  }
  // ONE sincos evaluation for the yaw (lanes 8..63; every lane then takes it from lane 8) and for the eight phase rows of the observation
  // (lanes 0..7: cos x4, sin x4 of the phases the step starts with - parked in LDS until the rows are formed): the expansion is ~80 vector
  // instructions whatever the number of lanes that want it
  float sy, cy;
  if (OMODE == OBS_SCAN_ONLY || OMODE == OBS_SCAN_LIFT) sincosf(yaw, &sy, &cy);
  else {
    const int f = lane & 3;
    const float ph = OMODE == OBS_RESET ? ((f == 1 || f == 2) ? (float)M_PI : 0.f) : sh_st[PGTT_S_PHASE + f];
    float sn, cs; sincosf(lane < 8 ? ph : yaw, &sn, &cs);
    if (lane < 8) sh_drv[OD_PHASE + lane] = lane < 4 ? cs : sn;
    sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sn), 8)); cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs), 8));
  }
  const float oz = bz + cfg->scan_z_offset;
  V3 org[2]; float hit[2];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    int idx = lane + 64 * h;
    int ii = idx < PGTT_NSCAN ? idx : 0;
    int r = ii / PGTT_SCAN_W, c = ii - r * PGTT_SCAN_W;
    float ox = ((float)(PGTT_SCAN_H - 1) * 0.5f - (float)r) * cfg->scan_dist_x;
    float oy = ((float)(PGTT_SCAN_W - 1) * 0.5f - (float)c) * cfg->scan_dist_y;
    float wx = ox * cy + oy * (-sy), wy = ox * sy + oy * cy;
    org[h] = v3(bx + wx, by + wy, oz);
    if (r == (PGTT_SCAN_H - 1) / 2 && c == (PGTT_SCAN_W - 1) / 2) org[h] = v3(bx, by, oz);
    // plane z=0 (normal +z): x = -pnt_z / vec_z = pnt_z, valid if x >= 0
    float x = oz;
    hit[h] = x >= 0.f ? x : INFINITY;
  }
  if (HAS_TERRAIN) {
    // cull, lane-parallel: lane j looks at boxes j and j + 64: world AABB of the box against the scan footprint, a
    // rectangle of half-sides (hx, hy) turned by the yaw - the four separating axes of a rectangle / AABB pair in the
    // plane, with 1 mm of slack (a vertical ray can only hit a box whose footprint contains it, so dropping the boxes
    // that do not overlap the rectangle changes no hit).  min() over the hits does not depend on the visiting order.
    const float hx = 0.5f * (PGTT_SCAN_H - 1) * fabsf(cfg->scan_dist_x) + 1e-3f, hy = 0.5f * (PGTT_SCAN_W - 1) * fabsf(cfg->scan_dist_y) + 1e-3f;
    const float acy = fabsf(cy), asy = fabsf(sy);
    // The survivors are COMPACTED into LDS (slot = number of surviving boxes before this one), prepared by the lane that owns them, and the ray
    // loop reads them from there through wave-uniform addresses, the next one requested while the current one is tested: no scalar-memory
    // round trip and no per-box arithmetic in the loop (a wave used to wait ~0.3 us for every 80-byte record; the slowest waves of a launch
    // are those that stand among many boxes).  kScanSlots survivors fit; further ones (never on the shipped terrains) are read from the table.
    constexpr int kScanSlots = 40;
    __shared__ float4 sh_box[kScanSlots * 3];
    unsigned long long todo[2];
    int nsurv = 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int b = lane + 64 * h;
      bool reach = false;
      if (b < a.B) {
        const float4 C = brec[h];            // centre x, y; world-AABB half extents x, y
        const float dx = C.x - bx, dy = C.y - by;
        reach = (fabsf(dx) <= C.z + hx * acy + hy * asy) & (fabsf(dy) <= C.w + hx * asy + hy * acy) &
                (fabsf(dx * cy + dy * sy) <= hx + C.z * acy + C.w * asy) & (fabsf(dy * cy - dx * sy) <= hy + C.z * asy + C.w * acy);
      }
      const unsigned long long mk = __ballot(reach);
      const int slot = nsurv + __popcll(mk & ((1ull << lane) - 1ull));
      if (reach && slot < kScanSlots) {
        // The lane that owns a surviving box prepares it for the ray loop - all survivors at once, one per lane.  A box turned about z only
        // (m20 = m21 = m02 = m12 = 0 exactly; every shipped / generated terrain) has lp2 = 0 * relx + 0 * rely + m22 * relz = round(m22 * relz)
        // with relz = oz - pz the same for every ray of the env: the parameters of its two z faces and their validity are properties of the
        // BOX (`hz` below, the value ray_box_down's two-face form returns for a ray inside the footprint); a ray only decides "inside or not".
        // Such a box goes to LDS as the 9 numbers that test needs; any other box as its index (the loop reads its record from the table).
        const float4 q0 = reinterpret_cast<const float4*>(boxes + b)[0], q1 = reinterpret_cast<const float4*>(boxes + b)[1], q2 = reinterpret_cast<const float4*>(boxes + b)[2],
                     q3 = reinterpret_cast<const float4*>(boxes + b)[3];
        const float m22 = q3.w;
        const bool fast = (q3.y == 0.f) & (q3.z == 0.f) & (m22 != 0.f) & (q2.y == 0.f) & (q3.x == 0.f);      // m20, m21, m22, m02, m12
        const float il2 = __builtin_amdgcn_rcpf(-m22);
        const float lp2 = mul_unfused(m22, oz - q0.z);       // a product on its own (the generic form adds it to two exact zeros): must not be fused into sz - lp2
        const float xt = (q1.z - lp2) * il2, xb = (-q1.z - lp2) * il2;
        float hz = xt >= 0.f ? xt : INFINITY;
        hz = ((xb >= 0.f) & (xb < hz)) ? xb : hz;
        sh_box[slot * 3 + 0] = make_float4(q0.x, q0.y, q1.w, q2.z);                    // px, py, m00, m10
        sh_box[slot * 3 + 1] = make_float4(q2.x, q2.w, q1.x, q1.y);                    // m01, m11, sx, sy
        sh_box[slot * 3 + 2] = make_float4(hz, fast ? 1.f : 0.f, __int_as_float(b), 0.f);
      }
      // boxes beyond the slots: the highest set bits of this half (slot order = bit order)
      int over = nsurv + __popcll(mk) - kScanSlots;
      unsigned long long rest = 0ull, t = mk;
      while (over > 0 && t != 0ull) { const int hb = 63 - __builtin_clzll(t); rest |= 1ull << hb; t &= ~(1ull << hb); over--; }
      todo[h] = rest;
      nsurv += __popcll(mk);
    }
    __syncthreads();
    const int nslot = min(nsurv, kScanSlots);
    auto load_slot = [&](int i, float4 (&r)[3]) {
#pragma unroll
      for (int q = 0; q < 3; q++) r[q] = sh_box[i * 3 + q];
    };
    auto test_box = [&](const float4 (&r)[3]) {
      if (__builtin_amdgcn_readfirstlane(__float_as_int(r[2].y)) != 0) {        // the record is the same in every lane: a scalar branch
        const float hz = r[2].x;
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const float rx = org[k].x - r[0].x, ry = org[k].y - r[0].y;
          const float lp0 = r[0].z * rx + r[0].w * ry, lp1 = r[1].x * rx + r[1].y * ry;      // + m20 * relz = + (+-0): only the sign of a zero, and |lp| is what is tested
          const bool inside = (fabsf(lp0) <= r[1].z) & (fabsf(lp1) <= r[1].w);
          hit[k] = fminf(hit[k], inside ? hz : INFINITY);
        }
        return;
      }
      const TerrainBox tb = boxes[__builtin_amdgcn_readfirstlane(__float_as_int(r[2].z))];
      const RayBox rb = ray_box_prepare(tb);
#pragma unroll
      for (int k = 0; k < 2; k++) hit[k] = fminf(hit[k], ray_box_down(tb, rb, org[k]));
    };
    float4 cur[3], nxt[3];
    if (nslot > 0) load_slot(0, cur);
    for (int i = 0; i < nslot; i++) {
      if (i + 1 < nslot) load_slot(i + 1, nxt);
      test_box(cur);
#pragma unroll
      for (int q = 0; q < 3; q++) cur[q] = nxt[q];
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
      unsigned long long mk = todo[h];
      while (mk != 0ull) {
        const int b = 64 * h + __builtin_ctzll(mk);
        mk &= mk - 1ull;
        TerrainBox tb = boxes[b];
        const RayBox rb = ray_box_prepare(tb);
#pragma unroll
        for (int k = 0; k < 2; k++) hit[k] = fminf(hit[k], ray_box_down(tb, rb, org[k]));
      }
    }
  }
  PG_OTICK(1);
  float z[2];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    float dist = hit[h] == INFINITY ? -1.0f : hit[h];
    z[h] = org[h].z + (-1.0f) * dist;
    int idx = lane + 64 * h;
    if ((OMODE == OBS_STEP || OMODE == OBS_STEP_OBS) && PG_SCAN_PRESET(a)) z[h] = a.buf.scan_z[(long)e * PGTT_NSCAN + (idx < PGTT_NSCAN ? idx : 0)];   // test hook
    if (idx < PGTT_NSCAN) { sh_scan[idx] = z[h]; a.buf.scan_z[(long)e * PGTT_NSCAN + idx] = z[h]; }
  }
  if (OMODE == OBS_SCAN_ONLY) return;
  const bool v1 = lane + 64 < PGTT_NSCAN;
  if (OMODE == OBS_SCAN_LIFT) {
    float zmax = wave_max(fmaxf(z[0], v1 ? z[1] : -INFINITY));
    if (lane == 0) S[(PGTT_S_QPOS + 2) * (long)N + e] = bz + zmax;
    return;
  }
  // ---------------- quadrant statistics (joystick_pgtt.py:169-190): n = 6 on both axes of the 13x9 grid
  float qmax[4], qmin[4];
  {
    __syncthreads();                     // the scan is in LDS
    const char* const srcb = reinterpret_cast<const char*>(sh_src);
    const float c0 = *reinterpret_cast<const float*>(srcb + (qcell.x & 0xffffu)), c1 = *reinterpret_cast<const float*>(srcb + (qcell.x >> 16)),
                c2 = *reinterpret_cast<const float*>(srcb + (qcell.y & 0xffffu));
    float mx = fmaxf(fmaxf(c0, c1), c2), mn = fminf(fminf(c0, c1), c2);
    mx = fmaxf(mx, dpp_f<0xB1>(mx)); mx = fmaxf(mx, dpp_f<0x4E>(mx)); mx = fmaxf(mx, dpp_f<0x124>(mx)); mx = fmaxf(mx, dpp_f<0x128>(mx));
    mn = fminf(mn, dpp_f<0xB1>(mn)); mn = fminf(mn, dpp_f<0x4E>(mn)); mn = fminf(mn, dpp_f<0x124>(mn)); mn = fminf(mn, dpp_f<0x128>(mn));
#pragma unroll
    for (int k = 0; k < 4; k++) {
      qmax[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mx), 16 * k));
      qmin[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mn), 16 * k));
    }
  }
  const float zmin = wave_min(fminf(z[0], v1 ? z[1] : INFINITY));

  PG_OTICK(2);
  // ---------------- per-env scalars (computed redundantly by every lane from LDS)
  // method 1 = the baseline task go2/joystick.py: no phase / gait_freq rows in the observation (162 / 206 instead of
  // 171 / 215), H_max = quadrant max, world-frame clearance target, 0.5 s air-time threshold
  const int OBSD = baseline ? PGTT_OBS_BASELINE : PGTT_OBS, PRIVD = OBSD + (PGTT_PRIV - PGTT_OBS);
  const unsigned id = (unsigned)(a.env_off + e);
  const unsigned ep = (unsigned)I[PGTT_I_RNG_CTR * (long)N + e];
  int step_ctr = I[PGTT_I_STEP * (long)N + e];
  int timer = I[PGTT_I_STEPS_UNTIL_CMD * (long)N + e];
  int ep_steps = I[PGTT_I_EP_STEPS * (long)N + e];
  const float dt = cfg->ctrl_dt;
  float cmd[3], phase[4], air[4], peak[4], hmax[4], hmin[4], last_contact[4], contact[4], first_contact[4];
  float gait_freq, phase_dt;
  bool prev_done = false;
  if (OMODE == OBS_RESET) {
#pragma unroll
    for (int i = 0; i < 3; i++)
      cmd[i] = rng_uniform(a.seed, id, ep, PGTT_RS_RESET_CMD, i, a.rng_fix) * (cfg->cmd_u_max[i] - cfg->cmd_u_min[i]) + cfg->cmd_u_min[i];
    gait_freq = rng_uniform(a.seed, id, ep, PGTT_RS_RESET_FREQ, 0, a.rng_fix) * (cfg->gait_freq[1] - cfg->gait_freq[0]) + cfg->gait_freq[0];
    phase_dt = (float)(2 * M_PI) * dt * gait_freq;
    phase[0] = 0.f; phase[1] = (float)M_PI; phase[2] = (float)M_PI; phase[3] = 0.f;
#pragma unroll
    for (int f = 0; f < 4; f++) { air[f] = 0.f; peak[f] = 0.f; hmax[f] = 0.1f; hmin[f] = 0.f; last_contact[f] = 0.f; contact[f] = 0.f; first_contact[f] = 0.f; }
    timer = exp_timer(a.seed, id, ep, PGTT_RS_RESET_TIMER, dt, a.rng_fix);
    step_ctr = 0; ep_steps = 0;
    __syncthreads();
    // info arrays that _get_obs reads
    if (lane < 12) { sh_st[PGTT_S_LAST_ACT + lane] = 0.f; sh_st[PGTT_S_LAST_LAST_ACT + lane] = 0.f; sh_st[PGTT_S_MOTOR_TARGETS + lane] = 0.f; }
    if (lane < 24) { sh_st[PGTT_S_QERR_HIST + lane] = 0.f; sh_st[PGTT_S_QVEL_HIST + lane] = 0.f; }
    __syncthreads();
  } else {
    prev_done = cfg->autoreset && a.buf.done[e] != 0.f;
    if (prev_done) ep_steps = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) cmd[i] = sh_st[PGTT_S_CMD + i];
    gait_freq = sh_st[PGTT_S_GAIT_FREQ]; phase_dt = sh_st[PGTT_S_PHASE_DT];
#pragma unroll
    for (int f = 0; f < 4; f++) {
      phase[f] = sh_st[PGTT_S_PHASE + f];
      last_contact[f] = sh_st[PGTT_S_LAST_CONTACT + f];
      contact[f] = sh_fr[PGTT_F_CONTACT + f];
      bool filt = (contact[f] != 0.f) || (last_contact[f] != 0.f);
      first_contact[f] = (sh_st[PGTT_S_AIR_TIME + f] > 0.f ? 1.f : 0.f) * (filt ? 1.f : 0.f);
      air[f] = sh_st[PGTT_S_AIR_TIME + f] + dt;
      peak[f] = fmaxf(sh_st[PGTT_S_SWING_PEAK + f], sh_fr[PGTT_F_FEET_POS + 3 * f + 2]);
      hmax[f] = baseline ? qmax[f] : qmax[f] - qmin[f]; hmin[f] = qmin[f];     // joystick.py:186 vs joystick_pgtt.py:189
    }
  }

  PG_OTICK(3);
  // ---------------- observation rows in LDS (joystick_pgtt.py:336-365)
  const float lvl = cfg->noise_level;
  // The noise draws of all rows in ONE Philox pass: the 147 noisy rows need 38 counter blocks (gyro 1, gravity 1, joint
  // positions 3, joint velocities 3, scan 30; rng_uniform(.., stream, idx) is word idx & 3 of block idx >> 2), lane j forms
  // block j and parks its four words in LDS - same draws as one rng_uniform call per row, a third of the multiplies.
  __shared__ unsigned sh_rng[64 * 4];
  static_assert(8 + (PGTT_NSCAN + 3) / 4 <= 64, "one lane per Philox block");
  {
    const int stream = lane == 0 ? PGTT_RS_GYRO : (lane == 1 ? PGTT_RS_GRAVITY : (lane < 5 ? PGTT_RS_QPOS : (lane < 8 ? PGTT_RS_QVEL : PGTT_RS_SCAN)));
    const int blk = lane < 2 ? 0 : (lane < 5 ? lane - 2 : (lane < 8 ? lane - 5 : lane - 8));
    unsigned c0 = id, c1 = ep, c2 = (unsigned)stream, c3 = (unsigned)blk;
    philox4x32_10((unsigned)a.seed, (unsigned)(a.seed >> 32), c0, c1, c2, c3);
    sh_rng[4 * lane + 0] = c0; sh_rng[4 * lane + 1] = c1; sh_rng[4 * lane + 2] = c2; sh_rng[4 * lane + 3] = c3;
  }
  // the per-env scalars the rows draw on, next to the rows' other sources (the phase rows are there since the yaw was formed): gait frequency, command, last contact, air time, the five noise scales and the twelve joint offsets
  if (lane < 12) sh_drv[OD_OFFS + 1 + lane] = m->key_qpos[7 + lane];
  if (lane == 0) {
    sh_drv[OD_GAIT] = gait_freq; sh_drv[OD_ZERO] = 0.f; sh_drv[OD_OFFS] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) sh_drv[OD_CMD + k] = cmd[k];
#pragma unroll
    for (int f = 0; f < 4; f++) { sh_drv[OD_LASTC + f] = last_contact[f]; sh_drv[OD_AIR + f] = air[f]; }
    sh_drv[OD_SCALE + 0] = 0.f; sh_drv[OD_SCALE + 1] = cfg->noise_gyro; sh_drv[OD_SCALE + 2] = cfg->noise_gravity;
    sh_drv[OD_SCALE + 3] = cfg->noise_joint_pos; sh_drv[OD_SCALE + 4] = cfg->noise_joint_vel; sh_drv[OD_SCALE + 5] = cfg->noise_heightscan;
  }
  __syncthreads();
  const char* const srcb = reinterpret_cast<const char*>(sh_src);
#pragma unroll
  for (int it = 0; it < 3; it++) {
    const int io = lane + 64 * it;
    if (io >= OBSD) continue;
    // every row is (base + noise) - offset with noise = (2u - 1) * level * scale (rows without noise read a word too, with scale 0; rows
    // without an offset subtract the table's zero: x - 0 is x)
    const unsigned d = rdesc[it], so = (d >> 21) & 0x1fu;
    const float base0 = *reinterpret_cast<const float*>(srcb + (d & 0x7ffu));
    const unsigned word = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(sh_rng) + ((d >> 11) & 0x3ffu));
    const float scale = *reinterpret_cast<const float*>(srcb + (OL_DRV + OD_SCALE) * 4 + so);
    const float offs = *reinterpret_cast<const float*>(srcb + (OL_DRV + OD_OFFS) * 4 + (d >> 26));
    const float base = base0 - (so == 20u ? zmin : 0.f);              // scan rows are heights above the lowest scan point
    const float uw = (float)(word >> 8) * (1.0f / 16777216.0f);
    const float u = a.rng_fix == a.rng_fix ? a.rng_fix : uw;
    const float noisy = scale != 0.f ? base + (2.f * u - 1.f) * lvl * scale : base;
    sh_obs[io] = noisy - offs;
  }
  if (lane < PGTT_PRIV - PGTT_OBS) sh_obs[OBSD + OBSD + lane] = *reinterpret_cast<const float*>(srcb + psrc);
  // history buffers (joystick_pgtt.py:319-334): motor_targets in sh_st was written by the physics kernel
  float hist_q = 0.f, hist_v = 0.f; const bool upd = (step_ctr % cfg->history_update_steps) == 0;
  if (lane < 24) {
    if (upd) {
      hist_v = lane < 12 ? sh_st[PGTT_S_QVEL + 6 + lane] : sh_st[PGTT_S_QVEL_HIST + lane - 12];
      hist_q = lane < 12 ? sh_st[PGTT_S_QPOS + 7 + lane] - sh_st[PGTT_S_MOTOR_TARGETS + lane] : sh_st[PGTT_S_QERR_HIST + lane - 12];
    } else { hist_v = sh_st[PGTT_S_QVEL_HIST + lane]; hist_q = sh_st[PGTT_S_QERR_HIST + lane]; }
  }
  __syncthreads();
  for (int i = lane; i < OBSD; i += 64) sh_obs[OBSD + i] = sh_obs[i];     // privileged = state || extras
  __syncthreads();
  if (OMODE == OBS_STEP_OBS) {
    // scan + observation half: H_max / H_min of this scan for task_kernel, the observation rows (a finished episode's
    // rows are replaced by task_kernel), nothing else
    if (lane < 4) {
      S[(PGTT_S_HMAX + lane) * (long)N + e] = sel4(lane, hmax[0], hmax[1], hmax[2], hmax[3]);
      S[(PGTT_S_HMIN + lane) * (long)N + e] = sel4(lane, hmin[0], hmin[1], hmin[2], hmin[3]);
    }
    for (int i = lane; i < OBSD; i += 64) a.buf.obs_state[(long)e * OBSD + i] = sh_obs[i];
    for (int i = lane; i < PRIVD; i += 64) a.buf.obs_priv[(long)e * PRIVD + i] = sh_obs[OBSD + i];
    return;
  }

  PG_OTICK(4);
  // ---------------- rewards, termination, bookkeeping
  float reward = 0.f; bool done = false; float metrics[PGTT_NMETRIC];
#pragma unroll
  for (int k = 0; k < PGTT_NMETRIC; k++) metrics[k] = 0.f;
  float act_i = 0.f;
  if (OMODE == OBS_STEP) {
    TaskScalars t;
#pragma unroll
    for (int i = 0; i < 3; i++) t.cmd[i] = cmd[i];
#pragma unroll
    for (int f = 0; f < 4; f++) { t.phase[f] = phase[f]; t.air[f] = air[f]; t.peak[f] = peak[f]; t.hmax[f] = hmax[f]; t.last_contact[f] = last_contact[f]; t.contact[f] = contact[f]; t.first_contact[f] = first_contact[f]; }
    t.phase_dt = phase_dt; t.step_ctr = step_ctr; t.timer = timer;
    task_rewards<true>(sh_st, sh_fr, sh_act, cfg, m, baseline, a.seed, id, ep, dt, a.rng_fix, t);
#pragma unroll
    for (int i = 0; i < 3; i++) cmd[i] = t.cmd[i];
#pragma unroll
    for (int f = 0; f < 4; f++) { phase[f] = t.phase[f]; air[f] = t.air[f]; peak[f] = t.peak[f]; last_contact[f] = t.last_contact[f]; }
    step_ctr = t.step_ctr; timer = t.timer; done = t.done; reward = t.reward;
#pragma unroll
    for (int k = 0; k < PGTT_NMETRIC; k++) metrics[k] = t.metrics[k];
    if (lane < 12) act_i = sh_act[lane];
  }
  PG_OTICK(5);
  // ---------------- Episode / AutoReset wrapper semantics (UPSTREAM-RECALL)
  bool wdone = done;
  if (OMODE == OBS_STEP && cfg->autoreset) {
    ep_steps += 1;
    if (ep_steps >= cfg->episode_length) wdone = true;
  }
  // Every value below is the same in all lanes (a per-env scalar in a vector register).  Lane 0 parks them in LDS - the new values of the
  // env's state rows in their slots of the row image sh_st, the metrics with (reward, 1) behind them - and the wave then stores ROW RANGES
  // with one address per lane; picking "my row's value" out of registers costs a compare + select per candidate (22 for a metric) and a
  // 64-bit address per store site, in a kernel whose four waves per SIMD share the vector ALU.
  // AutoReset of a finished episode: the first state and its observation are requested HERE, all at once (the copy loops at the end of the
  // wave were eight dependent round trips - each store could alias the next load - and the few waves with a finished episode were the last
  // of the launch to leave)
  const bool restore = OMODE == OBS_STEP && cfg->autoreset && wdone && a.buf.first_state && a.buf.first_obs;
  constexpr int kFirstObsPasses = (PGTT_OBS + PGTT_PRIV + 63) / 64;
  float first_st = 0.f, first_ob[kFirstObsPasses];
  if (restore) {
    if (lane < PGTT_S_CMD) first_st = a.buf.first_state[lane * (long)N + e];
    const float* __restrict__ fo = a.buf.first_obs + (long)e * (OBSD + PRIVD);
#pragma unroll
    for (int i = 0; i < kFirstObsPasses; i++) first_ob[i] = lane + 64 * i < OBSD + PRIVD ? fo[lane + 64 * i] : 0.f;
  }
  __shared__ float sh_met[PGTT_NMETRIC + 2];
  __syncthreads();                       // the row image has been read for the last time (rewards, history)
  if (lane < 12) {
    const float prev = sh_st[PGTT_S_LAST_ACT + lane];
    sh_st[PGTT_S_LAST_LAST_ACT + lane] = OMODE == OBS_STEP ? prev : 0.f;
    sh_st[PGTT_S_LAST_ACT + lane] = OMODE == OBS_STEP ? act_i : 0.f;
    if (OMODE != OBS_STEP) sh_st[PGTT_S_MOTOR_TARGETS + lane] = 0.f;
  }
  if (lane < 24) { sh_st[PGTT_S_QVEL_HIST + lane] = hist_v; sh_st[PGTT_S_QERR_HIST + lane] = hist_q; }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 3; i++) sh_st[PGTT_S_CMD + i] = cmd[i];
#pragma unroll
    for (int f = 0; f < 4; f++) {
      sh_st[PGTT_S_PHASE + f] = phase[f]; sh_st[PGTT_S_AIR_TIME + f] = air[f]; sh_st[PGTT_S_SWING_PEAK + f] = peak[f];
      sh_st[PGTT_S_HMAX + f] = hmax[f]; sh_st[PGTT_S_HMIN + f] = hmin[f]; sh_st[PGTT_S_LAST_CONTACT + f] = last_contact[f];
    }
    sh_st[PGTT_S_PHASE_DT] = phase_dt; sh_st[PGTT_S_GAIT_FREQ] = gait_freq;
#pragma unroll
    for (int k = 0; k < PGTT_NMETRIC; k++) sh_met[k] = metrics[k];
    sh_met[PGTT_NMETRIC] = reward; sh_met[PGTT_NMETRIC + 1] = 1.0f;
  }
  __syncthreads();
  if (OMODE == OBS_STEP && cfg->autoreset && a.buf.ep_metrics && lane < PGTT_NMETRIC + 2)
    a.buf.ep_metrics[lane * (long)N + e] = (epm_old + sh_met[lane]) * (prev_done ? 0.f : 1.f);

  PG_OTICK(6);
  // ---------------- stores: rows PGTT_S_CMD .. PGTT_NSTATE - 1 of the image (the step leaves the motor targets, which are the physics kernel's, alone)
  static_assert(PGTT_NSTATE - PGTT_S_CMD <= 128 && PGTT_NMETRIC + 2 <= 64, "two passes over the rows, one over the metrics");
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int r = PGTT_S_CMD + lane + 64 * h;
    // the step leaves the histories alone except on the steps that shift them (1 in history_update_steps)
    const bool mine = r < PGTT_NSTATE && (OMODE != OBS_STEP || r < PGTT_S_MOTOR_TARGETS || r >= PGTT_S_LAST_CONTACT || (upd && r >= PGTT_S_QERR_HIST));
    if (mine) S[r * (long)N + e] = sh_st[r];
  }
  if (lane == 0) {
    I[PGTT_I_STEP * (long)N + e] = step_ctr; I[PGTT_I_STEPS_UNTIL_CMD * (long)N + e] = timer;
    I[PGTT_I_RNG_CTR * (long)N + e] = (int)(ep + 1u); I[PGTT_I_EP_STEPS * (long)N + e] = ep_steps;
    a.buf.reward[e] = reward; a.buf.done[e] = wdone ? 1.f : 0.f;
  }
  if (lane < PGTT_NMETRIC) {
    const float v = sh_met[lane];
    a.buf.metrics[lane * (long)N + e] = v;
  }
  if (OMODE == OBS_STEP && a.buf.interval_sums && lane < PGTT_NMETRIC + 2)
    a.buf.interval_sums[lane * (long)N + e] = ivs_old + (lane < PGTT_NMETRIC ? sh_met[lane] : (lane == PGTT_NMETRIC ? reward : (wdone ? 1.f : 0.f)));
  if (restore) {                         // the first observation takes the place of this step's in LDS (same layout: state rows, then privileged rows)
    static_assert(PGTT_S_CMD <= 64, "one lane per restored state row");
    if (lane < PGTT_S_CMD) S[lane * (long)N + e] = first_st;
#pragma unroll
    for (int i = 0; i < kFirstObsPasses; i++) if (lane + 64 * i < OBSD + PRIVD) sh_obs[lane + 64 * i] = first_ob[i];
    __syncthreads();
  }
  for (int i = lane; i < OBSD; i += 64) a.buf.obs_state[(long)e * OBSD + i] = sh_obs[i];
  for (int i = lane; i < PRIVD; i += 64) a.buf.obs_priv[(long)e * PRIVD + i] = sh_obs[OBSD + i];
  PG_OTICK(7);
#ifdef PGTT_TIME
  if (OMODE == OBS_STEP && a.trace && lane == 0 && blockIdx.x < 4096) {      // placement and timeline of every observe wave
    unsigned* tw = (unsigned*)(a.trace + 40960 + 4 * blockIdx.x);
    tw[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4); tw[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20); tw[2] = ow0_; tw[3] = (unsigned)wall_clock64();
  }
#endif
  if (OMODE == OBS_RESET) {
    if (a.buf.first_state) for (int r = lane; r < PGTT_S_CMD; r += 64) a.buf.first_state[r * (long)N + e] = sh_st[r];
    if (a.buf.first_obs) for (int i = lane; i < OBSD + PRIVD; i += 64) a.buf.first_obs[(long)e * (OBSD + PRIVD) + i] = sh_obs[i];
    if (a.buf.ep_metrics) for (int k = lane; k < PGTT_NMETRIC + 2; k += 64) a.buf.ep_metrics[k * (long)N + e] = 0.f;
  }
}


// ------------------------------------------------------------------ interval reduction: one block per row of interval_sums
// out[k] (+)= sum over the envs of row k, and the row is cleared: read, zero and reduce in one pass (a GEMV, a fill and an add otherwise);
// block `rows` only hands the caller's env-step count over
template <int UNUSED>
__global__ __launch_bounds__(256) void interval_reduce_kernel(float* __restrict__ sums, int N, int rows, float* __restrict__ acc, float env_steps, int accumulate) {
  if ((int)blockIdx.x == rows) { if (threadIdx.x == 0) acc[rows] = (accumulate ? acc[rows] : 0.f) + env_steps; return; }
  float* __restrict__ row = sums + (long)blockIdx.x * N;
  float s = 0.f;
  if ((N & 3) == 0) {
    float4* __restrict__ row4 = reinterpret_cast<float4*>(row);
    for (int i = threadIdx.x; i < (N >> 2); i += 256) { const float4 v = row4[i]; row4[i] = make_float4(0.f, 0.f, 0.f, 0.f); s += (v.x + v.y) + (v.z + v.w); }
  } else {
    for (int i = threadIdx.x; i < N; i += 256) { s += row[i]; row[i] = 0.f; }
  }
  s = row16_sum(s);                                      // the 16 lanes of a DPP row, then the rows and the waves through LDS
  __shared__ float part[16];
  if ((threadIdx.x & 15) == 0) part[threadIdx.x >> 4] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) t += part[i];
    acc[blockIdx.x] = (accumulate ? acc[blockIdx.x] : 0.f) + t;
  }
}

// ------------------------------------------------------------------ task: one env per LANE (coalesced SoA rows)
// The per-env scalar half of a control step: contact bookkeeping, the 21 rewards, termination, command resampling,
// history buffers, Episode / AutoReset wrapper semantics.  In the fused observe kernel every wave repeats this ~2.5 k
// instruction stream for ONE env; here a wave does it for 64.  Runs after observe_kernel<OBS_STEP_OBS>, which left
// H_max / H_min of the current scan in the state rows and wrote the observation rows.
template <int UNUSED>
__global__ __launch_bounds__(64) void task_kernel(KArgs a, const float* __restrict__ action) {
  const int N = a.N;
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= N) return;
  const PgttModel* __restrict__ m = a.model;
  const PgttConfig* __restrict__ cfg = a.cfg;
  float* __restrict__ S = a.buf.state;
  int* __restrict__ I = a.buf.istate;
  const float* __restrict__ Fr = a.buf.frame;
  // rows this half reads, as per-lane arrays with compile-time indices (registers)
  float st[PGTT_NSTATE], fr[PGTT_NFRAME], act[12];
#pragma unroll
  for (int r = 0; r < PGTT_S_QWARM; r++) st[r] = S[r * (long)N + e];                      // qpos, qvel
#pragma unroll
  for (int r = PGTT_S_CMD; r < PGTT_NSTATE; r++) st[r] = S[r * (long)N + e];              // task rows
#pragma unroll
  for (int r = 0; r < PGTT_NFRAME; r++) fr[r] = Fr[r * (long)N + e];
#pragma unroll
  for (int i = 0; i < 12; i++) act[i] = action[(long)e * 12 + i];
  const bool baseline = cfg->method == PGTT_METHOD_BASELINE;
  const int OBSD = baseline ? PGTT_OBS_BASELINE : PGTT_OBS, PRIVD = OBSD + (PGTT_PRIV - PGTT_OBS);
  const unsigned id = (unsigned)(a.env_off + e);
  const unsigned ep = (unsigned)I[PGTT_I_RNG_CTR * (long)N + e];
  int ep_steps = I[PGTT_I_EP_STEPS * (long)N + e];
  const float dt = cfg->ctrl_dt;
  const bool prev_done = cfg->autoreset && a.buf.done[e] != 0.f;
  if (prev_done) ep_steps = 0;
  TaskScalars t;
  t.step_ctr = I[PGTT_I_STEP * (long)N + e]; t.timer = I[PGTT_I_STEPS_UNTIL_CMD * (long)N + e];
#pragma unroll
  for (int i = 0; i < 3; i++) t.cmd[i] = st[PGTT_S_CMD + i];
  t.phase_dt = st[PGTT_S_PHASE_DT];
#pragma unroll
  for (int f = 0; f < 4; f++) {
    t.phase[f] = st[PGTT_S_PHASE + f];
    t.last_contact[f] = st[PGTT_S_LAST_CONTACT + f];
    t.contact[f] = fr[PGTT_F_CONTACT + f];
    const bool filt = (t.contact[f] != 0.f) || (t.last_contact[f] != 0.f);
    t.first_contact[f] = (st[PGTT_S_AIR_TIME + f] > 0.f ? 1.f : 0.f) * (filt ? 1.f : 0.f);
    t.air[f] = st[PGTT_S_AIR_TIME + f] + dt;
    t.peak[f] = fmaxf(st[PGTT_S_SWING_PEAK + f], fr[PGTT_F_FEET_POS + 3 * f + 2]);
    t.hmax[f] = st[PGTT_S_HMAX + f];                                                     // of the current scan
  }
  // history buffers (joystick_pgtt.py:319-334) use the step counter BEFORE it advances
  float hist_q[24], hist_v[24];
  const bool upd = (t.step_ctr % cfg->history_update_steps) == 0;
#pragma unroll
  for (int i = 0; i < 24; i++) {
    const float nv = i < 12 ? st[PGTT_S_QVEL + 6 + (i < 12 ? i : 0)] : st[PGTT_S_QVEL_HIST + (i >= 12 ? i - 12 : 0)];
    const float nq = i < 12 ? st[PGTT_S_QPOS + 7 + (i < 12 ? i : 0)] - st[PGTT_S_MOTOR_TARGETS + (i < 12 ? i : 0)] : st[PGTT_S_QERR_HIST + (i >= 12 ? i - 12 : 0)];
    hist_v[i] = upd ? nv : st[PGTT_S_QVEL_HIST + i];
    hist_q[i] = upd ? nq : st[PGTT_S_QERR_HIST + i];
  }
  task_rewards<false>(st, fr, act, cfg, m, baseline, a.seed, id, ep, dt, a.rng_fix, t);
  // Episode / AutoReset wrapper semantics
  bool wdone = t.done;
  if (cfg->autoreset) {
    ep_steps += 1;
    if (ep_steps >= cfg->episode_length) wdone = true;
    if (a.buf.ep_metrics) {
      const float keep = prev_done ? 0.f : 1.f;
#pragma unroll
      for (int k = 0; k < PGTT_NMETRIC + 2; k++) {
        const float add = k < PGTT_NMETRIC ? t.metrics[k < PGTT_NMETRIC ? k : 0] : (k == PGTT_NMETRIC ? t.reward : 1.0f);
        float* p = a.buf.ep_metrics + k * (long)N + e;
        *p = (*p + add) * keep;
      }
    }
  }
  // stores
#pragma unroll
  for (int i = 0; i < 3; i++) S[(PGTT_S_CMD + i) * (long)N + e] = t.cmd[i];
#pragma unroll
  for (int f = 0; f < 4; f++) {
    S[(PGTT_S_PHASE + f) * (long)N + e] = t.phase[f];
    S[(PGTT_S_AIR_TIME + f) * (long)N + e] = t.air[f];
    S[(PGTT_S_SWING_PEAK + f) * (long)N + e] = t.peak[f];
    S[(PGTT_S_LAST_CONTACT + f) * (long)N + e] = t.last_contact[f];
  }
#pragma unroll
  for (int i = 0; i < 24; i++) { S[(PGTT_S_QVEL_HIST + i) * (long)N + e] = hist_v[i]; S[(PGTT_S_QERR_HIST + i) * (long)N + e] = hist_q[i]; }
#pragma unroll
  for (int i = 0; i < 12; i++) { S[(PGTT_S_LAST_LAST_ACT + i) * (long)N + e] = st[PGTT_S_LAST_ACT + i]; S[(PGTT_S_LAST_ACT + i) * (long)N + e] = act[i]; }
  I[PGTT_I_STEP * (long)N + e] = t.step_ctr; I[PGTT_I_STEPS_UNTIL_CMD * (long)N + e] = t.timer;
  I[PGTT_I_RNG_CTR * (long)N + e] = (int)(ep + 1u); I[PGTT_I_EP_STEPS * (long)N + e] = ep_steps;
  a.buf.reward[e] = t.reward; a.buf.done[e] = wdone ? 1.f : 0.f;
#pragma unroll
  for (int k = 0; k < PGTT_NMETRIC; k++) a.buf.metrics[k * (long)N + e] = t.metrics[k];
  if (a.buf.interval_sums) {
#pragma unroll
    for (int k = 0; k < PGTT_NMETRIC; k++) a.buf.interval_sums[k * (long)N + e] += t.metrics[k];
    a.buf.interval_sums[PGTT_NMETRIC * (long)N + e] += t.reward;
    a.buf.interval_sums[(PGTT_NMETRIC + 1) * (long)N + e] += wdone ? 1.f : 0.f;
  }
  // AutoReset: a finished episode continues from the env's first state and first observation
  if (cfg->autoreset && wdone && a.buf.first_state && a.buf.first_obs) {
    for (int r = 0; r < PGTT_S_CMD; r++) S[r * (long)N + e] = a.buf.first_state[r * (long)N + e];
    const float* fo = a.buf.first_obs + (long)e * (OBSD + PRIVD);
    for (int i = 0; i < OBSD; i++) a.buf.obs_state[(long)e * OBSD + i] = fo[i];
    for (int i = 0; i < PRIVD; i++) a.buf.obs_priv[(long)e * PRIVD + i] = fo[OBSD + i];
  }
}

}  // namespace pgtt
