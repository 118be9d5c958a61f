// pgtt_physics_quad.hip.h — the physics of one environment spread over 4 or 16 lanes (gfx950, wave64).
//
// Lane l of an env owns LEG l (FL,FR,RL,RR = body-tree order): its 3 links, 3 dofs, 3 joint-limit rows, its foot's plane
// contact and box contacts, the leg blocks M_ll (3x3) / M_lb (3x6) of the arrowhead inertia and of the Newton Hessian.
// Everything that belongs to the floating base (pose, COM, 6x6 base block, base parts of every 18-vector) is REPLICATED
// in the lanes of the env and kept bitwise identical: cross-lane sums are symmetric DPP butterflies that return the same
// bits in every lane.  Two layouts (compile-time PG_SUBS, see "cross-lane primitives" below): quad = one lane per leg,
// 16 envs per wave; hex = four sub-lanes per leg (leg state replicated, selected loops split), 4 envs per wave.
// Box-contact records live in LDS (one column per leg); everything else is in registers (256 VGPRs plus 230-255 AGPRs used
// as spill space, no scratch).
//
// Arithmetic contract: MJX forward + Euler for the Go2 tree, active contact set identical to MJX's top-k semantics,
// Newton(5) x linesearch(5); see pgtt_physics.hip.h for the shared helpers and DESIGN.md 5.1 for the invariants.
#pragma once
#include "pgtt_physics.hip.h"

namespace pgtt {

// ------------------------------------------------------------------ cross-lane primitives (DPP, VALU only)
// Two lane layouts share this file (compile-time PG_SUBS, one translation unit each):
//   PG_SUBS = 1  "quad": lane = 4*env + leg                  16 envs per wave
//   PG_SUBS = 4  "hex" : lane = 16*env + 4*leg + sub          4 envs per wave; the four SUB-lanes of a leg hold the
//                       leg's state replicated (bit-identical) and split selected loops among themselves (constraint
//                       rows of the line search / Hessian, boxes of the collision passes).  A launch lasts as long as
//                       one wave's instruction stream, so for small batches the shorter stream wins.
//   PG_SUBS = 2  "oct" : lane = 16*row + 4*leg + 2*env_in_row + sub   8 envs per wave (two per 16-lane DPP row, interleaved
//                       so that the legs of an env are 4 lanes apart exactly as in the hex layout and its two sub-lanes are
//                       neighbours).  Two sub-lanes per leg split the row pairs of the line search and the boxes of the
//                       collision passes; everything else follows the quad layout's per-leg code, done twice.  It is the
//                       layout for 4097..8192 envs: 1024 waves, still one per SIMD.
// "quad_*" primitives act over the four LEGS of an env, "sub_*" over the sub-lanes of a leg.
#ifndef PG_SUBS
#define PG_SUBS 1
#endif
constexpr int kSubs = PG_SUBS;
// this translation unit's kernel has box terrain (one TU per physics_kernel variant, pgtt_physics_inst.hip); 0 where the header is only parsed
#ifdef PG_TERRAIN
constexpr bool kTerrainTU = PG_TERRAIN != 0;
#else
constexpr bool kTerrainTU = false;
#endif
constexpr int kEnvsPerWave = 16 / PG_SUBS;
// centre / world-AABB half extents of the env's boxes staged in LDS per env (hex, oct) or read from the resident table where they are needed (quad:
// 16 envs x 100 boxes x 24 B = 38 KB per workgroup were what kept a CU at two quad workgroups; without them four fit, one per SIMD - collide())
constexpr bool kBoxLds = PG_SUBS != 1;
static_assert(kBoxLds || PG_ADDR32, "collide()'s boxA / boxH read the resident table through box0 (a 32-bit element index from the table's base)");
static_assert(PG_SUBS == 1 || PG_SUBS == 2 || PG_SUBS == 4, "lane layouts: quad, oct, hex");
// who am I: sub-lane of the leg, leg of the env, env of the wave, lane of the env, LDS column of the (env, leg) pair
PG_INL int lane_sub() { return kSubs == 4 ? (int)(threadIdx.x & 3) : (kSubs == 2 ? (int)(threadIdx.x & 1) : 0); }
PG_INL int lane_leg() { return kSubs == 1 ? (int)(threadIdx.x & 3) : (int)((threadIdx.x >> 2) & 3); }
PG_INL int lane_env() { return kSubs == 1 ? (int)(threadIdx.x >> 2) : (kSubs == 4 ? (int)(threadIdx.x >> 4) : (int)(2 * (threadIdx.x >> 4) + ((threadIdx.x >> 1) & 1))); }
PG_INL int lane_in_env() { return kSubs == 2 ? 2 * lane_leg() + lane_sub() : (int)(threadIdx.x % (4 * kSubs)); }
PG_INL int lane_col() { return kSubs == 2 ? 4 * lane_env() + lane_leg() : (int)(threadIdx.x / kSubs); }
template <int CTRL>
PG_INL float dpp_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
PG_INL int dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true); }
// The butterfly adds must stay plain adds: with contraction on, `p*q + dpp(p*q)` may be fused into
// fma(p, q, dpp(round(p*q))), which differs between the two lanes of a pair and breaks the invariant that every
// lane of an env holds the bit-identical sum (base-body quantities are replicated, never broadcast).
// Both steps pair lane i with a lane that computes the same two operands in the other order, so all lanes agree.
#if PG_SUBS == 1
constexpr int kLegStep1 = 0xB1, kLegStep2 = 0x4E;       // quad_perm [1,0,3,2], [2,3,0,1]
#else
constexpr int kLegStep1 = 0x128, kLegStep2 = 0x124;     // row_ror:8, row_ror:4 (legs are 4 lanes apart in a 16-lane row)
#endif
PG_INL float quad_sum(float x) {
#pragma clang fp contract(off)
  x = x + dpp_f<kLegStep1>(x);
  x = x + dpp_f<kLegStep2>(x);
  return x;
}
PG_INL int quad_sum_i(int x) { x += dpp_i<kLegStep1>(x); x += dpp_i<kLegStep2>(x); return x; }
PG_INL float quad_max(float x) { x = fmaxf(x, dpp_f<kLegStep1>(x)); x = fmaxf(x, dpp_f<kLegStep2>(x)); return x; }      // max over the legs (exact, order-free)
PG_INL V3 quad_sum(V3 a) { return v3(quad_sum(a.x), quad_sum(a.y), quad_sum(a.z)); }
// sum over the sub-lanes of a leg (four in the hex layout, two neighbours in the oct layout; identity in the quad layout)
PG_INL float sub_sum(float x) {
#pragma clang fp contract(off)
#if PG_SUBS >= 2
  x = x + dpp_f<0xB1>(x);
#endif
#if PG_SUBS == 4
  x = x + dpp_f<0x4E>(x);
#endif
  return x;
}
PG_INL int sub_sum_i(int x) {
#if PG_SUBS >= 2
  x += dpp_i<0xB1>(x);
#endif
#if PG_SUBS == 4
  x += dpp_i<0x4E>(x);
#endif
  return x;
}
// value held by sub-lane Q of the own leg (hex: quad_perm [Q,Q,Q,Q]; oct, Q < 2: quad_perm [Q,Q,2+Q,2+Q])
constexpr int sub_bcast_ctrl(int q) { return PG_SUBS == 4 ? q * 0x55 : ((q & 1) ? 0xF5 : 0xA0); }
template <int Q> PG_INL int sub_bcast(int x) { return PG_SUBS >= 2 ? dpp_i<sub_bcast_ctrl(Q)>(x) : x; }
template <int Q> PG_INL float sub_bcast(float x) { return PG_SUBS >= 2 ? dpp_f<sub_bcast_ctrl(Q)>(x) : x; }
PG_INL unsigned sub_or(unsigned x) {
#if PG_SUBS >= 2
  x |= (unsigned)dpp_i<0xB1>((int)x);
#endif
#if PG_SUBS == 4
  x |= (unsigned)dpp_i<0x4E>((int)x);
#endif
  return x;
}
// hex layout: sub-lane k of a leg OWNS box slot k of that leg: it alone completes the slot's record, forms its Jacobian
// products and its force / Hessian contributions (summed over the sub-lanes afterwards), in ONE pass, so that this work
// does not grow with the number of slots in use; in the quad layout every lane loops over all its slots.
// value held by leg J (same sub-lane)
#if PG_SUBS == 1
template <int J> PG_INL float quad_bcast(float x) { return dpp_f<J * 0x55>(x); }
template <int J> PG_INL int quad_bcast(int x) { return dpp_i<J * 0x55>(x); }
#else
// row_ror:4k hands lane i the value of lane i - 4k, i.e. of leg (own - k) & 3; picked with two levels of selects
template <int J> PG_INL int quad_bcast(int x) {
  const int k = ((int)(threadIdx.x >> 2) - J) & 3;
  const bool b0 = (k & 1) != 0, b1 = (k & 2) != 0;
  const int r1 = dpp_i<0x124>(x), r2 = dpp_i<0x128>(x), r3 = dpp_i<0x12C>(x);
  const int lo = b0 ? r1 : x, hi = b0 ? r3 : r2;
  return b1 ? hi : lo;
}
template <int J> PG_INL float quad_bcast(float x) { return __int_as_float(quad_bcast<J>(__float_as_int(x))); }
#endif

constexpr int kMaxB = 4;          // box contacts one foot can hold (= max_contact_points of the reference)
constexpr int kGridG = 16;        // cells per side of the terrain grid (pgtt_set_terrain): 128-bit box mask per cell and variant
constexpr int kMaxPenQ = 4;       // penetrating (foot, box) pairs tracked per foot

struct QArrow { float bb[21]; float lb[18]; float ll[6]; };

struct QContact {
  bool on;            // slot in use
  bool row_active;    // dist < margin
  int box;            // -1 plane, else box index
  float dist, mu, D;
  float aref[4];
  V3 off;             // contact point relative to the robot COM
  V3 fr[3];           // contact frame rows (normal, tangent1, tangent2), already multiplied by the body sign
};
// A contact Jacobian row is fr[a] . (V + W x off) where (W, V) is the calf's spatial motion: J itself (3x9) is
// never stored; J x, J^T f and J^T W J are formed from (off, fr) and the leg's cdofs.

// Box-contact records of the own foot live in LDS, one column per LEG (= per lane in the quad layout; shared by the four
// sub-lanes of the leg in the hex layout): field f of slot k at sh[(k*kSlotFields + f)*kSlotCols + col] (consecutive
// columns -> consecutive banks).  This keeps the Newton loop's register footprint independent of the number of box
// slots and lets the slot loops be real (wave-uniform) loops; in the hex layout it is also how the sub-lane that owns a
// slot hands its rows to the sub-lanes that own the rows (LDS accesses of one wave complete in program order).
constexpr int kSlotFields = 29;   // dist mu D aref[4] off[3] fr[9] flags box | jar[4] jv[4]
constexpr int kSlotCols = 64 / PG_SUBS;
struct BoxSlots {
  float* sh; int lane;            // lane = column = leg index within the wave
  PG_INL float& at(int k, int f) const { return sh[(k * kSlotFields + f) * kSlotCols + lane]; }
  PG_INL void store(int k, const QContact& c) const {
    at(k, 0) = c.dist; at(k, 1) = c.mu; at(k, 2) = c.D;
#pragma unroll
    for (int r = 0; r < 4; r++) at(k, 3 + r) = c.aref[r];
    at(k, 7) = c.off.x; at(k, 8) = c.off.y; at(k, 9) = c.off.z;
#pragma unroll
    for (int a = 0; a < 3; a++) { at(k, 10 + 3 * a) = c.fr[a].x; at(k, 11 + 3 * a) = c.fr[a].y; at(k, 12 + 3 * a) = c.fr[a].z; }
    at(k, 19) = __int_as_float((c.on ? 1 : 0) | (c.row_active ? 2 : 0));
    at(k, 20) = __int_as_float(c.box);
  }
  PG_INL QContact load(int k) const {
    QContact c;
    c.dist = at(k, 0); c.mu = at(k, 1); c.D = at(k, 2);
#pragma unroll
    for (int r = 0; r < 4; r++) c.aref[r] = at(k, 3 + r);
    c.off = v3(at(k, 7), at(k, 8), at(k, 9));
#pragma unroll
    for (int a = 0; a < 3; a++) c.fr[a] = v3(at(k, 10 + 3 * a), at(k, 11 + 3 * a), at(k, 12 + 3 * a));
    int fl = __float_as_int(at(k, 19));
    c.on = (fl & 1) != 0; c.row_active = (fl & 2) != 0;
    c.box = __float_as_int(at(k, 20));
    return c;
  }
  PG_INL void clear_all() const {       // once per launch: LDS comes up uninitialised, and stale NaNs would survive 0 * x
#pragma unroll
    for (int k = 0; k < kMaxB; k++)
#pragma unroll
      for (int f = 0; f < kSlotFields; f++) at(k, f) = f == 0 ? 1.f : (f == 20 ? __int_as_float(-2) : 0.f);
  }
  PG_INL float& jar(int k, int r) const { return at(k, 21 + r); }
  PG_INL float& jv(int k, int r) const { return at(k, 25 + r); }
};

struct QSim {
  // state: base replicated, leg own
  float qb[7], vb[6], wb[6];
  float ql[3], vl[3], wl[3];
  float ctrl[3];
  // kinematics
  V3 p0, com, imu; M3 R0;
  V3 anchor[3], axis[3], footc, sitef;
  I10 cin0, cinl[3];
  S6 cdr[3], cdl[3];
  QArrow M, LM;
  // velocity
  S6 cvel0, cvell[3], cddr[3], cddl[3];
  float qfs_b[6], qfs_l[3], qas_b[6], qas_l[3], act_force[3];
  // constraints
  bool lim_active[3]; float lim_sign[3], lim_D[3], lim_aref[3];
  QContact con0;             // own foot vs the plane (registers)
  int nbox;                  // own box contacts; their records live in LDS (see BoxSlots)
#ifdef PGTT_TRACE
  float* tr = nullptr; int trn = 0;
  PG_INL void rec(float v) { if (tr) tr[4 * trn] = v; trn++; }
  PG_INL void rec(V3 v) { rec(v.x); rec(v.y); rec(v.z); }
#endif
#ifdef PGTT_TIME
  // stage timer (-DPGTT_TIME builds): cyc[i] accumulates shader-clock ticks of stage i over the launch
  long long tlast = 0; float cyc[28] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // 20..27: sub-stages of the line search (own clock)
  long long tls = 0; PG_INL void ltick(int i) { long long t = __builtin_readcyclecounter(); cyc[i] += (float)(t - tls); tls = t; }
#define PG_LTICK(sim, i) (sim).ltick(i)
  PG_INL void tick(int stage) { long long t = __builtin_readcyclecounter(); cyc[stage] += (float)(t - tlast); tlast = t; }
#define PG_TICK(sim, stage) (sim).tick(stage)
  PG_INL void cyc_iter() { cyc[10] += 1.f; }
#else
#define PG_TICK(sim, stage) ((void)0)
#define PG_LTICK(sim, i) ((void)0)
  PG_INL void cyc_iter() {}
#endif
  // outputs
  float qacc_b[6], qacc_l[3];
  int niter, niter_max;
#ifdef PGTT_EFFORT
  // -DPGTT_EFFORT builds (tools/gpu_effort.py): 1 + the line-search rounds THIS env needed, 3 bits per Newton trip, 5 trips per substep
  unsigned long long eff = 0ull; int eff_pos = 0, eff_sub = 0;
  // ... and how often the Hessian is rebuilt although NO lane of the wave has a row that changed sides since the last Newton trip
  unsigned eff_pat = 0xffffffffu; int eff_hess = 0, eff_hess_same = 0;
#endif
  bool pen_overflow;         // some substep of this call met more than kMaxPenQ simultaneously penetrating boxes under this foot (collide())
};

// y = A x
PG_INL void qarrow_mul(const QArrow& A, const float* xb, const float* xl, float* yb, float* yl) {
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float s = 0.f, t = 0.f;
#pragma unroll
    for (int j = 0; j < 6; j++) s += A.bb[i >= j ? tri(i, j) : tri(j, i)] * xb[j];
#pragma unroll
    for (int k = 0; k < 3; k++) t += A.lb[k * 6 + i] * xl[k];
    yb[i] = s + quad_sum(t);
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 6; i++) s += A.lb[k * 6 + i] * xb[i];
#pragma unroll
    for (int j = 0; j < 3; j++) s += A.ll[k >= j ? tri(k, j) : tri(j, k)] * xl[j];
    yl[k] = s;
  }
}
// -DPGTT_EXACT_PIVOT side builds (profiles/r06_rsq_ab.txt): the inverse pivots from a correctly rounded square root and a correctly rounded division
#ifdef PGTT_EXACT_PIVOT
#define PG_RSQ(x) (1.0f / sqrtf(x))
#else
#define PG_RSQ(x) __builtin_amdgcn_rsqf(x)
#endif
PG_INL void qarrow_factor(QArrow& A) {
  float* c = A.ll;
  // The diagonal of a Cholesky factor is only ever used as a divisor (here and in qarrow_solve), so the factor keeps its
  // INVERSE: one v_rsq_f32 (1 ulp) per pivot and multiplications instead of a square root plus ~40 divisions (8
  // instructions each in the 1-ulp form) per factorisation + solve.
  const float i00 = PG_RSQ(c[0]);
  float l10 = c[1] * i00, l20 = c[3] * i00;
  const float i11 = PG_RSQ(c[2] - l10 * l10);
  float l21 = (c[4] - l20 * l10) * i11;
  const float i22 = PG_RSQ(c[5] - l20 * l20 - l21 * l21);
  c[0] = i00; c[1] = l10; c[2] = i11; c[3] = l20; c[4] = l21; c[5] = i22;
  float* w = A.lb;
  if (kSubs != 4) {
#pragma unroll
    for (int k = 0; k < 6; k++) {
      float w0 = w[k] * i00;
      float w1 = (w[6 + k] - l10 * w0) * i11;
      float w2 = (w[12 + k] - l20 * w0 - l21 * w1) * i22;
      w[k] = w0; w[6 + k] = w1; w[12 + k] = w2;
    }
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j <= i; j++)
        A.bb[tri(i, j)] -= quad_sum(w[i] * w[j] + w[6 + i] * w[6 + j] + w[12 + i] * w[12 + j]);
  } else {
    // hex layout: the leg blocks are replicated over the four sub-lanes, so the six columns of W = L^-1 A_lb (three
    // divisions each) and the 21 Schur sums go round the sub-lanes and are handed back with sub-lane broadcasts; every
    // lane ends with the same bits as if it had computed them all.
    const int r = threadIdx.x & 3;
    const bool b0 = (r & 1) != 0, b1 = (r & 2) != 0;
    auto pick = [&](float x0, float x1, float x2, float x3) { const float lo = b0 ? x1 : x0, hi = b0 ? x3 : x2; return b1 ? hi : lo; };
    // column r (all sub-lanes) and column 4 + r (sub-lanes 0, 1; the others redo column r + 2, unused)
    float ca[3], cb[3];
    {
      const float a0 = pick(w[0], w[1], w[2], w[3]), a1 = pick(w[6], w[7], w[8], w[9]), a2 = pick(w[12], w[13], w[14], w[15]);
      ca[0] = a0 * i00; ca[1] = (a1 - l10 * ca[0]) * i11; ca[2] = (a2 - l20 * ca[0] - l21 * ca[1]) * i22;
      const float e0 = b0 ? w[5] : w[4], e1 = b0 ? w[11] : w[10], e2 = b0 ? w[17] : w[16];
      cb[0] = e0 * i00; cb[1] = (e1 - l10 * cb[0]) * i11; cb[2] = (e2 - l20 * cb[0] - l21 * cb[1]) * i22;
    }
#pragma unroll
    for (int m = 0; m < 3; m++) {
      w[6 * m + 0] = sub_bcast<0>(ca[m]); w[6 * m + 1] = sub_bcast<1>(ca[m]); w[6 * m + 2] = sub_bcast<2>(ca[m]); w[6 * m + 3] = sub_bcast<3>(ca[m]);
      w[6 * m + 4] = sub_bcast<0>(cb[m]); w[6 * m + 5] = sub_bcast<1>(cb[m]);
    }
    // Schur sums: entry t of the packed 6x6 triangle belongs to sub-lane t & 3
    float mine[6];
#pragma unroll
    for (int q = 0; q < 6; q++) mine[q] = 0.f;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) {
        const int t = tri(i, j);
        const float p = w[i] * w[j] + w[6 + i] * w[6 + j] + w[12 + i] * w[12 + j];
        mine[t >> 2] = (t & 3) == r ? p : mine[t >> 2];
      }
#pragma unroll
    for (int q = 0; q < 6; q++) mine[q] = quad_sum(mine[q]);
#pragma unroll
    for (int t = 0; t < 21; t++) {
      const float sm = (t & 3) == 0 ? sub_bcast<0>(mine[t >> 2]) : ((t & 3) == 1 ? sub_bcast<1>(mine[t >> 2]) : ((t & 3) == 2 ? sub_bcast<2>(mine[t >> 2]) : sub_bcast<3>(mine[t >> 2])));
      A.bb[t] -= sm;
    }
  }
#pragma unroll
  for (int j = 0; j < 6; j++) {
    float s = A.bb[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) s -= A.bb[tri(j, k)] * A.bb[tri(j, k)];
    const float id = PG_RSQ(s);
    A.bb[tri(j, j)] = id;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      float t = A.bb[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) t -= A.bb[tri(i, k)] * A.bb[tri(j, k)];
      A.bb[tri(i, j)] = t * id;
    }
  }
}
PG_INL void qarrow_solve(const QArrow& F, const float* bb, const float* bl, float* xb, float* xl) {
  const float* c = F.ll;
  float y0 = bl[0] * c[0];                    // c[0], c[2], c[5] and bb[tri(i, i)] hold inverse pivots (qarrow_factor)
  float y1 = (bl[1] - c[1] * y0) * c[2];
  float y2 = (bl[2] - c[3] * y0 - c[4] * y1) * c[5];
  float z[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float rb = bb[i] - quad_sum(F.lb[i] * y0 + F.lb[6 + i] * y1 + F.lb[12 + i] * y2);
#pragma unroll
    for (int k = 0; k < i; k++) rb -= F.bb[tri(i, k)] * z[k];
    z[i] = rb * F.bb[tri(i, i)];
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    float s = z[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) s -= F.bb[tri(k, i)] * xb[k];
    xb[i] = s * F.bb[tri(i, i)];
  }
  float t0 = y0, t1 = y1, t2 = y2;
#pragma unroll
  for (int k = 0; k < 6; k++) { t0 -= F.lb[k] * xb[k]; t1 -= F.lb[6 + k] * xb[k]; t2 -= F.lb[12 + k] * xb[k]; }
  float x2 = t2 * c[5];
  float x1 = (t1 - c[4] * x2) * c[2];
  float x0 = (t0 - c[1] * x1 - c[3] * x2) * c[0];
  xl[0] = x0; xl[1] = x1; xl[2] = x2;
}

// per-env model view for one leg
struct QEnvModel {
  float mass0, massl[3];
  V3 base_ipos;
  float qpos0j[3], armature[3], damping[3], gain[3], bias1[3];
  float floor_friction;
};
template <bool HAS_DR>
PG_INL void qload_env_model(const PgttModel* __restrict__ m, const float* __restrict__ prm, int N, int e, int l, QEnvModel& em) {
  if (HAS_DR) {
    em.mass0 = PG_ROW(prm, PGTT_P_BODY_MASS + 0, N, e);
    em.base_ipos = v3(PG_ROW(prm, PGTT_P_BASE_IPOS + 0, N, e), PG_ROW(prm, PGTT_P_BASE_IPOS + 1, N, e), PG_ROW(prm, PGTT_P_BASE_IPOS + 2, N, e));
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int j = 3 * l + k, a = 3 * (l ^ 1) + k;
      em.massl[k] = PG_ROW(prm, PGTT_P_BODY_MASS + 1 + j, N, e);
      em.qpos0j[k] = PG_ROW(prm, PGTT_P_QPOS0 + j, N, e);
      em.armature[k] = PG_ROW(prm, PGTT_P_ARMATURE + j, N, e);
      em.damping[k] = PG_ROW(prm, PGTT_P_DAMPING + j, N, e);
      em.gain[k] = PG_ROW(prm, PGTT_P_GAIN + a, N, e);
      em.bias1[k] = PG_ROW(prm, PGTT_P_BIAS1 + a, N, e);
    }
    em.floor_friction = PG_ROW(prm, PGTT_P_FLOOR_FRICTION, N, e);
  } else {
    em.mass0 = m->body_mass[0];
    em.base_ipos = v3(m->body_ipos[0][0], m->body_ipos[0][1], m->body_ipos[0][2]);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int j = 3 * l + k, a = 3 * (l ^ 1) + k;
      em.massl[k] = m->body_mass[1 + j]; em.qpos0j[k] = m->qpos0[7 + j]; em.armature[k] = m->dof_armature[6 + j];
      em.damping[k] = m->dof_damping[6 + j]; em.gain[k] = m->act_gain[a]; em.bias1[k] = m->act_bias[a][1];
    }
    em.floor_friction = m->floor_friction[0];
  }
}

struct QPen { float dist, key; int idx; };
// (broad-phase key, pair index) as one integer that sorts like the pair (key through the order-preserving float -> uint map)
PG_INL unsigned long long packed_key(float key, int idx) {
  const unsigned u = __float_as_uint(key);
  const unsigned mono = u ^ ((unsigned)((int)u >> 31) | 0x80000000u);
  return ((unsigned long long)mono << 32) | (unsigned)idx;
}     // a penetrating (foot, box) pair; its contact point and normal wait in LDS

struct QPhysics {
  const PgttModel* __restrict__ m;
  const QEnvModel& em;
  QSim& s;
  const int l;      // own leg
  PG_INL QPhysics(const PgttModel* m_, const QEnvModel& em_, QSim& s_, int l_) : m(m_), em(em_), s(s_), l(l_) {}
  // pass 2a of collide(), remembered across the substeps of a launch: where the own foot stood and the (inflated) radius for which the env's
  // pair count was last found <= max_geom_pairs; c2_thr < 0: nothing remembered
  V3 c2_foot = v3(0, 0, 0); float c2_thr = -1.f;

  PG_INL void body_inertia(int mb, V3 xp, Q4 xq, V3 ipos, float mass, V3& xipos, float* Iw) const {
    xipos = xp + qrot(ipos, xq);
    Q4 iq{m->body_iquat[mb][0], m->body_iquat[mb][1], m->body_iquat[mb][2], m->body_iquat[mb][3]};
    M3 xi = qmat(qmul(xq, iq));
    float d0 = m->body_inertia[mb][0], d1 = m->body_inertia[mb][1], d2 = m->body_inertia[mb][2];
    Iw[0] = xi.m[0] * d0 * xi.m[0] + xi.m[1] * d1 * xi.m[1] + xi.m[2] * d2 * xi.m[2];
    Iw[1] = xi.m[3] * d0 * xi.m[3] + xi.m[4] * d1 * xi.m[4] + xi.m[5] * d2 * xi.m[5];
    Iw[2] = xi.m[6] * d0 * xi.m[6] + xi.m[7] * d1 * xi.m[7] + xi.m[8] * d2 * xi.m[8];
    Iw[3] = xi.m[0] * d0 * xi.m[3] + xi.m[1] * d1 * xi.m[4] + xi.m[2] * d2 * xi.m[5];
    Iw[4] = xi.m[0] * d0 * xi.m[6] + xi.m[1] * d1 * xi.m[7] + xi.m[2] * d2 * xi.m[8];
    Iw[5] = xi.m[3] * d0 * xi.m[6] + xi.m[4] * d1 * xi.m[7] + xi.m[5] * d2 * xi.m[8];
    (void)mass;
  }
  PG_INL static void make_cinert(V3 xipos, const float* Iw, float mass, V3 com, I10& c) {
    V3 o = xipos - com; float oo = dot(o, o);
    c.i[0] = Iw[0] + mass * (oo - o.x * o.x); c.i[1] = Iw[1] + mass * (oo - o.y * o.y); c.i[2] = Iw[2] + mass * (oo - o.z * o.z);
    c.i[3] = Iw[3] - mass * o.x * o.y; c.i[4] = Iw[4] - mass * o.x * o.z; c.i[5] = Iw[5] - mass * o.y * o.z;
    c.i[6] = o.x * mass; c.i[7] = o.y * mass; c.i[8] = o.z * mass; c.i[9] = mass;
  }

  // world inertia of the links, handed from kinematics() to inertia()
  V3 xi0, xil[3]; float Iw0[6], Iwl[3][6];

  // Stage order of one forward pass (same arithmetic as fwd_position / fwd_velocity / make_constraint, reordered so that
  // box collision DETECTION runs while only the kinematic frames are live):
  //   kinematics() -> collide() -> inertia() -> velocity_stage() -> constraint_stage()
  PG_INL void kinematics() {
    Q4 q0{s.qb[3], s.qb[4], s.qb[5], s.qb[6]};
    normalize4(q0);
    s.qb[3] = q0.w; s.qb[4] = q0.x; s.qb[5] = q0.y; s.qb[6] = q0.z;
    s.p0 = v3(s.qb[0], s.qb[1], s.qb[2]);
    s.R0 = qmat(q0);
    if (kSubs != 4) body_inertia(0, s.p0, q0, em.base_ipos, em.mass0, xi0, Iw0);
    s.imu = s.p0 + qrot(v3(m->imu_pos[0], m->imu_pos[1], m->imu_pos[2]), q0);
    V3 pp = s.p0; Q4 pq = q0;
    V3 lpos[3]; Q4 lq[3];          // link frames (hex layout: the world inertias are formed afterwards, one body per sub-lane)
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int j = 3 * l + k, mb = 1 + j;
      V3 pos = pp + qrot(v3(m->body_pos[mb][0], m->body_pos[mb][1], m->body_pos[mb][2]), pq);
      V3 ax = v3(m->jnt_axis[j][0], m->jnt_axis[j][1], m->jnt_axis[j][2]);
      float ang = s.ql[k] - em.qpos0j[k];
      float sn, cs; sincosf(0.5f * ang, &sn, &cs);
      s.axis[k] = qrot(ax, pq);
      s.anchor[k] = pos;
      Q4 xq = qmul(pq, Q4{cs, ax.x * sn, ax.y * sn, ax.z * sn});
      if (kSubs != 4) body_inertia(mb, pos, xq, v3(m->body_ipos[mb][0], m->body_ipos[mb][1], m->body_ipos[mb][2]), em.massl[k], xil[k], Iwl[k]);
      lpos[k] = pos; lq[k] = xq;
      pp = pos; pq = xq;
    }
    if (kSubs == 4) {
      // hex layout: the four sub-lanes of a leg held four copies of the same four world inertias (3 links + base, ~100
      // instructions each); sub-lane r now forms the one of body r (3 = the base) and hands it round - same arithmetic per
      // body, so every lane ends with the same bits as before
      const int r = threadIdx.x & 3;
      const int mbr = r == 3 ? 0 : 1 + 3 * l + r;
      const V3 xp = v3(sel4(r, lpos[0].x, lpos[1].x, lpos[2].x, s.p0.x), sel4(r, lpos[0].y, lpos[1].y, lpos[2].y, s.p0.y), sel4(r, lpos[0].z, lpos[1].z, lpos[2].z, s.p0.z));
      const Q4 xq{sel4(r, lq[0].w, lq[1].w, lq[2].w, q0.w), sel4(r, lq[0].x, lq[1].x, lq[2].x, q0.x), sel4(r, lq[0].y, lq[1].y, lq[2].y, q0.y), sel4(r, lq[0].z, lq[1].z, lq[2].z, q0.z)};
      const V3 ipl = v3(m->body_ipos[mbr][0], m->body_ipos[mbr][1], m->body_ipos[mbr][2]);
      const V3 ipos = r == 3 ? em.base_ipos : ipl;
      V3 xi; float Iw[6];
      body_inertia(mbr, xp, xq, ipos, 0.f, xi, Iw);
      xil[0] = v3(sub_bcast<0>(xi.x), sub_bcast<0>(xi.y), sub_bcast<0>(xi.z)); xil[1] = v3(sub_bcast<1>(xi.x), sub_bcast<1>(xi.y), sub_bcast<1>(xi.z));
      xil[2] = v3(sub_bcast<2>(xi.x), sub_bcast<2>(xi.y), sub_bcast<2>(xi.z)); xi0 = v3(sub_bcast<3>(xi.x), sub_bcast<3>(xi.y), sub_bcast<3>(xi.z));
#pragma unroll
      for (int i = 0; i < 6; i++) { Iwl[0][i] = sub_bcast<0>(Iw[i]); Iwl[1][i] = sub_bcast<1>(Iw[i]); Iwl[2][i] = sub_bcast<2>(Iw[i]); Iw0[i] = sub_bcast<3>(Iw[i]); }
    }
    s.footc = pp + qrot(v3(m->foot_geom_pos[l][0], m->foot_geom_pos[l][1], m->foot_geom_pos[l][2]), pq);
    s.sitef = pp + qrot(v3(m->foot_site_pos[l][0], m->foot_site_pos[l][1], m->foot_site_pos[l][2]), pq);
  }

  PG_INL void inertia() {
    V3 part = v3(0, 0, 0); float pm = 0.f;
#pragma unroll
    for (int k = 2; k >= 0; k--) { part = part + xil[k] * em.massl[k]; pm += em.massl[k]; }
    V3 tot = quad_sum(part) + xi0 * em.mass0;
    float mt = quad_sum(pm) + em.mass0;
    s.com = tot * (1.0f / fmaxf(mt, kMinVal));
    make_cinert(xi0, Iw0, em.mass0, s.com, s.cin0);
#pragma unroll
    for (int k = 0; k < 3; k++) make_cinert(xil[k], Iwl[k], em.massl[k], s.com, s.cinl[k]);
    V3 ob = s.com - s.p0;
#pragma unroll
    for (int k = 0; k < 3; k++) { V3 a = mcol(s.R0, k); s.cdr[k] = S6{a, cross(a, ob)}; }
#pragma unroll
    for (int k = 0; k < 3; k++) s.cdl[k] = S6{s.axis[k], cross(s.axis[k], s.com - s.anchor[k])};
    // composite inertia of the own leg, then of the whole robot (quad sum), and the arrowhead M
    I10 crb = s.cinl[2];
#pragma unroll
    for (int k = 2; k >= 0; k--) {
      if (k < 2) {
#pragma unroll
        for (int i = 0; i < 10; i++) crb.i[i] += s.cinl[k].i[i];
      }
      S6 f = inert_mul(crb, s.cdl[k]);
#pragma unroll
      for (int kk = 0; kk <= k; kk++) s.M.ll[tri(k, kk)] = dot6(s.cdl[kk], f);
      s.M.lb[k * 6 + 0] = f.l.x; s.M.lb[k * 6 + 1] = f.l.y; s.M.lb[k * 6 + 2] = f.l.z;
#pragma unroll
      for (int r = 0; r < 3; r++) s.M.lb[k * 6 + 3 + r] = dot6(s.cdr[r], f);
      s.M.ll[tri(k, k)] += em.armature[k];
    }
    I10 crb_base;
#pragma unroll
    for (int i = 0; i < 10; i++) crb_base.i[i] = s.cin0.i[i] + quad_sum(crb.i[i]);
#pragma unroll
    for (int k = 0; k < 6; k++) {
      S6 cd = k < 3 ? S6{v3(0, 0, 0), v3(k == 0, k == 1, k == 2)} : s.cdr[k - 3];
      S6 f = inert_mul(crb_base, cd);
      float fl[3] = {f.l.x, f.l.y, f.l.z};
#pragma unroll
      for (int kk = 0; kk <= k; kk++) s.M.bb[tri(k, kk)] = kk < 3 ? fl[kk] : dot6(s.cdr[kk - 3], f);
    }
    s.LM = s.M;
#ifdef PGTT_TRACE
    s.rec(400.f);
    for (int i = 0; i < 7; i++) s.rec(s.qb[i]);            // 1..7
    for (int i = 0; i < 3; i++) s.rec(s.ql[i]);            // 8..10
    s.rec(em.mass0); for (int i = 0; i < 3; i++) s.rec(em.massl[i]);   // 11..14
    s.rec(xi0); for (int i = 0; i < 6; i++) s.rec(Iw0[i]);             // 15..17, 18..23
    for (int k = 0; k < 3; k++) s.rec(xil[k]);                          // 24..32
    s.rec(part); s.rec(pm); s.rec(tot); s.rec(mt); s.rec(s.com);       // 33..35, 36, 37..39, 40, 41..43
    for (int i = 0; i < 10; i++) s.rec(s.cin0.i[i]);                    // 44..53
    for (int i = 0; i < 10; i++) s.rec(crb.i[i]);                       // 54..63
    for (int i = 0; i < 10; i++) s.rec(crb_base.i[i]);                  // 64..73
    for (int k = 0; k < 3; k++) { s.rec(s.cdr[k].a); s.rec(s.cdr[k].l); }   // 74..91
    for (int i = 0; i < 21; i++) s.rec(s.M.bb[i]);                      // 92..112
#endif
    qarrow_factor(s.LM);
  }

  PG_INL void velocity_stage() {
    S6 cv0{v3(0, 0, 0), v3(s.vb[0], s.vb[1], s.vb[2])};
#pragma unroll
    for (int k = 0; k < 3; k++) s.cddr[k] = motion_cross(cv0, s.cdr[k]);
    S6 cvb = cv0;
#pragma unroll
    for (int k = 0; k < 3; k++) cvb = cvb + s.cdr[k] * s.vb[3 + k];
    s.cvel0 = cvb;
    S6 caccb{v3(0, 0, 0), v3(-m->gravity[0], -m->gravity[1], -m->gravity[2])};
#pragma unroll
    for (int k = 0; k < 3; k++) caccb = caccb + s.cddr[k] * s.vb[3 + k];
    auto body_force = [&](const I10& ci, S6 cv, S6 cacc) {
      S6 f1 = inert_mul(ci, cacc);
      S6 f2 = inert_mul(ci, cv);
      return f1 + motion_cross_force(cv, f2);
    };
    S6 cv = cvb, ca = caccb, fb[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      s.cddl[k] = motion_cross(cv, s.cdl[k]);
      cv = cv + s.cdl[k] * s.vl[k];
      s.cvell[k] = cv;
      ca = ca + s.cddl[k] * s.vl[k];
      fb[k] = body_force(s.cinl[k], cv, ca);
    }
    fb[1] = fb[1] + fb[2]; fb[0] = fb[0] + fb[1];
    float bias_l[3];
#pragma unroll
    for (int k = 0; k < 3; k++) bias_l[k] = dot6(s.cdl[k], fb[k]);
    S6 f0 = body_force(s.cin0, cvb, caccb);
    S6 fbase{f0.a + quad_sum(fb[0].a), f0.l + quad_sum(fb[0].l)};
    float bias_b[6] = {fbase.l.x, fbase.l.y, fbase.l.z, dot6(s.cdr[0], fbase), dot6(s.cdr[1], fbase), dot6(s.cdr[2], fbase)};
#pragma unroll
    for (int i = 0; i < 6; i++) s.qfs_b[i] = -m->dof_damping[i] * s.vb[i] - bias_b[i];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int a = 3 * (l ^ 1) + k;
      float c = fminf(fmaxf(s.ctrl[k], m->act_ctrlrange[a][0]), m->act_ctrlrange[a][1]);
      float force = em.gain[k] * c + (m->act_bias[a][0] + em.bias1[k] * s.ql[k] + m->act_bias[a][2] * s.vl[k]);
      force = fminf(fmaxf(force, m->act_forcerange[a][0]), m->act_forcerange[a][1]);
      s.act_force[k] = force;
      s.qfs_l[k] = -em.damping[k] * s.vl[k] - bias_l[k] + force;
    }
    qarrow_solve(s.LM, s.qfs_b, s.qfs_l, s.qas_b, s.qas_l);
#ifdef PGTT_TRACE
    s.rec(500.f);
    for (int i = 0; i < 21; i++) s.rec(s.LM.bb[i]);       // 1..21
    for (int i = 0; i < 18; i++) s.rec(s.LM.lb[i]);       // 22..39
    for (int i = 0; i < 6; i++) s.rec(s.LM.ll[i]);        // 40..45
    for (int i = 0; i < 6; i++) s.rec(s.M.ll[i]);         // 46..51
    for (int i = 0; i < 6; i++) s.rec(s.qfs_b[i]);        // 52..57
    for (int i = 0; i < 3; i++) s.rec(s.qfs_l[i]);        // 58..60
    for (int i = 0; i < 6; i++) s.rec(s.qas_b[i]);        // 61..66
    for (int i = 0; i < 3; i++) s.rec(s.qas_l[i]);        // 67..69
    for (int i = 0; i < 3; i++) s.rec(s.ctrl[i]);         // 70..72
    for (int i = 0; i < 3; i++) s.rec(bias_l[i]);         // 73..75
    for (int i = 0; i < 3; i++) s.rec(s.act_force[i]);    // 76..78
    for (int i = 0; i < 6; i++) s.rec(s.vb[i]);           // 79..84
    for (int i = 0; i < 3; i++) s.rec(s.vl[i]);           // 85..87
#endif
  }

  PG_INL void contact_jac(QContact& c, V3 pos, V3 n, V3 t1, V3 t2, float sign) const {
    c.off = pos - s.com;
    c.fr[0] = n * sign; c.fr[1] = t1 * sign; c.fr[2] = t2 * sign;
  }
  PG_INL void finish_contact(QContact& c, const float* solref, const float* solimp, float includemargin, float invw_body) const {
    float pos = c.dist - includemargin;
    c.row_active = c.on && pos < 0.f;
    float kimp, b, imp;
    kbi(m->timestep, solref, solimp, pos, kimp, b, imp);
    float mu = c.mu;
    float invweight = (invw_body + mu * mu * invw_body) * 2.0f * mu * mu / m->impratio;
    float r = fmaxf(invweight * (1.0f - imp) / imp, kMinVal);
    c.D = c.row_active ? 1.0f / r : 0.f;
    // J qvel = frame . (velocity of the contact point) ; the calf's spatial velocity is cvell[2]
    V3 vp = s.cvell[2].l + cross(s.cvell[2].a, c.off);
    float t[3] = {dot(c.fr[0], vp), dot(c.fr[1], vp), dot(c.fr[2], vp)};
    float jv[4] = {t[0] + mu * t[1], t[0] - mu * t[1], t[0] + mu * t[2], t[0] - mu * t[2]};
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) c.aref[r4] = c.row_active ? (-b * jv[r4] - kimp * pos) : 0.f;
  }
  PG_INL void clear_contact(QContact& c) const {
    c.on = false; c.row_active = false; c.box = -2; c.dist = 1.f; c.mu = 0.f; c.D = 0.f;
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) c.aref[r4] = 0.f;
    c.off = v3(0, 0, 0); c.fr[0] = v3(0, 0, 0); c.fr[1] = v3(0, 0, 0); c.fr[2] = v3(0, 0, 0);
  }

  // sh_box: LDS copy of the env's box centres + bounding radii, [b*16 + quad]; sh_key: LDS scratch for the
  // broad-phase keys of the own foot, [b*64 + lane] (both staged / owned by the calling kernel)
  // constraint rows: joint limits, the plane contact, and the box contacts found by collide() (world point / normal parked
  // in the slot record) completed with their Jacobian frame, impedance and reference acceleration
  PG_INL void constraint_stage(bool has_boxes, const float* __restrict__ box_fr, int N, int e, const BoxSlots& slots) {
    float lpos[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int j = 3 * l + k;
      float q = s.ql[k];
      float dmin = q - m->jnt_range[j][0], dmax = m->jnt_range[j][1] - q;
      lpos[k] = fminf(dmin, dmax);
      s.lim_active[k] = lpos[k] < 0.f;
      s.lim_sign[k] = dmin < dmax ? 1.0f : -1.0f;
      s.lim_D[k] = 0.f; s.lim_aref[k] = 0.f;
    }
    // impedance of a limit row only matters when the row is active: skipped while no lane of the wave is past a limit
    if (__ballot(s.lim_active[0] | s.lim_active[1] | s.lim_active[2]) != 0ull) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int j = 3 * l + k;
        const bool act = s.lim_active[k];
        float kimp, b, imp;
        kbi(m->timestep, m->jnt_solref, m->jnt_solimp, lpos[k], kimp, b, imp);
        float r = fmaxf(m->dof_invweight0[6 + j] * (1.0f - imp) / imp, kMinVal);
        s.lim_D[k] = act ? 1.0f / r : 0.f;
        s.lim_aref[k] = act ? (-b * (s.lim_sign[k] * s.vl[k]) - kimp * lpos[k]) : 0.f;
      }
    }
    auto mix = [&](const float* sr1, const float* si1, float sm1, const float* sr2, const float* si2, float sm2, float* sr, float* si) {
      float mixw = sm1 / (sm1 + sm2);
      if (sm1 < kMinVal && sm2 < kMinVal) mixw = 0.5f; else if (sm1 < kMinVal) mixw = 0.f; else if (sm2 < kMinVal) mixw = 1.f;
      if (sr1[0] > 0.f && sr2[0] > 0.f) { sr[0] = mixw * sr1[0] + (1 - mixw) * sr2[0]; sr[1] = mixw * sr1[1] + (1 - mixw) * sr2[1]; }
      else { sr[0] = fminf(sr1[0], sr2[0]); sr[1] = fminf(sr1[1], sr2[1]); }
#pragma unroll
      for (int i = 0; i < 5; i++) si[i] = mixw * si1[i] + (1 - mixw) * si2[i];
    };
    const float invw_calf = m->body_invweight0[3 + 3 * l][0];
    const float rad = m->foot_radius[l];
    {   // own foot vs the plane
      float sr[2], si[5];
      mix(m->floor_solref, m->floor_solimp, m->floor_solmix, m->foot_solref, m->foot_solimp, m->foot_solmix, sr, si);
      float margin = fmaxf(m->floor_margin, m->foot_margin) - fmaxf(m->floor_gap, m->foot_gap);
      QContact& c = s.con0;
      c.on = true; c.box = -1; c.mu = fmaxf(em.floor_friction, m->foot_friction[0]);
      c.dist = s.footc.z - rad;
      V3 pos = s.footc - v3(0, 0, 1) * (rad + 0.5f * c.dist);
      contact_jac(c, pos, v3(0, 0, 1), v3(0, 1, 0), v3(-1, 0, 0), 1.0f);
      // a foot farther from the plane than the margin has no active row (D = 0, aref = 0): skipped wave-wide on terrain
      if (__ballot(c.dist - margin < 0.f) != 0ull) finish_contact(c, sr, si, margin, invw_calf);
      else { c.row_active = false; c.D = 0.f; c.aref[0] = 0.f; c.aref[1] = 0.f; c.aref[2] = 0.f; c.aref[3] = 0.f; }
    }
    if (!has_boxes) return;
    // hex layout: box slot 3 is free unless some foot of the wave holds four box contacts; the plane contact then goes
    // into its record and sub-lane 3 works on it like the other sub-lanes work on their box slots (QSolver::plane_sub)
    if (kSubs == 4 && __ballot(s.nbox > 3) == 0ull) slots.store(3, s.con0);
    float sr[2], si[5];
    mix(m->foot_solref, m->foot_solimp, m->foot_solmix, m->box_solref, m->box_solimp, m->box_solmix, sr, si);
    float margin = fmaxf(m->foot_margin, m->box_margin) - fmaxf(m->foot_gap, m->box_gap);
#pragma unroll
    for (int k0 = 0; k0 < (kSubs == 1 ? kMaxB : (kSubs == 2 ? 2 : 1)); k0++) {
      const int k = kSubs == 1 ? k0 : lane_sub() + kSubs * k0;       // hex: sub-lane k completes slot k, in one pass; oct: slots k, k + 2
      if (__ballot(k < s.nbox) == 0ull) break;
      if (k < s.nbox) {
        QContact cc;
        cc.on = true; cc.dist = slots.at(k, 0);
        const int b = __float_as_int(slots.at(k, 20));
        cc.box = b;
        float bf = box_fr ? (PG_ADDR32 ? PG_ROW(box_fr, b, N, e) : box_fr[(long)b * N + e]) : m->box_friction[0];
        cc.mu = fmaxf(bf, m->foot_friction[0]);
        V3 n, t1, t2;
        make_frame(v3(slots.at(k, 10), slots.at(k, 11), slots.at(k, 12)), n, t1, t2);
        contact_jac(cc, v3(slots.at(k, 7), slots.at(k, 8), slots.at(k, 9)), n, t1, t2, -1.0f);
        finish_contact(cc, sr, si, margin, invw_calf);
        slots.store(k, cc);
      }
    }
  }

  // box collision detection of the own foot (needs the kinematic frames only): leaves, for each of the s.nbox selected
  // pairs, (dist, box, world contact point, world normal) in the slot record; constraint_stage() completes them
  // PG_ADDR32: `boxes` / `grid` are the tables of ALL variants (wave-uniform bases) and box0 / cell0 the first box / cell of the own env's variant
  // (32-bit element indices: "SGPR base + 32-bit byte offset", see PG_ROW); otherwise they point at the variant's own records and box0 = cell0 = 0
  PG_INL void collide(const TerrainBox* __restrict__ boxes, unsigned box0, int nbox, const float4* sh_box, const float2* sh_box2, const BoxSlots& slots, int quad,
                      const uint4* __restrict__ grid, unsigned cell0, float grid_E, float grid_inv) {
    auto box_rec = [&](int b) -> TerrainBox { if (PG_ADDR32) return pg_at(boxes, box0 + (unsigned)b); return boxes[b]; };
    // (centre, hx) and (hy, hz) of box b: the LDS copy of the launch prologue, or - quad layout - the record itself (L2-resident: every lane of the
    // wave reads the same few tables; the 100-box loops run about once per launch since the broad-phase proof is carried over the substeps)
    auto boxA = [&](int b) -> float4 {
      if (kBoxLds) return sh_box[b * kEnvsPerWave + quad];
      const TerrainBox& t = pg_at(boxes, box0 + (unsigned)b);
      return make_float4(t.px, t.py, t.pz, t.hx);
    };
    auto boxH = [&](int b) -> float2 {
      if (kBoxLds) return sh_box2[b * kEnvsPerWave + quad];
      const TerrainBox& t = pg_at(boxes, box0 + (unsigned)b);
      return make_float2(t.hy, t.hz);
    };
    const float rad = m->foot_radius[l];
    s.nbox = 0;
    if (boxes == nullptr || nbox <= 0) return;
    PG_TICK(s, 11);
    // a slot that is not used in this substep must contribute exact zeros: no active row, D = 0, aref = 0.  Its other fields
    // (mu, offset, frame) only ever meet those zeros and keep whatever finite values they hold (the records are cleared in
    // full once per launch, BoxSlots::clear_all)
#pragma unroll
    for (int k = 0; k < kMaxB; k++) {
      slots.at(k, 2) = 0.f;
#pragma unroll
      for (int r4 = 0; r4 < 4; r4++) slots.at(k, 3 + r4) = 0.f;
      slots.at(k, 19) = __int_as_float(0);
    }
    const int maxp = m->max_geom_pairs, maxc = m->max_contact_points;
    const bool broad = maxp > -1 && 4 * nbox > maxp;
    const float keyC = rad + m->box_rbound;
    // pass 1a: which boxes' world AABBs (grown by the foot radius) contain the own foot centre?  The terrain is static, so the
    // host laid a kGridG x kGridG grid over the map once (pgtt_set_terrain): every cell holds the 128-bit set of boxes whose grown
    // AABB touches it (border cells reach to infinity).  The foot's cell gives a small candidate set, and only those boxes go
    // through the exact AABB test against the LDS-resident centres / extents - the outcome mask is identical to testing all
    // boxes (the cell sets are supersets), at a few LDS rows per foot instead of one hundred.
    unsigned cm[4];
    auto candidates = [&]() {
      cm[0] = 0u; cm[1] = 0u; cm[2] = 0u; cm[3] = 0u;
      const float pad = rad + 1e-5f;
      const float fx = s.footc.x, fy = s.footc.y, fz = s.footc.z;
      const int ix = min(max((int)floorf((fx + grid_E) * grid_inv), 0), kGridG - 1), iy = min(max((int)floorf((fy + grid_E) * grid_inv), 0), kGridG - 1);
      const uint4 cell = PG_ADDR32 ? pg_at(grid, cell0 + (unsigned)(iy * kGridG + ix)) : grid[iy * kGridG + ix];
      // hex layout: sub-lane r tests the candidates with box index = r (mod 4); an OR over the sub-lanes gives every lane the mask
      const unsigned own = kSubs == 1 ? 0xFFFFFFFFu : (kSubs == 4 ? (0x11111111u << (threadIdx.x & 3)) : (0x55555555u << (threadIdx.x & 1)));
      unsigned cd[4] = {cell.x & own, cell.y & own, cell.z & own, cell.w & own};
      for (;;) {
        if (__ballot((cd[0] | cd[1] | cd[2] | cd[3]) != 0u) == 0ull) break;
        const bool z0 = cd[0] == 0u, z1 = z0 & (cd[1] == 0u), z2 = z1 & (cd[2] == 0u);
        const bool have = !(z2 & (cd[3] == 0u));
        const int w = (int)z0 + (int)z1 + (int)z2;
        const unsigned w23 = z2 ? cd[3] : cd[2], w12 = z1 ? w23 : cd[1], word = z0 ? w12 : cd[0];
        const int bit = have ? (__ffs(word) - 1) : 0;
        const unsigned one = have ? (1u << bit) : 0u;
        cd[0] &= w == 0 ? ~one : ~0u; cd[1] &= w == 1 ? ~one : ~0u; cd[2] &= w == 2 ? ~one : ~0u; cd[3] &= w == 3 ? ~one : ~0u;
        const int bx = w * 32 + bit;
        const int b = bx < nbox ? bx : 0;
        const float4 A = boxA(b);
        const float2 H2 = boxH(b);
        const float ex = fabsf(A.x - fx) - A.w, ey = fabsf(A.y - fy) - H2.x, ez = fabsf(A.z - fz) - H2.y;
        const float t = fmaxf(fmaxf(ex, ey), ez) - pad;
        const unsigned hit = ((__float_as_uint(t) >> 31) != 0u && bx < nbox) ? one : 0u;
        cm[0] |= w == 0 ? hit : 0u; cm[1] |= w == 1 ? hit : 0u; cm[2] |= w == 2 ? hit : 0u; cm[3] |= w == 3 ? hit : 0u;
      }
      if (kSubs > 1) { cm[0] = sub_or(cm[0]); cm[1] = sub_or(cm[1]); cm[2] = sub_or(cm[2]); cm[3] = sub_or(cm[3]); }
    };
    candidates();
    PG_TICK(s, 12);
    // pass 1b: narrow phase on the candidates in box order (every lane pops its own lowest set bit); penetrating pairs kept
    QPen pen[kMaxPenQ]; int npen = 0;
#pragma unroll
    for (int i = 0; i < kMaxPenQ; i++) { pen[i].dist = 1.f; pen[i].key = 3.0e38f; pen[i].idx = 0x7fffffff; }
    static_assert(kMaxPenQ <= kMaxB, "pair i parks its point / normal in the record of slot i");
    // lowest set bit of the 128-bit mask (-1 if empty), cleared
    auto pop = [&]() {
      const bool z0 = cm[0] == 0u, z1 = z0 & (cm[1] == 0u), z2 = z1 & (cm[2] == 0u);
      const bool have = !(z2 & (cm[3] == 0u));
      const int w = (int)z0 + (int)z1 + (int)z2;                      // first non-empty word
      const unsigned w23 = z2 ? cm[3] : cm[2], w12 = z1 ? w23 : cm[1], word = z0 ? w12 : cm[0];
      const int bit = have ? (__ffs(word) - 1) : 0;
      const unsigned clr = have ? ~(1u << bit) : ~0u;
      cm[0] &= w == 0 ? clr : ~0u; cm[1] &= w == 1 ? clr : ~0u; cm[2] &= w == 2 ? clr : ~0u; cm[3] &= w == 3 ? clr : ~0u;
      return have ? w * 32 + bit : -1;
    };
    // a penetrating pair takes the next free entry; (dist, key, idx) stay in registers for the selection, the contact
    // point and normal wait in fields 7..12 of the slot record with the same index (dynamic LDS index instead of selects)
    auto keep = [&](const QPen& pp) {
      const bool take = (pp.dist < 0.f) & (npen < kMaxPenQ);
#pragma unroll
      for (int i = 0; i < kMaxPenQ; i++) { const bool hit = take & (i == npen); pen[i].dist = hit ? pp.dist : pen[i].dist; pen[i].key = hit ? pp.key : pen[i].key; pen[i].idx = hit ? pp.idx : pen[i].idx; }
      npen += take ? 1 : 0;
    };
    auto park = [&](int at, V3 pw, V3 nw) {
      slots.at(at, 7) = pw.x; slots.at(at, 8) = pw.y; slots.at(at, 9) = pw.z;
      slots.at(at, 10) = nw.x; slots.at(at, 11) = nw.y; slots.at(at, 12) = nw.z;
    };
    bool redo = false;      // wave-uniform: some foot of the wave penetrates more than kMaxPenQ boxes -> collide_many() below
    for (;;) {
      if (__ballot((cm[0] | cm[1] | cm[2] | cm[3]) != 0u) == 0ull) break;
      int b;
      const int r = threadIdx.x & 3;
      if (kSubs == 1) {
        b = pop();
      } else if (kSubs == 2) {
        const int b0 = pop(), b1 = pop();                           // oct layout: the next two candidates, one per sub-lane
        b = (threadIdx.x & 1) ? b1 : b0;
      } else {
        // hex layout: the next four candidates go to the four sub-lanes (the mask is replicated, so every sub-lane pops
        // all four and keeps its own)
        const int b0 = pop(), b1 = pop(), b2 = pop(), b3 = pop();
        const bool h0 = (r & 1) != 0, h1 = (r & 2) != 0;           // selects, not branches
        const int lo = h0 ? b1 : b0, hi = h0 ? b3 : b2;
        b = h1 ? hi : lo;
      }
      const bool have = b >= 0;
      TerrainBox tb = box_rec(have ? b : 0);
      float nd; V3 pw, nw;
      sphere_box(s.footc, rad, tb, nd, pw, nw);
      QPen pp; pp.dist = have ? nd : 1.f; pp.key = norm(v3(tb.px, tb.py, tb.pz) - s.footc) - keyC; pp.idx = l * nbox + (have ? b : 0);
      if (kSubs == 1) {
        if (__builtin_expect(__ballot((pp.dist < 0.f) & (npen >= kMaxPenQ)) != 0ull, 0)) { redo = true; break; }
        if ((pp.dist < 0.f) & (npen < kMaxPenQ)) park(npen, pw, nw);
        keep(pp);
      } else if (kSubs == 2) {
        QPen g0{sub_bcast<0>(pp.dist), sub_bcast<0>(pp.key), sub_bcast<0>(pp.idx)}, g1{sub_bcast<1>(pp.dist), sub_bcast<1>(pp.key), sub_bcast<1>(pp.idx)};
        const int before = ((threadIdx.x & 1) && g0.dist < 0.f) ? 1 : 0;
        if (__builtin_expect(__ballot(npen + (g0.dist < 0.f ? 1 : 0) + (g1.dist < 0.f ? 1 : 0) > kMaxPenQ) != 0ull, 0)) { redo = true; break; }
        if ((pp.dist < 0.f) & (npen + before < kMaxPenQ)) park(npen + before, pw, nw);
        keep(g0); keep(g1);
      } else {
        // every lane appends the four results in candidate (= box index) order, as the sequential loop would; the
        // sub-lane that computed a penetrating pair parks its point / normal at the entry the pair is going to take
        QPen g0{sub_bcast<0>(pp.dist), sub_bcast<0>(pp.key), sub_bcast<0>(pp.idx)}, g1{sub_bcast<1>(pp.dist), sub_bcast<1>(pp.key), sub_bcast<1>(pp.idx)};
        QPen g2{sub_bcast<2>(pp.dist), sub_bcast<2>(pp.key), sub_bcast<2>(pp.idx)}, g3{sub_bcast<3>(pp.dist), sub_bcast<3>(pp.key), sub_bcast<3>(pp.idx)};
        const int f0 = g0.dist < 0.f ? 1 : 0, f1 = g1.dist < 0.f ? 1 : 0, f2 = g2.dist < 0.f ? 1 : 0;
        if (__builtin_expect(__ballot(npen + f0 + f1 + f2 + (g3.dist < 0.f ? 1 : 0) > kMaxPenQ) != 0ull, 0)) { redo = true; break; }
        const bool h0 = (r & 1) != 0, h1 = (r & 2) != 0;
        const int before = h1 ? (h0 ? f0 + f1 + f2 : f0 + f1) : (h0 ? f0 : 0);        // penetrating pairs of the lower sub-lanes
        if ((pp.dist < 0.f) & (npen + before < kMaxPenQ)) park(npen + before, pw, nw);
        keep(g0); keep(g1); keep(g2); keep(g3);
      }
    }
    if (__builtin_expect(redo, 0)) {
      // ---- MORE than kMaxPenQ boxes penetrated by one foot at once (never on the shipped / generated terrains - at most three boxes meet at
      // a seam - but reachable through pgtt_set_terrain).  MJX ranks all 4 * nbox pairs by centre distance, cuts at max_geom_pairs and keeps
      // the max_contact_points DEEPEST survivors of the env, equal depths in broad-phase order (go2_mjx_feetonly.xml:14-15 -> base.py:153-171).
      // Of one foot only its kMaxPenQ best SURVIVING pairs can be among them, so the pass is run again for the whole wave with a table that
      // keeps exactly those: a pair enters only if it survives the rank cut (exact rank over the env's pairs, when the cut can bite at all),
      // and a full table drops its last entry in MJX's order - depth, then (key, pair index), or the pair index alone where the broad phase
      // does not sort.  Entries stay in scan order (the tie rule of the selection below), which then runs unchanged on the right table.
      // Every sub-lane of a leg does the whole of its leg's work here (replicated, bit-identical); the call raises PGTT_DBG_PEN_OVERFLOW.
      candidates();
      npen = 0;
#pragma unroll
      for (int i = 0; i < kMaxPenQ; i++) { pen[i].dist = 1.f; pen[i].key = 3.0e38f; pen[i].idx = 0x7fffffff; }
      bool cutting = false;      // per env: the max_geom_pairs cut may remove a candidate
      if (broad) {
        // conservative, as pass 2a: pairs of the env at least as close as the farthest candidate (penetrating or not) of any of its feet
        const unsigned sv0 = cm[0], sv1 = cm[1], sv2 = cm[2], sv3 = cm[3];
        float kmax = -3.0e38f;
        for (;;) {
          if (__ballot((cm[0] | cm[1] | cm[2] | cm[3]) != 0u) == 0ull) break;
          const int b = pop();
          const float4 A = boxA(b >= 0 ? b : 0);
          kmax = b >= 0 ? fmaxf(kmax, norm(v3(A.x, A.y, A.z) - s.footc) - keyC) : kmax;
        }
        cm[0] = sv0; cm[1] = sv1; cm[2] = sv2; cm[3] = sv3;
        kmax = quad_max(kmax);
        const float thr = (kmax + keyC) * 1.000002f + 1e-7f, thr2 = kmax > -1.0e38f ? thr * thr : -1.f;
        int cnt = 0;
#pragma unroll 1
        for (int b = 0; b < nbox; b++) {
          const float4 A = boxA(b);
          const V3 dv = v3(A.x, A.y, A.z) - s.footc;
          cnt += dot(dv, dv) <= thr2 ? 1 : 0;
        }
        cutting = quad_sum_i(cnt) > maxp;
      }
      auto ordkey = [&](const QPen& p) { return broad ? packed_key(p.key, p.idx) : (unsigned long long)(unsigned)p.idx; };
      int npenetr = 0;
      for (;;) {
        if (__ballot((cm[0] | cm[1] | cm[2] | cm[3]) != 0u) == 0ull) break;
        const int b = pop();
        const bool have = b >= 0;
        TerrainBox tb = box_rec(have ? b : 0);
        float nd; V3 pw, nw;
        sphere_box(s.footc, rad, tb, nd, pw, nw);
        QPen pp; pp.dist = have ? nd : 1.f; pp.key = norm(v3(tb.px, tb.py, tb.pz) - s.footc) - keyC; pp.idx = l * nbox + (have ? b : 0);
        const bool penn = pp.dist < 0.f;
        npenetr += penn ? 1 : 0;
        bool alive = penn;
        if (__ballot(penn & cutting) != 0ull) {
          // exact broad-phase rank of the four legs' current pairs: pairs of the env that sort before them by (key, pair index)
          const unsigned long long pk = penn ? packed_key(pp.key, pp.idx) : 0ull;
          const int plo = (int)(unsigned)pk, phi = (int)(unsigned)(pk >> 32);
          auto wide = [](int hi, int lo) { return ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo; };
          const unsigned long long c0 = wide(quad_bcast<0>(phi), quad_bcast<0>(plo)), c1 = wide(quad_bcast<1>(phi), quad_bcast<1>(plo));
          const unsigned long long c2 = wide(quad_bcast<2>(phi), quad_bcast<2>(plo)), c3 = wide(quad_bcast<3>(phi), quad_bcast<3>(plo));
          int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
#pragma unroll 1
          for (int bb = 0; bb < nbox; bb++) {
            const float4 A = boxA(bb);
            const unsigned long long q = packed_key(norm(v3(A.x, A.y, A.z) - s.footc) - keyC, l * nbox + bb);
            r0 += q < c0 ? 1 : 0; r1 += q < c1 ? 1 : 0; r2 += q < c2 ? 1 : 0; r3 += q < c3 ? 1 : 0;
          }
          r0 = quad_sum_i(r0); r1 = quad_sum_i(r1); r2 = quad_sum_i(r2); r3 = quad_sum_i(r3);
          const int rk = l == 0 ? r0 : (l == 1 ? r1 : (l == 2 ? r2 : r3));
          alive = penn & !(cutting & (rk >= maxp));
        }
        // the table's last entry in MJX's order, and whether the newcomer sorts before it
        const bool room = npen < kMaxPenQ;
        int wi = 0; float wd = pen[0].dist; unsigned long long wk = ordkey(pen[0]);
#pragma unroll
        for (int i = 1; i < kMaxPenQ; i++) {
          const unsigned long long ki = ordkey(pen[i]);
          const bool later = (pen[i].dist > wd) | ((pen[i].dist == wd) & (ki > wk));
          wd = later ? pen[i].dist : wd; wk = later ? ki : wk; wi = later ? i : wi;
        }
        const bool better = (pp.dist < wd) | ((pp.dist == wd) & (ordkey(pp) < wk));
        const bool put = alive & (room | better);
        const bool swap = put & !room;
        // a full table loses entry wi: the entries behind it move up (registers by selects, point / normal in the slot records) ...
#pragma unroll
        for (int i = 0; i + 1 < kMaxPenQ; i++) {
          const bool mv = swap & (i >= wi);
          pen[i].dist = mv ? pen[i + 1].dist : pen[i].dist; pen[i].key = mv ? pen[i + 1].key : pen[i].key; pen[i].idx = mv ? pen[i + 1].idx : pen[i].idx;
          if (mv) {
#pragma unroll
            for (int f = 7; f <= 12; f++) { const float v = slots.at(i + 1, f); slots.at(i, f) = v; }
          }
        }
        // ... and the newcomer goes behind them (it has the highest pair index so far)
        const int at = room ? npen : kMaxPenQ - 1;
#pragma unroll
        for (int i = 0; i < kMaxPenQ; i++) { const bool hit = put & (i == at); pen[i].dist = hit ? pp.dist : pen[i].dist; pen[i].key = hit ? pp.key : pen[i].key; pen[i].idx = hit ? pp.idx : pen[i].idx; }
        if (put) park(at, pw, nw);
        npen += (put & room) ? 1 : 0;
      }
      s.pen_overflow |= npenetr > kMaxPenQ;
    }
    PG_TICK(s, 13);
    if (__ballot(npen > 0) == 0ull) return;
    // wave-uniform number of candidate columns that are in use anywhere
    int ncol = 0;
#pragma unroll
    for (int i = 0; i < kMaxPenQ; i++) if (__ballot(npen > i) != 0ull) ncol = i + 1;
    // candidate table of the whole quad (replicated): depth of lane j's i-th penetrating pair (columns in use only); the keys
    // and indices of the table are fetched by the exact-rank pass alone
    float cdist[4][kMaxPenQ]; int crank[4][kMaxPenQ];
#pragma unroll
    for (int i = 0; i < kMaxPenQ; i++) {
#pragma unroll
      for (int j = 0; j < 4; j++) { cdist[j][i] = 1.f; crank[j][i] = 0; }
      if (i >= ncol) continue;
      cdist[0][i] = quad_bcast<0>(pen[i].dist); cdist[1][i] = quad_bcast<1>(pen[i].dist); cdist[2][i] = quad_bcast<2>(pen[i].dist); cdist[3][i] = quad_bcast<3>(pen[i].dist);
    }
    bool need_exact = false;
    unsigned nearmask = 0u;
    if (broad) {
      // pass 2a: cheap conservative test.  rank(p) < #pairs with key <= max candidate key; if that count (over the
      // 400 pairs, quad-summed) is <= max_geom_pairs, every candidate survives MJX's top-k and the exact ranks are
      // not needed.  d^2 against a slightly inflated threshold: over-counting keeps the test conservative.
      float kmax = -3.0e38f;
#pragma unroll
      for (int i = 0; i < kMaxPenQ; i++) kmax = fmaxf(kmax, pen[i].dist < 0.f ? pen[i].key : -3.0e38f);
      kmax = quad_max(kmax);       // two butterfly steps instead of four leg broadcasts (in the hex / oct layouts: 3 rotations + 3 selects each)
      const float thr = (kmax + keyC) * 1.000002f + 1e-7f, thr2 = kmax > -1.0e38f ? thr * thr : -1.f;
      // The count is a PROOF (every candidate survives the max_geom_pairs cut), and a proof can be carried over: if an earlier substep of this
      // launch counted, with every foot at f_ref and a radius R, no more than max_geom_pairs pairs, then by the triangle inequality every box
      // within thr of a foot that has moved by d <= R - thr is among the boxes counted then - the count of this substep cannot exceed that one.
      // On the shipped terrains the count is ~3 of the 25 allowed, so R is generous (thr + 0.1 m) and the 100-box pass runs once per launch.
      {
        const bool holds = (thr2 < 0.f) | ((c2_thr >= 0.f) & ((norm(s.footc - c2_foot) + thr) * 1.0001f + 1e-5f <= c2_thr));      // own foot
#ifdef PGTT_TIME
        s.cyc[26] += __ballot(!holds) == 0ull ? 1.f : 0.f;       // substeps in which the wave carries the proof over
#endif
        if (__ballot(!holds) == 0ull) goto counted;           // every foot of every env of the wave: need_exact stays false
      }
      {
      int cnt = 0;
      int cnt_r = 0;
      const float thr_r = thr + 0.1f, thr_r2 = thr_r * thr_r;
      static_assert(PGTT_MAX_BOX <= 128, "hex layout: the <= 32 boxes of a sub-lane are one mask word");
#pragma unroll 4
      for (int t = 0, b = lane_sub(); b < nbox; t++, b += kSubs) {     // hex / oct: boxes go round the sub-lanes
        const float4 A = boxA(b);
        V3 dv = v3(A.x, A.y, A.z) - s.footc;
        const float d2 = dot(dv, dv);
        const bool in = d2 <= thr2;
        cnt += in ? 1 : 0;
        cnt_r += d2 <= thr_r2 ? 1 : 0;
        if (kSubs == 4) nearmask |= (in ? 1u : 0u) << t;     // own boxes that can sort before a candidate (pass 2b)
      }
      need_exact = quad_sum_i(sub_sum_i(cnt)) > maxp;
#ifdef PGTT_TIME
      s.cyc[25] += (float)quad_sum_i(sub_sum_i(cnt));      // pairs counted by the passes of this launch (own env)
#endif
      // remember the wider count when it is a proof as well (an env without a candidate has thr2 < 0 and needs none)
      c2_foot = s.footc;
      c2_thr = (thr2 >= 0.f && quad_sum_i(sub_sum_i(cnt_r)) <= maxp) ? thr_r : -1.f;
      }
      counted:;
    }
    if (broad && __ballot(need_exact) != 0ull) {
      // pass 2b (rare, but the wave that takes it is the one the launch waits for): exact broad-phase rank = number of
      // the 400 (foot, box) pairs that sort before the candidate by (key, index).  The pair (key, index) is packed into
      // one 64-bit integer (key through the order-preserving float -> uint map; keys are never -0 or NaN), so a
      // candidate costs one compare and one add per box; every lane counts over its own foot's pairs (hex: the boxes
      // go round the sub-lanes), the sums over the lanes of the env give the ranks.
      auto packed = [](float key, int idx) { return packed_key(key, idx); };
      unsigned long long cpk[4][kMaxPenQ];
#pragma unroll
      for (int i = 0; i < kMaxPenQ; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) cpk[j][i] = ~0ull;
        if (i >= ncol) continue;
        cpk[0][i] = packed(quad_bcast<0>(pen[i].key), quad_bcast<0>(pen[i].idx)); cpk[1][i] = packed(quad_bcast<1>(pen[i].key), quad_bcast<1>(pen[i].idx));
        cpk[2][i] = packed(quad_bcast<2>(pen[i].key), quad_bcast<2>(pen[i].idx)); cpk[3][i] = packed(quad_bcast<3>(pen[i].key), quad_bcast<3>(pen[i].idx));
      }
      auto rank_against = [&](unsigned long long pk) {
#pragma unroll
        for (int i = 0; i < kMaxPenQ; i++) {
          if (i >= ncol) continue;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            // cnt += (pk < cpk) as compare + add-with-carry through VCC: written out because the compiler keeps the 16
            // compare masks of a box alive in SGPR pairs and spills them through v_writelane (a borrow chain
            // v_sub_co / v_subb_co / v_addc_co instead of the 64-bit compare measured the same)
            asm("v_cmp_lt_u64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(crank[j][i]) : "v"(pk), "v"(cpk[j][i]) : "vcc");
          }
        }
      };
      if (kSubs == 1) {
        float4 An = boxA(0);
#pragma unroll 1
        for (int b = 0; b < nbox; b++) {
          const float4 A = An;
          const int bn = b + 1 < nbox ? b + 1 : b;               // next row, read while this one is ranked (never a record past the variant's own: the table ends with the last variant's last box)
          An = boxA(bn);
          rank_against(packed(norm(v3(A.x, A.y, A.z) - s.footc) - keyC, l * nbox + b));
        }
      } else if (kSubs == 2) {
        // oct layout: even boxes in sub-lane 0, odd ones in sub-lane 1 (an odd box count leaves sub-lane 1 a last turn without a box)
#pragma unroll 1
        for (int b0 = 0; b0 < nbox; b0 += 2) {
          const int b = b0 + (int)(threadIdx.x & 1);
          const bool have = b < nbox;
          const float4 A = boxA(have ? b : 0);
          const unsigned long long pk = packed(norm(v3(A.x, A.y, A.z) - s.footc) - keyC, l * nbox + b);
          rank_against(have ? pk : ~0ull);
        }
      } else {
        // hex layout: only the own boxes that pass 2a found at least as close as the farthest candidate can sort before a
        // candidate (nearmask, a superset: 2-4 of a sub-lane's 25 boxes); every lane pops its own, the loop ends by ballot
        unsigned mk = nearmask;
#pragma unroll 1
        for (;;) {
          if (__ballot(mk != 0u) == 0ull) break;
          const bool have = mk != 0u;
          const int t = have ? __ffs(mk) - 1 : 0;
          mk &= mk - 1u;                                            // 0 stays 0
          const int b = (int)(threadIdx.x & 3) + 4 * t;
          const float4 A = boxA(b);
          const unsigned long long pk = packed(norm(v3(A.x, A.y, A.z) - s.footc) - keyC, l * nbox + b);
          rank_against(have ? pk : ~0ull);                          // ~0 sorts after every candidate
        }
      }
#pragma unroll
      for (int i = 0; i < kMaxPenQ; i++) {
        if (i >= ncol) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) crank[j][i] = quad_sum_i(sub_sum_i(crank[j][i]));
      }
    }
    else {
      // Exact ranks were not needed for the max_geom_pairs cut, but lax.top_k is stable and MJX's narrow phase runs in
      // broad-phase order, so EQUAL depths at the max_contact_points cut are still decided by rank (two boxes with a common
      // top face under one foot give bit-identical depths).  The order of two candidates by rank is the order of their
      // (key, pair index) pairs, and the pair index grows with the scan order of the table: when some env of the wave has
      // more penetrating pairs than slots, the rank column is filled with the keys themselves (order-preserving float ->
      // uint map; the selection compares ranks as unsigned numbers) - no tables beyond the ones that are live anyway.
      const int npn = quad_sum_i(npen);       // penetrating pairs of the env = the table entries with a negative depth
      const int nslot0 = (maxc > -1 && maxc < 4) ? maxc : 4;
      if (broad && __ballot(npn > nslot0) != 0ull) {
#pragma unroll
        for (int i = 0; i < kMaxPenQ; i++) {
          if (i >= ncol) continue;
          const unsigned u = __float_as_uint(pen[i].key);
          const int mono = (int)(u ^ ((unsigned)((int)u >> 31) | 0x80000000u));
          crank[0][i] = quad_bcast<0>(mono); crank[1][i] = quad_bcast<1>(mono); crank[2][i] = quad_bcast<2>(mono); crank[3][i] = quad_bcast<3>(mono);
        }
      }
    }
    PG_TICK(s, 14);
    // selection of the max_contact_points deepest survivors of the env (ties: lower broad-phase rank first, then the
    // scan order leg-major): MJX's sequential top-k picks exactly the pairs that fewer than max_contact_points others
    // beat, so every lane only ranks its OWN pairs against the table — no selection rounds.
    const int nslot = (maxc > -1 && maxc < 4) ? maxc : 4;
    // "still in the table" (penetrating and not removed by the max_geom_pairs cut) is formed where an entry is looked at, i.e. for the columns
    // in use only - as a table of sixteen masks it was computed in full and kept alive in SGPR pairs (spilled through v_writelane / v_readlane)
    const bool cut = broad & need_exact;
    auto ok = [&](int j, int i) { return (cdist[j][i] < 0.f) & !(cut & (crank[j][i] >= maxp)); };
    // is the own pair (d, rank r, scan order ord) among the nslot best of the env's table?
    auto selected = [&](float d, int r, bool okm, int ord) {
      int beat = 0;
      // candidate columns outermost: ONE wave-uniform test per column in use instead of one per table entry (the count is an integer sum)
#pragma unroll
      for (int i2 = 0; i2 < kMaxPenQ; i2++) {
        if (i2 >= ncol) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const bool first = (cdist[j][i2] < d) | ((cdist[j][i2] == d) & (((unsigned)crank[j][i2] < (unsigned)r) | ((crank[j][i2] == r) & (j * kMaxPenQ + i2 < ord))));
          beat += (ok(j, i2) & first) ? 1 : 0;
        }
      }
      return okm & (beat < nslot);
    };
    int nb = 0;
    if (kSubs != 4) {
      bool mine[kMaxPenQ];
#pragma unroll
      for (int i = 0; i < kMaxPenQ; i++) {
        if (i >= ncol) { mine[i] = false; continue; }
        // own pair (l, i): values through selects on the leg index
        const float d = sel4(l, cdist[0][i], cdist[1][i], cdist[2][i], cdist[3][i]);
        const int r = l == 0 ? crank[0][i] : (l == 1 ? crank[1][i] : (l == 2 ? crank[2][i] : crank[3][i]));
        const bool okm = (d < 0.f) & !(cut & (r >= maxp));
        mine[i] = selected(d, r, okm, l * kMaxPenQ + i);
      }
      // own selected pairs: park (dist, box, point, normal) in the slot records
#pragma unroll
      for (int i = 0; i < kMaxPenQ; i++) {
        if (i >= ncol) continue;
        if (mine[i]) {
          slots.at(nb, 0) = pen[i].dist;
          slots.at(nb, 20) = __int_as_float(pen[i].idx - l * nbox);
          // point / normal move down from entry i to slot nb <= i (entries below i were consumed already)
#pragma unroll
          for (int f = 7; f <= 12; f++) { const float v = slots.at(i, f); slots.at(nb, f) = v; }
          nb++;
        }
      }
    } else {
      // hex layout: sub-lane q ranks the leg's pair q and moves it - one pass whatever the number of candidate columns.
      // All sub-lanes READ their entry before any of them writes a slot (the LDS executes a wave's accesses in program
      // order), so a pair moving down into the entry of another one is safe.
      const int q = threadIdx.x & 3;
      const float d = sel4(q, pen[0].dist, pen[1].dist, pen[2].dist, pen[3].dist);       // = cdist[l][q], the own leg's column of the table
      auto seli = [&](int a0, int a1, int a2, int a3, int k) { return k == 0 ? a0 : (k == 1 ? a1 : (k == 2 ? a2 : a3)); };
      const int r = seli(seli(crank[0][0], crank[1][0], crank[2][0], crank[3][0], l), seli(crank[0][1], crank[1][1], crank[2][1], crank[3][1], l),
                         seli(crank[0][2], crank[1][2], crank[2][2], crank[3][2], l), seli(crank[0][3], crank[1][3], crank[2][3], crank[3][3], l), q);
      const bool okm = (d < 0.f) & !((broad & need_exact) & (r >= maxp));
      const bool mine_q = (q < ncol) & selected(d, r, okm, l * kMaxPenQ + q);
      const int mi = mine_q ? 1 : 0;
      const int m0 = sub_bcast<0>(mi), m1 = sub_bcast<1>(mi), m2 = sub_bcast<2>(mi), m3 = sub_bcast<3>(mi);
      const int dest = q == 0 ? 0 : (q == 1 ? m0 : (q == 2 ? m0 + m1 : m0 + m1 + m2));      // selected pairs below the own one
      nb = m0 + m1 + m2 + m3;
      float mv[6];
#pragma unroll
      for (int f = 0; f < 6; f++) mv[f] = slots.at(q, 7 + f);
      const int iq = seli(pen[0].idx, pen[1].idx, pen[2].idx, pen[3].idx, q);
      if (mine_q) {
        slots.at(dest, 0) = d;
        slots.at(dest, 20) = __int_as_float(iq - l * nbox);
#pragma unroll
        for (int f = 0; f < 6; f++) slots.at(dest, 7 + f) = mv[f];
      }
    }
    s.nbox = nb;
    PG_TICK(s, 16);
  }
};

// ------------------------------------------------------------------ Newton solver, quad version
struct QSolver {
  const PgttModel* __restrict__ m;
  QSim& s;
  float qb[6], ql[3], Mab[6], Mal[3], gb[6], gl[3], sb[6], sl[3], fcb[6], fcl[3];
  float jar_lim[3], jar0[4];
  float gauss, cost, prev_cost;
  int nslots;     // wave-uniform number of own-box-contact slots in use anywhere in the wave
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 hx_ja, hx_jv, hx_D;    // hex layout: (limit row, plane-contact row) of the own sub-lane for the current line search
  bool lds_slots = false;   // the kernel has slot records in LDS (box terrain)
  bool plane_sub = false;   // hex layout, wave-uniform: the plane contact sits in slot 3 and is worked on by sub-lane 3 (its
                            // Jacobian products, force and Hessian in the same single pass as the box slots of sub-lanes
                            // 0..2 instead of replicated in all four sub-lanes); its four rows are handed back to jar0 / jv0
  // hex layout: the contact this sub-lane works on (box slot = sub-lane index, or the plane contact) is read from its slot
  // record ONCE per solve and kept in registers with its four rows (mjar, mjv; mirrored in the record for the sub-lanes
  // that evaluate them in the line search); the rows r of the box slots that this sub-lane evaluates are read ONCE per
  // line search (ls_ja / ls_jv / ls_D, slot pairs (0,1) and (2,3)) - the Newton loop otherwise re-read ~100 LDS words per trip
  QContact mine; float mjar[4], mjv[4];
  f2 ls_ja[2], ls_jv[2], ls_D[2];
  f2 qd_ja[4], qd_jv[4]; float qd_D[2];     // quad layout: rows of box slots 0, 1 for the current line search (row pairs 01, 23 of each)
  // oct layout: sub-lane 0 evaluates the row pair (0, 1), sub-lane 1 the pair (2, 3) of every constraint of the leg - limit rows
  // (0, 1) / (2, -), the plane contact, box slot k (qd_ja[k] / qd_jv[k] then hold the OWN pair of slot k) - picked once per search
  f2 oc_lim_ja, oc_lim_jv, oc_lim_D, oc_pl_ja, oc_pl_jv; float oc_D[4];
  // oct layout: the same idea with two sub-lanes - the units of work of a leg are its box slots 0 .. nslots-1 plus, while a slot is
  // free (nslots <= 3), the plane contact in slot `plane_slot` = nslots; sub-lane q owns the units q, q + 2 (records and rows stay
  // in LDS), so the per-contact passes of the Newton loop number ceil(units / 2) instead of units
  int plane_slot = 3;
  bool oc_split = false;    // oct layout, wave-uniform: units are shared out between the sub-lanes (two or more box slots in use in the
                            // wave); otherwise both sub-lanes walk the slots like a lane of the quad layout does (plane contact in
                            // registers): with a single slot the hand-over through LDS costs more than the second pass it saves
  PG_INL bool own_on(int k) const { return (k < nslots) | (plane_sub & (k == plane_slot)); }
  PG_INL bool own_any() const { return nslots > 0 || plane_sub; }
  PG_INL int units() const { return nslots + (plane_sub ? 1 : 0); }
  // trips of a "for every contact this lane works on" loop and the contact of trip k0
  PG_INL int own_trips() const { return kSubs == 1 ? nslots : (kSubs == 4 ? (own_any() ? 1 : 0) : (oc_split ? (units() + 1) / 2 : nslots)); }
  PG_INL int own_slot(int k0) const { return (kSubs == 1 || (kSubs == 2 && !oc_split)) ? k0 : lane_sub() + kSubs * k0; }
  bool any_lim, any_con0;   // wave-uniform: some lane has an active joint-limit row / an active plane contact.
                            // Inactive rows have D = 0 and aref = 0: they add exact zeros, so skipping them is bit-neutral.
  const BoxSlots slots;
#ifdef PGTT_TRACE
  float last_alpha = 0.f;
  PG_INL void rec(float v) { s.rec(v); }
  PG_INL void rec_state(float tag) {
    rec(tag); rec(cost); rec(gauss); rec(prev_cost); rec(last_alpha);
    for (int i = 0; i < 6; i++) rec(qb[i]);
    for (int i = 0; i < 3; i++) rec(ql[i]);
    for (int i = 0; i < 6; i++) rec(gb[i]);
    for (int i = 0; i < 3; i++) rec(gl[i]);
    for (int i = 0; i < 6; i++) rec(sb[i]);
    for (int i = 0; i < 3; i++) rec(sl[i]);
    for (int i = 0; i < 6; i++) rec(fcb[i]);
    for (int i = 0; i < 3; i++) rec(fcl[i]);
    for (int i = 0; i < 4; i++) rec(jar0[i]);
    for (int i = 0; i < 3; i++) rec(jar_lim[i]);
  }
#endif

  PG_INL QSolver(const PgttModel* m_, QSim& s_, const BoxSlots& sl_) : m(m_), s(s_), slots(sl_) {}

  // spatial motion of the own calf generated by the generalised vector (xb, xl)
  PG_INL S6 twist(const float* xb, const float* xl) const {
    S6 t{v3(0, 0, 0), v3(xb[0], xb[1], xb[2])};
#pragma unroll
    for (int k = 0; k < 3; k++) t = t + s.cdr[k] * xb[3 + k];
#pragma unroll
    for (int k = 0; k < 3; k++) t = t + s.cdl[k] * xl[k];
    return t;
  }
  PG_INL void con_jx(const QContact& c, S6 tw, float* out4) const {
    V3 vp = tw.l + cross(tw.a, c.off);
    float t[3] = {dot(c.fr[0], vp), dot(c.fr[1], vp), dot(c.fr[2], vp)};
    out4[0] = t[0] + c.mu * t[1]; out4[1] = t[0] - c.mu * t[1]; out4[2] = t[0] + c.mu * t[2]; out4[3] = t[0] - c.mu * t[2];
  }

  PG_INL void init(const float* q0b, const float* q0l) {
#pragma unroll
    for (int i = 0; i < 6; i++) qb[i] = q0b[i];
#pragma unroll
    for (int k = 0; k < 3; k++) ql[k] = q0l[k];
    qarrow_mul(s.M, qb, ql, Mab, Mal);
#pragma unroll
    for (int k = 0; k < 3; k++) jar_lim[k] = s.lim_sign[k] * ql[k] * (s.lim_active[k] ? 1.f : 0.f) - s.lim_aref[k];
    const S6 tw = twist(qb, ql);
#pragma unroll
    for (int r = 0; r < 4; r++) jar0[r] = 0.f;
    if (any_con0 && !plane_sub) {
      float jx[4];
      con_jx(s.con0, tw, jx);
#pragma unroll
      for (int r = 0; r < 4; r++) jar0[r] = (s.con0.row_active ? jx[r] : 0.f) - s.con0.aref[r];
    }
    float pv[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < own_trips(); k0++) {
      const int k = own_slot(k0);                                   // hex: every sub-lane works on ITS slot, in one pass; oct: on its one or two
      if (!own_on(k)) continue;
      const QContact cn = kSubs != 4 ? slots.load(k) : mine;
      float jx[4];
      con_jx(cn, tw, jx);
#pragma unroll
      for (int r = 0; r < 4; r++) { pv[r] = (cn.row_active ? jx[r] : 0.f) - cn.aref[r]; slots.jar(k, r) = pv[r]; if (kSubs == 4) mjar[r] = pv[r]; }
    }
    if (kSubs == 4 && plane_sub) {
#pragma unroll
      for (int r = 0; r < 4; r++) jar0[r] = sub_bcast<3>(pv[r]);
    }
    cost = INFINITY; prev_cost = 0.f;
  }

  PG_INL void update_constraint() {
    float csum = 0.f, pb[6];
    S6 Fs{v3(0, 0, 0), v3(0, 0, 0)};       // spatial force of the own foot's contacts about the COM
#pragma unroll
    for (int k = 0; k < 3; k++) fcl[k] = 0.f;
    if (any_lim) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        float ja = jar_lim[k];
        float f = ja < 0.f ? -s.lim_D[k] * ja : 0.f;
        fcl[k] = s.lim_sign[k] * f;
        csum += ja < 0.f ? s.lim_D[k] * ja * ja : 0.f;
      }
    }
    auto add_contact = [](const QContact& cn, const float* ja4, S6& F, float& cs) {
      float f[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float ja = ja4[r];
        f[r] = ja < 0.f ? -cn.D * ja : 0.f;
        cs += ja < 0.f ? cn.D * ja * ja : 0.f;
      }
      float g[3] = {f[0] + f[1] + f[2] + f[3], cn.mu * (f[0] - f[1]), cn.mu * (f[2] - f[3])};
      V3 fw = cn.fr[0] * g[0] + cn.fr[1] * g[1] + cn.fr[2] * g[2];
      F.l = F.l + fw; F.a = F.a + cross(cn.off, fw);
    };
    if (any_con0 && !plane_sub) add_contact(s.con0, jar0, Fs, csum);
    if (kSubs == 1 || (kSubs == 2 && !oc_split)) {
      for (int k = 0; k < nslots; k++) {
        const QContact cn = slots.load(k);
        float ja4[4] = {slots.jar(k, 0), slots.jar(k, 1), slots.jar(k, 2), slots.jar(k, 3)};
        add_contact(cn, ja4, Fs, csum);
      }
    } else if (kSubs == 2) {
      if (own_any()) {
        // oct layout: each sub-lane forms the force of its own contacts; the sub-lane sum gives both the leg's total
        S6 Fd{v3(0, 0, 0), v3(0, 0, 0)}; float cd = 0.f;
        for (int k0 = 0; k0 < own_trips(); k0++) {
          const int k = own_slot(k0);
          if (!own_on(k)) continue;
          const QContact cn = slots.load(k);
          float ja4[4] = {slots.jar(k, 0), slots.jar(k, 1), slots.jar(k, 2), slots.jar(k, 3)};
          add_contact(cn, ja4, Fd, cd);
        }
        Fs.l = Fs.l + v3(sub_sum(Fd.l.x), sub_sum(Fd.l.y), sub_sum(Fd.l.z));
        Fs.a = Fs.a + v3(sub_sum(Fd.a.x), sub_sum(Fd.a.y), sub_sum(Fd.a.z));
        csum += sub_sum(cd);
      }
    } else if (own_any()) {
      // hex layout: the owner of a slot forms its force; the sub-lane sum gives every lane the leg's total
      S6 Fd{v3(0, 0, 0), v3(0, 0, 0)}; float cd = 0.f;
      for (int k0 = 0; k0 < 1; k0++) {
        const int k = (int)(threadIdx.x & 3);      // hex: every sub-lane works on ITS slot, in one pass
        if (!own_on(k)) continue;
        add_contact(mine, mjar, Fd, cd);
      }
      Fs.l = Fs.l + v3(sub_sum(Fd.l.x), sub_sum(Fd.l.y), sub_sum(Fd.l.z));
      Fs.a = Fs.a + v3(sub_sum(Fd.a.x), sub_sum(Fd.a.y), sub_sum(Fd.a.z));
      csum += sub_sum(cd);
    }
    pb[0] = Fs.l.x; pb[1] = Fs.l.y; pb[2] = Fs.l.z;
#pragma unroll
    for (int k = 0; k < 3; k++) { pb[3 + k] = dot6(s.cdr[k], Fs); fcl[k] += dot6(s.cdl[k], Fs); }
#pragma unroll
    for (int i = 0; i < 6; i++) fcb[i] = quad_sum(pb[i]);
    float gbase = 0.f, gleg = 0.f;
#pragma unroll
    for (int i = 0; i < 6; i++) gbase += (Mab[i] - s.qfs_b[i]) * (qb[i] - s.qas_b[i]);
#pragma unroll
    for (int k = 0; k < 3; k++) gleg += (Mal[k] - s.qfs_l[k]) * (ql[k] - s.qas_l[k]);
    gauss = 0.5f * (gbase + quad_sum(gleg));
    prev_cost = cost;
    cost = 0.5f * quad_sum(csum) + gauss;
  }

  PG_INL void update_gradient() {
#ifdef PGTT_EFFORT
    {
      unsigned pat = 0u;
#pragma unroll
      for (int k = 0; k < 3; k++) pat |= (jar_lim[k] < 0.f ? 1u : 0u) << k;
#pragma unroll
      for (int r = 0; r < 4; r++) pat |= (jar0[r] < 0.f ? 1u : 0u) << (3 + r);
      if (kSubs == 4) {
#pragma unroll
        for (int r = 0; r < 4; r++) pat |= ((own_on((int)(threadIdx.x & 3)) && mjar[r] < 0.f) ? 1u : 0u) << (7 + r);
      }
      s.eff_hess += 1; s.eff_hess_same += __ballot(pat != s.eff_pat) == 0ull ? 1 : 0;
      s.eff_pat = pat;
    }
#endif
#pragma unroll
    for (int i = 0; i < 6; i++) gb[i] = Mab[i] - s.qfs_b[i] - fcb[i];
#pragma unroll
    for (int k = 0; k < 3; k++) gl[k] = Mal[k] - s.qfs_l[k] - fcl[k];
    QArrow H;
    float Gbb[21];
#pragma unroll
    for (int i = 0; i < 21; i++) Gbb[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 18; i++) H.lb[i] = s.M.lb[i];
#pragma unroll
    for (int i = 0; i < 6; i++) H.ll[i] = s.M.ll[i];
#pragma unroll
    for (int k = 0; k < 3; k++) H.ll[tri(k, k)] += jar_lim[k] < 0.f ? s.lim_D[k] : 0.f;
    auto add_hessian = [&](const QContact& cn, const float* ja4, float* Gt, float* lbt, float* llt) {
      float w[4];
#pragma unroll
      for (int r = 0; r < 4; r++) w[r] = ja4[r] < 0.f ? cn.D : 0.f;
      float mu = cn.mu;
      float W00 = w[0] + w[1] + w[2] + w[3], W01 = mu * (w[0] - w[1]), W02 = mu * (w[2] - w[3]);
      float W11 = mu * mu * (w[0] + w[1]), W22 = mu * mu * (w[2] + w[3]);
      // world-frame weight A = F^T Wc F (symmetric 3x3), then G = col^T A col over the 9 Jacobian columns
      V3 r0 = cn.fr[0] * W00 + cn.fr[1] * W01 + cn.fr[2] * W02, r1 = cn.fr[0] * W01 + cn.fr[1] * W11, r2 = cn.fr[0] * W02 + cn.fr[2] * W22;
      V3 Ax = v3(cn.fr[0].x * r0.x + cn.fr[1].x * r1.x + cn.fr[2].x * r2.x, cn.fr[0].x * r0.y + cn.fr[1].x * r1.y + cn.fr[2].x * r2.y, cn.fr[0].x * r0.z + cn.fr[1].x * r1.z + cn.fr[2].x * r2.z);
      V3 Ay = v3(Ax.y, cn.fr[0].y * r0.y + cn.fr[1].y * r1.y + cn.fr[2].y * r2.y, cn.fr[0].y * r0.z + cn.fr[1].y * r1.z + cn.fr[2].y * r2.z);
      V3 Az = v3(Ax.z, Ay.z, cn.fr[0].z * r0.z + cn.fr[1].z * r1.z + cn.fr[2].z * r2.z);
      V3 col[9], y[9];
      col[0] = v3(1, 0, 0); col[1] = v3(0, 1, 0); col[2] = v3(0, 0, 1);
#pragma unroll
      for (int k = 0; k < 3; k++) { col[3 + k] = s.cdr[k].l + cross(s.cdr[k].a, cn.off); col[6 + k] = s.cdl[k].l + cross(s.cdl[k].a, cn.off); }
      // The first three columns are the identity: y[0..2] are the columns of A, an entry G[i][j] with j < 3 is component
      // j of y[i] (A is symmetric; same products in the same order as dot(col[i], y[j])) - 24 of the 45 entries cost no
      // arithmetic (the compiler may not drop the 0 * x terms of the generic dot products by itself)
      y[0] = Ax; y[1] = Ay; y[2] = Az;
#pragma unroll
      for (int k = 3; k < 9; k++) y[k] = v3(dot(Ax, col[k]), dot(Ay, col[k]), dot(Az, col[k]));
      auto comp = [](V3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); };
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) Gt[tri(i, j)] += j < 3 ? comp(y[i], j) : dot(col[i], y[j]);
#pragma unroll
      for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int k = 0; k < 6; k++) lbt[i * 6 + k] += k < 3 ? comp(y[6 + i], k) : dot(col[6 + i], y[k]);
#pragma unroll
        for (int j = 0; j <= i; j++) llt[tri(i, j)] += dot(col[6 + i], y[6 + j]);
      }
    };
    auto add_hessian_set = [&](const QContact& cn, const float* ja4, float* Gt, float* lbt, float* llt) {
      float w[4];
#pragma unroll
      for (int r = 0; r < 4; r++) w[r] = ja4[r] < 0.f ? cn.D : 0.f;
      float mu = cn.mu;
      float W00 = w[0] + w[1] + w[2] + w[3], W01 = mu * (w[0] - w[1]), W02 = mu * (w[2] - w[3]);
      float W11 = mu * mu * (w[0] + w[1]), W22 = mu * mu * (w[2] + w[3]);
      // world-frame weight A = F^T Wc F (symmetric 3x3), then G = col^T A col over the 9 Jacobian columns
      V3 r0 = cn.fr[0] * W00 + cn.fr[1] * W01 + cn.fr[2] * W02, r1 = cn.fr[0] * W01 + cn.fr[1] * W11, r2 = cn.fr[0] * W02 + cn.fr[2] * W22;
      V3 Ax = v3(cn.fr[0].x * r0.x + cn.fr[1].x * r1.x + cn.fr[2].x * r2.x, cn.fr[0].x * r0.y + cn.fr[1].x * r1.y + cn.fr[2].x * r2.y, cn.fr[0].x * r0.z + cn.fr[1].x * r1.z + cn.fr[2].x * r2.z);
      V3 Ay = v3(Ax.y, cn.fr[0].y * r0.y + cn.fr[1].y * r1.y + cn.fr[2].y * r2.y, cn.fr[0].y * r0.z + cn.fr[1].y * r1.z + cn.fr[2].y * r2.z);
      V3 Az = v3(Ax.z, Ay.z, cn.fr[0].z * r0.z + cn.fr[1].z * r1.z + cn.fr[2].z * r2.z);
      V3 col[9], y[9];
      col[0] = v3(1, 0, 0); col[1] = v3(0, 1, 0); col[2] = v3(0, 0, 1);
#pragma unroll
      for (int k = 0; k < 3; k++) { col[3 + k] = s.cdr[k].l + cross(s.cdr[k].a, cn.off); col[6 + k] = s.cdl[k].l + cross(s.cdl[k].a, cn.off); }
      // The first three columns are the identity: y[0..2] are the columns of A, an entry G[i][j] with j < 3 is component
      // j of y[i] (A is symmetric; same products in the same order as dot(col[i], y[j])) - 24 of the 45 entries cost no
      // arithmetic (the compiler may not drop the 0 * x terms of the generic dot products by itself)
      y[0] = Ax; y[1] = Ay; y[2] = Az;
#pragma unroll
      for (int k = 3; k < 9; k++) y[k] = v3(dot(Ax, col[k]), dot(Ay, col[k]), dot(Az, col[k]));
      auto comp = [](V3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); };
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) Gt[tri(i, j)] = j < 3 ? comp(y[i], j) : dot(col[i], y[j]);
#pragma unroll
      for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int k = 0; k < 6; k++) lbt[i * 6 + k] = k < 3 ? comp(y[6 + i], k) : dot(col[6 + i], y[k]);
#pragma unroll
        for (int j = 0; j <= i; j++) llt[tri(i, j)] = dot(col[6 + i], y[6 + j]);
      }
    };
    if (any_con0 && !plane_sub) add_hessian(s.con0, jar0, Gbb, H.lb, H.ll);
    if (kSubs == 1 || (kSubs == 2 && !oc_split)) {
      for (int k = 0; k < nslots; k++) {
        const QContact cn = slots.load(k);
        float ja4[4] = {slots.jar(k, 0), slots.jar(k, 1), slots.jar(k, 2), slots.jar(k, 3)};
        add_hessian(cn, ja4, Gbb, H.lb, H.ll);
      }
    } else if (kSubs == 2) {
      if (own_any()) {
        // oct layout: the 45 entries of the own contacts (a record that is not in use has finite fields and gets weight 0: exact
        // zeros), one or two passes, then the sub-lane sums
        float Gd[21], lbd[18], lld[6];
        {
          const int k = lane_sub();
          QContact cn = slots.load(k);
          cn.D = own_on(k) ? cn.D : 0.f;
          float ja4[4] = {slots.jar(k, 0), slots.jar(k, 1), slots.jar(k, 2), slots.jar(k, 3)};
          add_hessian_set(cn, ja4, Gd, lbd, lld);
        }
        if (units() > 2) {
          const int k = lane_sub() + 2;
          QContact cn = slots.load(k);
          cn.D = own_on(k) ? cn.D : 0.f;
          float ja4[4] = {slots.jar(k, 0), slots.jar(k, 1), slots.jar(k, 2), slots.jar(k, 3)};
          add_hessian(cn, ja4, Gd, lbd, lld);
        }
#pragma unroll
        for (int i = 0; i < 21; i++) Gbb[i] += sub_sum(Gd[i]);
#pragma unroll
        for (int i = 0; i < 18; i++) H.lb[i] += sub_sum(lbd[i]);
#pragma unroll
        for (int i = 0; i < 6; i++) H.ll[i] += sub_sum(lld[i]);
      }
    } else if (own_any()) {
      // hex layout: the owner of a slot forms its 45 Hessian entries; the sub-lane sums give every lane the leg's total
      float Gd[21], lbd[18], lld[6];
      {
        // a sub-lane whose slot is not in use contributes exact zeros: its weight is forced to 0 (every field of a slot record
        // is finite, BoxSlots::clear_all) instead of branching round the products and pre-clearing 45 accumulators
        QContact cn = mine;
        cn.D = own_on((int)(threadIdx.x & 3)) ? cn.D : 0.f;
        add_hessian_set(cn, mjar, Gd, lbd, lld);
      }
#pragma unroll
      for (int i = 0; i < 21; i++) Gbb[i] += sub_sum(Gd[i]);
#pragma unroll
      for (int i = 0; i < 18; i++) H.lb[i] += sub_sum(lbd[i]);
#pragma unroll
      for (int i = 0; i < 6; i++) H.ll[i] += sub_sum(lld[i]);
    }
#pragma unroll
    for (int i = 0; i < 21; i++) H.bb[i] = s.M.bb[i] + quad_sum(Gbb[i]);
#ifdef PGTT_TRACE
    rec(300.f);
    for (int i = 0; i < 21; i++) rec(H.bb[i]);
    for (int i = 0; i < 21; i++) rec(s.M.bb[i]);
    for (int i = 0; i < 18; i++) rec(H.lb[i]);
    for (int i = 0; i < 6; i++) rec(H.ll[i]);
#endif
    qarrow_factor(H);
    float mgb[6], mgl[3];
    qarrow_solve(H, gb, gl, mgb, mgl);
#pragma unroll
    for (int i = 0; i < 6; i++) sb[i] = -mgb[i];
#pragma unroll
    for (int k = 0; k < 3; k++) sl[k] = -mgl[k];
  }

  struct LSPoint { float alpha, cost, d0, d1; };

  // Two constraint rows at NA trial steps, in packed fp32 (v_pk_*: two rows per instruction).  The quadratic pieces
  // h = D * (ja^2/2, jv ja, jv^2/2) are formed once and added where the row is active (ja + alpha jv < 0); m * h with
  // m in {0, 1} is exact, so the sums are those of an un-fused evaluation (even and odd rows in separate chains).
  // mk = (ja + alpha jv < 0) ? 1 : 0 for two rows WITHOUT compares and selects: t = fma(alpha, -2^64 jv, -2^64 ja) is exactly
  // -2^64 (ja + alpha jv) (binary scaling commutes with the rounding of the fma), and the saturating packed multiply
  // clamp(t * 2^100) maps every t > 0 - down to the smallest scaled denormal - to 1 and t <= 0 / NaN to 0
  // (tools/probes/satmul_probe.hip: checked against `x < 0` on 2^20 bit patterns incl. denormals, +-0, NaN, +-inf).
  // Two packed instructions per row pair and trial step instead of a packed fma, two compares and two selects.
  PG_INL static f2 sat_mul(f2 a, f2 b) { f2 r; asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b)); return r; }
  template <int NA, bool COST = true>
  PG_INL static void ls_row2(f2 ja, f2 jv, float D, const float* al, f2 (*q)[3]) { ls_row2d<NA, COST>(ja, jv, f2{D, D}, al, q); }
  // COST = false: the derivatives alone (the bracketing rounds only look at d0 / d1; the costs of the two points the search
  // ends with are evaluated once, afterwards - same function of alpha, same bits)
  template <int NA, bool COST = true>
  PG_INL void ls_points(const float* al, const float* jv_lim, const float* jv0, float qg0, float qg1, float qg2, LSPoint* out) const {
    f2 q[NA][3];
#pragma unroll
    for (int a = 0; a < NA; a++) { q[a][0] = f2{0.f, 0.f}; q[a][1] = f2{0.f, 0.f}; q[a][2] = f2{0.f, 0.f}; }
    if (kSubs == 1) {
      if (any_lim) {
        // the three limit rows have their own D each: pairs (row0, row1) and (row2, empty)
        const f2 ja01{jar_lim[0], jar_lim[1]}, jv01{jv_lim[0], jv_lim[1]}, D01{s.lim_D[0], s.lim_D[1]};
        const f2 ja2{jar_lim[2], 0.f}, jv2{jv_lim[2], 0.f}, D2{s.lim_D[2], 0.f};
        ls_row2d<NA, COST>(ja01, jv01, D01, al, q);
        ls_row2d<NA, COST>(ja2, jv2, D2, al, q);
      }
      if (any_con0) {
        ls_row2<NA, COST>(f2{jar0[0], jar0[1]}, f2{jv0[0], jv0[1]}, s.con0.D, al, q);
        ls_row2<NA, COST>(f2{jar0[2], jar0[3]}, f2{jv0[2], jv0[3]}, s.con0.D, al, q);
      }
      // the rows of box slots 0 and 1 (all there is on the shipped terrains, bar a foot on a seam) were read from their LDS
      // records once for this line search (qd_*): a round does not wait for an LDS round trip; slots 2, 3 are read in place
      if (nslots > 0) { ls_row2<NA, COST>(qd_ja[0], qd_jv[0], qd_D[0], al, q); ls_row2<NA, COST>(qd_ja[1], qd_jv[1], qd_D[0], al, q); }
      if (nslots > 1) { ls_row2<NA, COST>(qd_ja[2], qd_jv[2], qd_D[1], al, q); ls_row2<NA, COST>(qd_ja[3], qd_jv[3], qd_D[1], al, q); }
      for (int k = 2; k < nslots; k++) {
        const float Dk = slots.at(k, 2);
        ls_row2<NA, COST>(f2{slots.jar(k, 0), slots.jar(k, 1)}, f2{slots.jv(k, 0), slots.jv(k, 1)}, Dk, al, q);
        ls_row2<NA, COST>(f2{slots.jar(k, 2), slots.jar(k, 3)}, f2{slots.jv(k, 2), slots.jv(k, 3)}, Dk, al, q);
      }
    } else if (kSubs == 2) {
      if (any_lim) ls_row2d<NA, COST>(oc_lim_ja, oc_lim_jv, oc_lim_D, al, q);
      if (any_con0 && !plane_sub) ls_row2<NA, COST>(oc_pl_ja, oc_pl_jv, s.con0.D, al, q);
      const int nu = units();
      if (nu > 0) ls_row2<NA, COST>(qd_ja[0], qd_jv[0], oc_D[0], al, q);
      if (nu > 1) ls_row2<NA, COST>(qd_ja[1], qd_jv[1], oc_D[1], al, q);
      if (nu > 2) ls_row2<NA, COST>(qd_ja[2], qd_jv[2], oc_D[2], al, q);
      if (nu > 3) ls_row2<NA, COST>(qd_ja[3], qd_jv[3], oc_D[3], al, q);
    } else {
      // hex layout: sub-lane r evaluates row r of every constraint of its leg (limit row r < 3, pyramid row r of the
      // plane contact and of each box slot); the sums below run over all 16 lanes of the env
      const int r = threadIdx.x & 3;
      // On box terrain these two row pairs are live in nearly every wave: evaluated unconditionally there (rows that are not in use hold
      // zeros and add exact zeros) - a not-taken branch costs a one-wave SIMD ~10 cycles, and a round has a dozen instructions per pair
      if (kTerrainTU || any_lim || any_con0) ls_row2d<NA, COST>(hx_ja, hx_jv, hx_D, al, q);       // own (limit row, plane row), picked once per search
      if (kTerrainTU || nslots > 0) ls_row2d<NA, COST>(ls_ja[0], ls_jv[0], ls_D[0], al, q);
      if (__builtin_expect(nslots > 2, 0)) ls_row2d<NA, COST>(ls_ja[1], ls_jv[1], ls_D[1], al, q);      // three or four box contacts on one foot: rare, out of line
    }
#pragma unroll
    for (int a = 0; a < NA; a++) {
      const float q1 = quad_sum(sub_sum(q[a][1].x + q[a][1].y)) + qg1, q2 = quad_sum(sub_sum(q[a][2].x + q[a][2].y)) + qg2;
      const float alpha = al[a];
      out[a].alpha = alpha;
      if (COST) {
        const float q0 = quad_sum(sub_sum(q[a][0].x + q[a][0].y)) + qg0;
        out[a].cost = alpha * alpha * q2 + alpha * q1 + q0;
      } else {
        out[a].cost = 0.f;
      }
      out[a].d0 = 2.0f * alpha * q2 + q1;
      out[a].d1 = 2.0f * q2 + (q2 == 0.f ? kMinVal : 0.f);
    }
  }
  // same with one D per row
  template <int NA, bool COST = true>
  PG_INL static void ls_row2d(f2 ja, f2 jv, f2 D, const float* al, f2 (*q)[3]) {
    const f2 h0 = D * (0.5f * ja * ja), h1 = D * (jv * ja), h2 = D * (0.5f * jv * jv);
    const f2 jaK = ja * -0x1p64f, jvK = jv * -0x1p64f;
#pragma unroll
    for (int a = 0; a < NA; a++) {
      const f2 t = __builtin_elementwise_fma(f2{al[a], al[a]}, jvK, jaK);
      const f2 mk = sat_mul(t, f2{0x1p100f, 0x1p100f});
      if (COST) q[a][0] = __builtin_elementwise_fma(mk, h0, q[a][0]);
      q[a][1] = __builtin_elementwise_fma(mk, h1, q[a][1]);
      q[a][2] = __builtin_elementwise_fma(mk, h2, q[a][2]);
    }
  }

  // a / b for a denominator in the normal range (the curvature of the line-search quadratic): reciprocal and multiply, the
  // same mantissa arithmetic as the 1-ulp division of this build without its frexp / ldexp range scaling; the product is
  // kept out of fp contraction so that `alpha - a / b` rounds the quotient first, like the division did
  PG_INL static float div_normal(float a, float b) {
#pragma clang fp contract(off)
    return a * __builtin_amdgcn_rcpf(b);
  }

  PG_INL void linesearch(bool frozen) {
    PG_LTICK(s, 27);
    float snb = 0.f, snl = 0.f;
#pragma unroll
    for (int i = 0; i < 6; i++) snb += sb[i] * sb[i];
#pragma unroll
    for (int k = 0; k < 3; k++) snl += sl[k] * sl[k];
    float smag = sqrtf(snb + quad_sum(snl)) * m->meaninertia * 18.0f;
    float gtol = m->tolerance * m->ls_tolerance * smag;
    float mvb[6], mvl[3];
    qarrow_mul(s.M, sb, sl, mvb, mvl);
    float jv_lim[3], jv0[4];
#pragma unroll
    for (int k = 0; k < 3; k++) jv_lim[k] = s.lim_active[k] ? s.lim_sign[k] * sl[k] : 0.f;
    const S6 tws = twist(sb, sl);
#pragma unroll
    for (int r = 0; r < 4; r++) jv0[r] = 0.f;
    if (any_con0 && !plane_sub) {
      float jx[4];
      con_jx(s.con0, tws, jx);
#pragma unroll
      for (int r = 0; r < 4; r++) jv0[r] = s.con0.row_active ? jx[r] : 0.f;
    }
    float pv[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < own_trips(); k0++) {
      const int k = own_slot(k0);                                   // hex: every sub-lane works on ITS slot, in one pass; oct: on its one or two
      if (!own_on(k)) continue;
      const QContact cn = kSubs != 4 ? slots.load(k) : mine;
      float jx[4];
      con_jx(cn, tws, jx);
#pragma unroll
      for (int r = 0; r < 4; r++) { pv[r] = cn.row_active ? jx[r] : 0.f; slots.jv(k, r) = pv[r]; if (kSubs == 4) mjv[r] = pv[r]; }
    }
    if (kSubs == 4 && plane_sub) {
#pragma unroll
      for (int r = 0; r < 4; r++) jv0[r] = sub_bcast<3>(pv[r]);
    }
    if (kSubs == 1 && lds_slots) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const bool on = k < nslots;           // wave-uniform
        // read first, select afterwards (see the hex layout below: otherwise every read sits in a branch of its own)
        float a0 = slots.jar(k, 0), a1 = slots.jar(k, 1), a2 = slots.jar(k, 2), a3 = slots.jar(k, 3), v0 = slots.jv(k, 0), v1 = slots.jv(k, 1), v2 = slots.jv(k, 2), v3 = slots.jv(k, 3), dk = slots.at(k, 2);
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(dk));
        qd_ja[2 * k] = f2{on ? a0 : 0.f, on ? a1 : 0.f}; qd_ja[2 * k + 1] = f2{on ? a2 : 0.f, on ? a3 : 0.f};
        qd_jv[2 * k] = f2{on ? v0 : 0.f, on ? v1 : 0.f}; qd_jv[2 * k + 1] = f2{on ? v2 : 0.f, on ? v3 : 0.f};
        qd_D[k] = on ? dk : 0.f;
      }
    }
    if (kSubs == 2) {
      const bool up = (threadIdx.x & 1) != 0;          // sub-lane 1: rows (2, 3)
      oc_lim_ja = f2{up ? jar_lim[2] : jar_lim[0], up ? 0.f : jar_lim[1]};
      oc_lim_jv = f2{up ? jv_lim[2] : jv_lim[0], up ? 0.f : jv_lim[1]};
      oc_lim_D = f2{up ? s.lim_D[2] : s.lim_D[0], up ? 0.f : s.lim_D[1]};
      oc_pl_ja = f2{up ? jar0[2] : jar0[0], up ? jar0[3] : jar0[1]};
      oc_pl_jv = f2{up ? jv0[2] : jv0[0], up ? jv0[3] : jv0[1]};
      if (lds_slots) {
        const int r0 = up ? 2 : 0;
#pragma unroll
        for (int k = 0; k < kMaxB; k++) {
          const bool on = k < units();          // wave-uniform
          float a0 = slots.jar(k, r0), a1 = slots.jar(k, r0 + 1), v0 = slots.jv(k, r0), v1 = slots.jv(k, r0 + 1), dk = slots.at(k, 2);      // read first, select afterwards
          asm volatile("" : "+v"(a0), "+v"(a1), "+v"(v0), "+v"(v1), "+v"(dk));
          qd_ja[k] = f2{on ? a0 : 0.f, on ? a1 : 0.f};
          qd_jv[k] = f2{on ? v0 : 0.f, on ? v1 : 0.f};
          oc_D[k] = on ? dk : 0.f;
        }
      }
    }
    if (kSubs == 4) {
      // pick by bit tests (selects, no branches): r = 0..3
      const int r = threadIdx.x & 3;
      const bool b0 = (r & 1) != 0, b1 = (r & 2) != 0;
      auto pick = [&](float x0, float x1, float x2, float x3) { const float lo = b0 ? x1 : x0, hi = b0 ? x3 : x2; return b1 ? hi : lo; };
      hx_ja = f2{pick(jar_lim[0], jar_lim[1], jar_lim[2], 0.f), pick(jar0[0], jar0[1], jar0[2], jar0[3])};
      hx_jv = f2{pick(jv_lim[0], jv_lim[1], jv_lim[2], 0.f), pick(jv0[0], jv0[1], jv0[2], jv0[3])};
      hx_D = f2{pick(s.lim_D[0], s.lim_D[1], s.lim_D[2], 0.f), s.con0.D};
      // rows r of the box slots (written by the sub-lanes that own the slots, just above and in the last epilogue)
#pragma unroll
      for (int p = 0; p < 2; p++) {
        const int k = 2 * p, k1 = 2 * p + 1;
        const bool on0 = k < nslots, on1 = k1 < nslots;           // wave-uniform
        // read first, select afterwards: with `on ? slots.x : 0.f` the compiler wraps every one of the twelve LDS reads in a branch of its own
        // (scalar mask test + s_cbranch, some with the mask negated through the vector unit): ~200 cycles of a one-wave SIMD per search
        float a0 = slots.jar(k, r), a1 = slots.jar(k1, r), v0 = slots.jv(k, r), v1 = slots.jv(k1, r), d0 = slots.at(k, 2), d1 = slots.at(k1, 2);
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(v0), "+v"(v1), "+v"(d0), "+v"(d1));
        ls_ja[p] = f2{on0 ? a0 : 0.f, on1 ? a1 : 0.f};
        ls_jv[p] = f2{on0 ? v0 : 0.f, on1 ? v1 : 0.f};
        ls_D[p] = f2{on0 ? d0 : 0.f, on1 ? d1 : 0.f};
      }
    }
    float ab = 0.f, bb_ = 0.f, eb = 0.f, al = 0.f, bl = 0.f, el = 0.f;
#pragma unroll
    for (int i = 0; i < 6; i++) { ab += sb[i] * Mab[i]; bb_ += sb[i] * s.qfs_b[i]; eb += sb[i] * mvb[i]; }
#pragma unroll
    for (int k = 0; k < 3; k++) { al += sl[k] * Mal[k]; bl += sl[k] * s.qfs_l[k]; el += sl[k] * mvl[k]; }
    const float qg0 = gauss, qg1 = (ab + quad_sum(al)) - (bb_ + quad_sum(bl)), qg2 = 0.5f * (eb + quad_sum(el));
    // Bracket update (mjx solver._update_bracket): a candidate y replaces the bracket end x when
    //   in_bracket(x, y) = (x.d0 < y.d0 < 0) or (x.d0 > y.d0 > 0),
    // tried in a fixed order, each test against the already updated end.  With u = d0 * sign(x.d0) (exact) this is a
    // running "0 < u_y < u_x": a chain over ONE scalar; the fields of the end follow every candidate that enters.
    // `frozen`: the env has left the search (its lanes only keep the wave company): threshold 0, nothing enters, the end stays as it is -
    // one select instead of an exec-mask region (save / branch / restore / branch, ~45 cycles of a one-wave SIMD) round the update
    // Every test is ONE compare whose outcome feeds selects directly: a candidate that is not on the bracket's side (u_c <= 0 or NaN) is
    // offered as +inf, so "0 < u_c < u" is the single `w < u`, and "something entered" is `u < u0` (an entering candidate lowers u strictly).
    // The former `(u_c > 0) & (u_c < u)` met in SGPR pairs (v_cmp -> s_and_b64 -> v_cndmask): on a SIMD with one wave each such chain
    // waits for the vector pipe to hand the mask to the scalar unit and back (tools/probes/issue_probe.hip: 33 cycles for four instructions).
    auto tighten = [](const LSPoint& x, const LSPoint& c1, const LSPoint& c2, const LSPoint& c3, bool frozen, bool& moved) {
      const float sg = x.d0 > 0.f ? 1.0f : -1.0f;
      const float u0 = frozen ? 0.f : x.d0 * sg;           // |x.d0| (0 if x.d0 == 0 or the env is frozen: nothing can enter the bracket)
      float u = u0;
      LSPoint r; r.alpha = x.alpha; r.cost = 0.f; r.d0 = x.d0; r.d1 = x.d1;
      auto offer = [&](const LSPoint& c) {
        const float uc = c.d0 * sg;
        const float w = uc > 0.f ? uc : INFINITY;
        const bool k = w < u;
        u = k ? w : u; r.alpha = k ? c.alpha : r.alpha; r.d0 = k ? c.d0 : r.d0; r.d1 = k ? c.d1 : r.d1;
      };
      offer(c1); offer(c2); offer(c3);
      moved = u < u0;
      return r;
    };
    PG_LTICK(s, 20);      // set-up: M s, J s, row hand-over, Gauss coefficients
    LSPoint p0, lo0;
    { const float a0 = 0.f; ls_points<1>(&a0, jv_lim, jv0, qg0, qg1, qg2, &p0); }
    { const float a1 = p0.alpha - div_normal(p0.d0, p0.d1); ls_points<1, false>(&a1, jv_lim, jv0, qg0, qg1, qg2, &lo0); }
    bool lesser = lo0.d0 < p0.d0;
    LSPoint hi = lesser ? p0 : lo0, lo = lesser ? lo0 : p0;
    bool swap = true; int it = 0;
    PG_LTICK(s, 21);      // the two initial points
#ifdef PGTT_EFFORT
    int eff_need = frozen ? 0 : 1;        // 0: the env was not in this trip at all
#endif
    for (;;) {
      bool done = it >= m->ls_iterations || !swap || ((lo.d0 < 0.f) && (lo.d0 > -gtol)) || ((hi.d0 > 0.f) && (hi.d0 < gtol));
      if (__ballot(!done) == 0ull) break;
#ifdef PGTT_EFFORT
      eff_need += (done || frozen) ? 0 : 1;
#endif
#ifdef PGTT_TIME
      s.cyc[18] += 1.f;           // line-search rounds executed by this wave
      s.cyc[19] += done ? 0.f : 1.f;   // ... of which this env needed
#endif
      const float al3[3] = {lo.alpha - div_normal(lo.d0, lo.d1), hi.alpha - div_normal(hi.d0, hi.d1), 0.5f * (lo.alpha + hi.alpha)};
      LSPoint pt[3];
      ls_points<3, false>(al3, jv_lim, jv0, qg0, qg1, qg2, pt);
      bool ml, mh;
      lo = tighten(lo, pt[0], pt[2], pt[1], done, ml);
      hi = tighten(hi, pt[1], pt[2], pt[0], done, mh);
      // a finished env moves nothing, so swap turns false and keeps it finished (`done` is sticky through !swap); its count may run on
      swap = ml | mh; it++;
    }
    PG_LTICK(s, 22);      // bracketing rounds
#ifdef PGTT_EFFORT
    s.eff |= (unsigned long long)eff_need << s.eff_pos; s.eff_pos += 3;
#endif
    {   // costs of the two points the bracket ended with
      const float al2[2] = {lo.alpha, hi.alpha};
      LSPoint fin[2];
      ls_points<2, true>(al2, jv_lim, jv0, qg0, qg1, qg2, fin);
      lo.cost = fin[0].cost; hi.cost = fin[1].cost;
    }
    PG_LTICK(s, 23);      // final costs
    bool improved = (lo.cost < p0.cost) || (hi.cost < p0.cost);
    float alpha = lo.cost < hi.cost ? lo.alpha : hi.alpha;
    float ia = (improved && !frozen) ? alpha : 0.f;
#ifdef PGTT_TRACE
    last_alpha = ia;
#endif
#pragma unroll
    for (int i = 0; i < 6; i++) { qb[i] += sb[i] * ia; Mab[i] += mvb[i] * ia; }
#pragma unroll
    for (int k = 0; k < 3; k++) { ql[k] += sl[k] * ia; Mal[k] += mvl[k] * ia; jar_lim[k] += jv_lim[k] * ia; }
#pragma unroll
    for (int r = 0; r < 4; r++) jar0[r] += jv0[r] * ia;
    for (int k0 = 0; k0 < own_trips(); k0++) {
      const int k = own_slot(k0);                                   // hex: every sub-lane works on ITS slot, in one pass; oct: on its one or two
      if (!own_on(k)) continue;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (kSubs != 4) slots.jar(k, r) += slots.jv(k, r) * ia;
        else { mjar[r] += mjv[r] * ia; slots.jar(k, r) = mjar[r]; }
      }
    }
    PG_LTICK(s, 24);      // update of qacc, M qacc, J qacc - aref
  }

  PG_INL void solve() {
#ifdef PGTT_EFFORT
    s.eff_pos = 15 * s.eff_sub; s.eff_sub++; s.eff_pat = 0xffffffffu;
#endif
    int nb = 0;
#pragma unroll
    for (int k = 0; k < kMaxB; k++) if (__ballot(s.nbox > k) != 0ull) nb = k + 1;
    nslots = nb;
#ifdef PGTT_TIME
    s.cyc[17] += (float)nb;
#endif
    any_lim = __ballot(s.lim_active[0] || s.lim_active[1] || s.lim_active[2]) != 0ull;
    any_con0 = __ballot(s.con0.row_active) != 0ull;
    plane_sub = kSubs == 4 && lds_slots && nb <= 3 && any_con0;
    if (kSubs == 2) {
      // oct layout: with two or three box slots in use somewhere in the wave the plane contact becomes one more unit, in the
      // first free slot (its record is rewritten every solve; collide() clears the slot's rows before the next one)
      oc_split = lds_slots && nb >= 2;
      plane_sub = oc_split && nb <= 3 && any_con0;
      plane_slot = nb;
      if (plane_sub) slots.store(nb, s.con0);
    }
    if (kSubs == 4 && lds_slots) mine = slots.load((int)(threadIdx.x & 3));
    PG_TICK(s, 3);
    // start from the cheaper of (unconstrained acceleration, warm start).  The warm start is evaluated LAST: when it wins
    // in every lane of the wave (the steady state) the solver state is already the one to continue from; only a wave
    // in which some env prefers the unconstrained acceleration pays a third evaluation of the per-env choice.
    init(s.qas_b, s.qas_l); update_constraint();
    const float cs = cost;
    init(s.wb, s.wl); update_constraint();
    const bool usew = cost < cs;
    if (__ballot(!usew) != 0ull) {
      float kb[6], kl[3];
#pragma unroll
      for (int i = 0; i < 6; i++) kb[i] = usew ? s.wb[i] : s.qas_b[i];
#pragma unroll
      for (int k = 0; k < 3; k++) kl[k] = usew ? s.wl[k] : s.qas_l[k];
      init(kb, kl); update_constraint();
    }
    PG_TICK(s, 4);
    update_gradient();
    PG_TICK(s, 5);
#ifdef PGTT_TRACE
    rec_state(100.f);
#endif
    const float scale = m->meaninertia * 18.0f;
    int niter = 0, trip = 0;
    for (;;) {
      float gnb = 0.f, gnl = 0.f;
#pragma unroll
      for (int i = 0; i < 6; i++) gnb += gb[i] * gb[i];
#pragma unroll
      for (int k = 0; k < 3; k++) gnl += gl[k] * gl[k];
      float gn = gnb + quad_sum(gnl);
      // `|`, not `||`: the three tests are per-env values, and short-circuit evaluation of a per-lane condition is an exec-mask region each
      const bool d_it = niter >= m->iterations, d_imp = div_normal(prev_cost - cost, scale) < m->tolerance, d_grad = div_normal(sqrtf(gn), scale) < m->tolerance;
      bool done = d_it | d_imp | d_grad;
      if (__ballot(!done) == 0ull) break;
      PG_TICK(s, 9);
      linesearch(done);
      PG_TICK(s, 6);
      // "done" is sticky and every other lane counts up, so the loop makes at most `iterations` trips: the constraint
      // forces, cost, gradient, Hessian and search direction of the last possible trip would never be used (mjx
      // computes them all the same); qacc is final after its line search
      if (++trip < m->iterations) {
        update_constraint();
        PG_TICK(s, 7);
        update_gradient();
      }
      PG_TICK(s, 8);
      s.cyc_iter();
      if (!done) niter++;
#ifdef PGTT_TRACE
      rec_state(200.f + niter);
#endif
    }
#pragma unroll
    for (int i = 0; i < 6; i++) { s.qacc_b[i] = qb[i]; s.wb[i] = qb[i]; }
#pragma unroll
    for (int k = 0; k < 3; k++) { s.qacc_l[k] = ql[k]; s.wl[k] = ql[k]; }
    s.niter = niter; s.niter_max = niter > s.niter_max ? niter : s.niter_max;
  }
};

}  // namespace pgtt
