// pgtt_physics.hip.h — shared device helpers of the physics kernel (gfx950, wave64): small vector / quaternion /
// spatial algebra, the constraint impedance law, the sphere-box narrow phase and the resident terrain table entry.
// The simulator itself is in pgtt_physics_quad.hip.h (one environment per 4 or 16 lanes).
//
// SoA state in HBM (buffer[row][env] => lane-coalesced loads/stores), model constants read through a
// wave-uniform pointer (scalar loads).  This is NOT the dense MJX formulation the oracle restates: it exploits
// the Go2 tree (free base + 4 independent 3-hinge legs) everywhere:
//   * joint-space inertia M and the Newton Hessian H = M + J^T D J are stored as symmetric ARROWHEAD
//     matrices (6x6 base block, four 3x6 couplings, four 3x3 leg blocks = 117 floats instead of 324) and
//     factorised leaf-first (leg Cholesky -> Schur complement on the base -> base Cholesky);
//   * a contact Jacobian is a 3x9 block (base 6 + own leg 3), pyramid rows are formed on the fly;
//   * joint-limit rows are +-e_dof and only touch the diagonal;
//   * the sphere-box broad phase (MJX top-k over 400 centre distances, then top-4 by penetration) is
//     evaluated as "penetrating pairs + exact rank counting", which yields the same ACTIVE contact set.
// Reference call sites this replaces: go2/joystick_pgtt.py:146-148 (mjx_env.step x n_substeps),
// go2/base.py:153-171 (compute_contact), go2/base.py:116-149 (sensor getters).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/pgtt.h"

namespace pgtt {

#define PG_INL __device__ __forceinline__
constexpr float kMinVal = 1e-15f;
constexpr float kMinImp = 0.0001f;
constexpr float kMaxImp = 0.9999f;

struct TerrainBox {                 // resident terrain table entry (80 B), built once by pgtt_set_terrain
  float px, py, pz, rb;             // centre, bounding radius |half-size|
  float sx, sy, sz, m00;            // half-size, rotation matrix row-major
  float m01, m02, m10, m11;
  float m12, m20, m21, m22;
  float hx, hy, hz, pad;            // half-extents of the WORLD-axis-aligned bounding box (|R| size, rounded up)
};

// Row `row` of an SoA block [rows][N] for env e (PG_ROW), element i of env e's record in an env-major array (PG_REC).  PG_ADDR32 (quad and oct
// layouts): through a 32-bit BYTE offset from a wave-uniform base - the form global_load / global_store take as "SGPR base + zero-extended VGPR
// offset": one register per address instead of a 64-bit pair, and one multiply-add to form it again where it is used.  With 64-bit indices the
// compiler kept the ~50 row addresses of the prologue alive until the stores (scratch in the oct layout: 18 of its 39 spilled dwords were address
// pairs) rather than redo a 64-bit multiply.  Valid while 4 * rows * N < 2^32 (pgtt_create refuses more envs).  The hex layout keeps the 64-bit
// forms it was tuned with: there the change moved code across fp-contraction decisions (results at rounding distance from the build before it)
// without a gain (round 4, docs/HISTORY.md 5.7).
#if defined(PG_SUBS) && PG_SUBS == 4
#define PG_ADDR32 0
#else
#define PG_ADDR32 1
#endif
template <class T> PG_INL const T& pg_at(const T* base, unsigned idx) { return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)(idx * (unsigned)sizeof(T))); }
template <class T> PG_INL T& pg_at(T* base, unsigned idx) { return *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + (size_t)(idx * (unsigned)sizeof(T))); }
#if PG_ADDR32
#define PG_ROW(base, row, N, e) pgtt::pg_at(base, (unsigned)(row) * (unsigned)(N) + (unsigned)(e))
#define PG_REC(base, e, stride, i) pgtt::pg_at(base, (unsigned)(e) * (unsigned)(stride) + (unsigned)(i))
#else
#define PG_ROW(base, row, N, e) ((base)[(row) * (long)(N) + (e)])
#define PG_REC(base, e, stride, i) ((base)[(long)(e) * (stride) + (i)])
#endif

// ------------------------------------------------------------------ small vector helpers
struct V3 { float x, y, z; };
PG_INL V3 v3(float x, float y, float z) { return V3{x, y, z}; }
PG_INL V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PG_INL V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PG_INL V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
PG_INL V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
PG_INL float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PG_INL V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
PG_INL float norm(V3 a) { return sqrtf(dot(a, a)); }
struct Q4 { float w, x, y, z; };
PG_INL Q4 qmul(Q4 u, Q4 v) {
  return Q4{u.w * v.w - u.x * v.x - u.y * v.y - u.z * v.z, u.w * v.x + u.x * v.w + u.y * v.z - u.z * v.y,
            u.w * v.y - u.x * v.z + u.y * v.w + u.z * v.x, u.w * v.z + u.x * v.y - u.y * v.x + u.z * v.w};
}
struct M3 { float m[9]; };
PG_INL M3 qmat(Q4 q) {
  float q00 = q.w * q.w, q01 = q.w * q.x, q02 = q.w * q.y, q03 = q.w * q.z;
  float q11 = q.x * q.x, q12 = q.x * q.y, q13 = q.x * q.z, q22 = q.y * q.y, q23 = q.y * q.z, q33 = q.z * q.z;
  M3 r;
  r.m[0] = q00 + q11 - q22 - q33; r.m[1] = 2 * (q12 - q03); r.m[2] = 2 * (q13 + q02);
  r.m[3] = 2 * (q12 + q03); r.m[4] = q00 - q11 + q22 - q33; r.m[5] = 2 * (q23 - q01);
  r.m[6] = 2 * (q13 - q02); r.m[7] = 2 * (q23 + q01); r.m[8] = q00 - q11 - q22 + q33;
  return r;
}
PG_INL V3 qrot(V3 v, Q4 q) {        // mjx math.rotate
  V3 u = v3(q.x, q.y, q.z);
  float uv = dot(u, v), uu = dot(u, u);
  V3 c = cross(u, v);
  return 2.0f * (uv * u) + (q.w * q.w - uu) * v + (2.0f * q.w) * c;
}
PG_INL V3 mcol(const M3& a, int i) { return v3(a.m[i], a.m[3 + i], a.m[6 + i]); }
PG_INL V3 mtmul(const M3& a, V3 v) {   // a^T v
  return v3(a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z, a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z,
            a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z);
}
PG_INL V3 mmul(const M3& a, V3 v) {
  return v3(a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z,
            a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z);
}
// normalise with MJX's zero guard (allclose(x, 0, atol=1e-8) -> treated as zero, norm 0)
PG_INL float normalize3(V3& a) {
  bool z = fabsf(a.x) <= 1e-8f && fabsf(a.y) <= 1e-8f && fabsf(a.z) <= 1e-8f;
  if (z) a = v3(1.f, 1.f, 1.f);
  float n = norm(a);
  float d = n + (z ? 1.0f : 0.0f);
  a = v3(a.x / d, a.y / d, a.z / d);
  return z ? 0.0f : n;
}
PG_INL void normalize4(Q4& q) {
  bool z = fabsf(q.w) <= 1e-8f && fabsf(q.x) <= 1e-8f && fabsf(q.y) <= 1e-8f && fabsf(q.z) <= 1e-8f;
  if (z) q = Q4{1.f, 1.f, 1.f, 1.f};
  float n = sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z) + (z ? 1.0f : 0.0f);
  q = Q4{q.w / n, q.x / n, q.y / n, q.z / n};
}
PG_INL void make_frame(V3 a, V3& n, V3& t1, V3& t2) {
  normalize3(a);
  V3 y = (a.y > -0.5f && a.y < 0.5f) ? v3(0, 1, 0) : v3(0, 0, 1);
  V3 b = y - a * dot(a, y);
  normalize3(b);
  n = a; t1 = b; t2 = cross(a, b);
}

// spatial 6-vectors [ang, lin] and 10-float spatial inertias [Ixx,Iyy,Izz,Ixy,Ixz,Iyz, m*off(3), m]
struct S6 { V3 a, l; };
PG_INL S6 operator+(S6 p, S6 q) { return S6{p.a + q.a, p.l + q.l}; }
PG_INL S6 operator*(S6 p, float s) { return S6{p.a * s, p.l * s}; }
PG_INL float dot6(S6 p, S6 q) { return dot(p.a, q.a) + dot(p.l, q.l); }
struct I10 { float i[10]; };
PG_INL S6 inert_mul(const I10& I, S6 v) {
  V3 h = v3(I.i[6], I.i[7], I.i[8]);
  V3 ang = v3(I.i[0] * v.a.x + I.i[3] * v.a.y + I.i[4] * v.a.z, I.i[3] * v.a.x + I.i[1] * v.a.y + I.i[5] * v.a.z,
              I.i[4] * v.a.x + I.i[5] * v.a.y + I.i[2] * v.a.z) + cross(h, v.l);
  V3 lin = I.i[9] * v.l - cross(h, v.a);
  return S6{ang, lin};
}
PG_INL S6 motion_cross(S6 u, S6 v) { return S6{cross(u.a, v.a), cross(u.l, v.a) + cross(u.a, v.l)}; }
PG_INL S6 motion_cross_force(S6 v, S6 f) { return S6{cross(v.a, f.a) + cross(v.l, f.l), cross(v.a, f.l)}; }

// ------------------------------------------------------------------ symmetric arrowhead matrix
constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }   // j <= i
PG_INL float sel4(int l, float a, float b, float c, float d) { return l == 0 ? a : (l == 1 ? b : (l == 2 ? c : d)); }
// impedance / stiffness / damping of a constraint row (mjx constraint._kbi); returns (k*imp, b, imp)
PG_INL void kbi(float timestep, const float* solref, const float* solimp, float pos, float& k_imp, float& b, float& imp) {
  float timeconst = fmaxf(solref[0], 2.0f * timestep), dampratio = solref[1];
  float dmin = fminf(fmaxf(solimp[0], kMinImp), kMaxImp), dmax = fminf(fmaxf(solimp[1], kMinImp), kMaxImp);
  float width = fmaxf(solimp[2], kMinVal), mid = fminf(fmaxf(solimp[3], kMinImp), kMaxImp), power = fmaxf(solimp[4], 1.0f);
  float k = 1.0f / (dmax * dmax * timeconst * timeconst * dampratio * dampratio);
  b = 2.0f / (dmax * timeconst);
  if (solref[0] <= 0.f) k = -solref[0] / (dmax * dmax);
  if (solref[1] <= 0.f) b = -solref[1] / dmax;
  float x = fabsf(pos) / width;
  float ya, yb;
  if (power == 2.0f) { ya = (1.0f / mid) * (x * x); yb = 1.0f - (1.0f / (1.0f - mid)) * ((1.0f - x) * (1.0f - x)); }
  else {
    // general solimp power (not the reference's: go2_mjx_feetonly.xml keeps MuJoCo's default 2): x^p = exp2(p log2 x) on the hardware
    // transcendentals (1 ulp each; x in [0, 1], log2(0) = -inf -> 0).  The library powf() costs ~800 inlined instructions per kbi() call site -
    // 3160 of the 17 k instructions (25 KB) of a kernel that never executes them
    auto pw = [](float b, float e) { return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(b)); };
    ya = (1.0f / pw(mid, power - 1.0f)) * pw(x, power); yb = 1.0f - (1.0f / pw(1.0f - mid, power - 1.0f)) * pw(1.0f - x, power);
  }
  float y = x < mid ? ya : yb;
  imp = dmin + y * (dmax - dmin);
  imp = fminf(fmaxf(imp, dmin), dmax);
  if (x > 1.0f) imp = dmax;
  k_imp = k * imp;
}

// sphere (centre c in world) vs box: mjx collision_convex._sphere_convex specialised to a box's 6 quads.
// faces in MJX order: 0:-y 1:-z 2:+x 3:+y 4:+z 5:-x
PG_INL void sphere_box(V3 c_world, float radius, const TerrainBox& tb, float& dist, V3& pos_w, V3& n_w) {
  M3 R; R.m[0] = tb.m00; R.m[1] = tb.m01; R.m[2] = tb.m02; R.m[3] = tb.m10; R.m[4] = tb.m11; R.m[5] = tb.m12;
  R.m[6] = tb.m20; R.m[7] = tb.m21; R.m[8] = tb.m22;
  V3 c = mtmul(R, c_world - v3(tb.px, tb.py, tb.pz));
  auto pick3 = [](int i, float x, float y, float z) { return i == 0 ? x : (i == 1 ? y : z); };
  const int fax[6] = {1, 2, 0, 1, 2, 0};
  const float fsg[6] = {-1.f, -1.f, 1.f, 1.f, 1.f, -1.f};
  const float ccs[3] = {c.x, c.y, c.z}, szs[3] = {tb.sx, tb.sy, tb.sz};
  // support_f = dot((c - r n) - v0, n) = sg*c[ax] - r - size[ax]
  int best = 0; float bs = -3.0e38f;
#pragma unroll
  for (int f = 0; f < 6; f++) {
    float s = fsg[f] * (ccs[fax[f]] - fsg[f] * radius - fsg[f] * szs[fax[f]]);
    if (s >= 0.f) s = -1e12f;
    if (s > bs) { bs = s; best = f; }
  }
  // face axis, in-plane axes (u, v) and the vertex cycle of the chosen face (MJX vertex / face tables)
  const int ax = best == 0 || best == 3 ? 1 : (best == 1 || best == 4 ? 2 : 0);
  const float sg = (best == 2 || best == 3 || best == 4) ? 1.f : -1.f;
  const int au = (best == 2 || best == 5) ? 1 : 0;
  const int av = (best == 1 || best == 4) ? 1 : 2;
  // vertex sign patterns: A = (-,-),(+,-),(+,+),(-,+) [faces 0,4]; B = (-,-),(-,+),(+,+),(+,-) [faces 1,3,5];
  // C = (+,-),(+,+),(-,+),(-,-) [face 2]
  const bool patA = best == 0 || best == 4, patC = best == 2;
  const float s0u = patC ? 1.f : -1.f, s0v = -1.f;
  const float s1u = patA ? 1.f : (patC ? 1.f : -1.f), s1v = patA ? -1.f : 1.f;
  const float s2u = patC ? -1.f : 1.f, s2v = 1.f;
  const float s3u = patA ? -1.f : (patC ? -1.f : 1.f), s3v = patA ? 1.f : -1.f;
  const float su = pick3(au, tb.sx, tb.sy, tb.sz), sv = pick3(av, tb.sx, tb.sy, tb.sz), sw = pick3(ax, tb.sx, tb.sy, tb.sz);
  const float cu = pick3(au, c.x, c.y, c.z), cvv = pick3(av, c.x, c.y, c.z), cw = pick3(ax, c.x, c.y, c.z);
  float fu[4] = {s0u * su, s1u * su, s2u * su, s3u * su}, fv[4] = {s0v * sv, s1v * sv, s2v * sv, s3v * sv};
  // project the centre onto the face plane: pt = c - ((c - v0).n) n   (n = sg * e_ax)
  float dd = (cw - sg * sw) * sg;
  float pw = cw - dd * sg;
  float pu = cu, pv = cvv;
  // edge k runs p0 = face[k-1] -> p1 = face[k]; edge normal = cross(p1 - p0, n) (NOT normalised, as in MJX)
  float hand = ((au + 1) % 3 == av) ? 1.0f : -1.0f;    // (au, av, ax) right-handed?
  float ed[4]; bool inside = true;
  float enu[4], env[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int k0 = (k + 3) & 3;
    float eu = fu[k] - fu[k0], ev = fv[k] - fv[k0];
    enu[k] = hand * ev * sg; env[k] = -hand * eu * sg;
    ed[k] = (pu - fu[k0]) * enu[k] + (pv - fv[k0]) * env[k];
    if (!(ed[k] <= 0.f)) inside = false;
  }
  int idx = 0; float bd = 0.f;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    bool degenerate = (enu[k] == 0.f && env[k] == 0.f);
    float e = (degenerate || ed[k] < 0.f) ? 1e12f : ed[k];
    if (k == 0 || e < bd) { bd = e; idx = k; }
  }
  float au0 = 0, av0 = 0, bu0 = 0, bv0 = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) { const bool hit = k == idx; const int k0 = (k + 3) & 3;
    au0 = hit ? fu[k0] : au0; av0 = hit ? fv[k0] : av0; bu0 = hit ? fu[k] : bu0; bv0 = hit ? fv[k] : bv0; }
  float abu = bu0 - au0, abv = bv0 - av0;
  float t = ((pu - au0) * abu + (pv - av0) * abv) / (abu * abu + abv * abv + 1e-6f);
  t = fminf(fmaxf(t, 0.f), 1.f);
  if (!inside) { pu = au0 + t * abu; pv = av0 + t * abv; }
  // back from (u, v, w) to (x, y, z)
  float pt[3];
#pragma unroll
  for (int i = 0; i < 3; i++) pt[i] = i == au ? pu : (i == av ? pv : pw);
  V3 ptv = v3(pt[0], pt[1], pt[2]);
  V3 n = ptv - c;
  float dn = normalize3(n);
#ifndef PGTT_SPHERE_CONVEX_FLIP
  // A sphere whose CENTRE is inside the box (penetration deeper than its radius) keeps the inward normal of the least-penetrated
  // face and a depth that goes on growing.  The literal `n = normalize(pt - c)` of _sphere_convex as recalled would flip the frame there and drop
  // the contact one radius further down - feet then fall through box tops at every hard landing (the 17.5 mm foot on this soft
  // contact is pushed deeper than its radius at ~2 m/s).  The reference's own statistics rule that behaviour out: with the
  // flip, policy177 tumbles on every stair level and its contact duty / tilt spread are 25 % / 130 % off the values its
  // normaliser recorded over 443 M samples of MJX; without it they agree to 1 % / 8 % (tests/test_gpu_policy.py).
  // -DPGTT_SPHERE_CONVEX_FLIP builds the recalled variant (oracle/Makefile has the same switch).
  if ((fabsf(c.x) <= tb.sx) & (fabsf(c.y) <= tb.sy) & (fabsf(c.z) <= tb.sz)) { n = n * -1.0f; dn = -dn; }      // centre INSIDE the box
#endif
  V3 spt = c + n * radius;
  dist = dn - radius;
  V3 pl = (ptv + spt) * 0.5f;
  n_w = mmul(R, n);
  pos_w = mmul(R, pl) + v3(tb.px, tb.py, tb.pz);
}

}  // namespace pgtt
