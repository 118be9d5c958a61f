// pgtt_physics.hip.h — device code of the physics kernel (gfx950, wave64).
//
// One environment per lane, SoA state in HBM (buffer[row][env] => lane-coalesced loads/stores), model
// constants read through a wave-uniform pointer (scalar loads).  This is NOT the dense MJX formulation the
// oracle restates: it exploits the Go2 tree (free base + 4 independent 3-hinge legs) everywhere:
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
constexpr int kMaxPen = 8;          // per-env cap on simultaneously penetrating (foot, box) pairs

struct TerrainBox {                 // resident terrain table entry (80 B), built once by pgtt_set_terrain
  float px, py, pz, rb;             // centre, bounding radius |half-size|
  float sx, sy, sz, m00;            // half-size, rotation matrix row-major
  float m01, m02, m10, m11;
  float m12, m20, m21, m22;
  float hx, hy, hz, pad;            // half-extents of the WORLD-axis-aligned bounding box (|R| size, rounded up)
};

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
struct Arrow {
  float bb[21];      // 6x6 base block, lower triangle
  float lb[4][18];   // leg x base coupling, [i*6 + k]
  float ll[4][6];    // 3x3 leg block, lower triangle
};
// y = A x   (x,y: 18-vectors, base first then legs FL,FR,RL,RR)
PG_INL void arrow_mul(const Arrow& A, const float* x, float* y) {
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 6; j++) s += A.bb[i >= j ? tri(i, j) : tri(j, i)] * x[j];
    y[i] = s;
  }
#pragma unroll
  for (int l = 0; l < 4; l++) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 6; k++) { s += A.lb[l][i * 6 + k] * x[k]; y[k] += A.lb[l][i * 6 + k] * x[6 + 3 * l + i]; }
#pragma unroll
      for (int j = 0; j < 3; j++) s += A.ll[l][i >= j ? tri(i, j) : tri(j, i)] * x[6 + 3 * l + j];
      y[6 + 3 * l + i] = s;
    }
  }
}
// in-place leaf-first Cholesky: ll <- L_l, lb <- W_l = L_l^-1 B_l, bb <- chol(A_bb - sum W_l^T W_l)
PG_INL void arrow_factor(Arrow& A) {
#pragma unroll
  for (int l = 0; l < 4; l++) {
    float* c = A.ll[l];
    float l00 = sqrtf(c[0]);
    float l10 = c[1] / l00, l20 = c[3] / l00;
    float l11 = sqrtf(c[2] - l10 * l10);
    float l21 = (c[4] - l20 * l10) / l11;
    float l22 = sqrtf(c[5] - l20 * l20 - l21 * l21);
    c[0] = l00; c[1] = l10; c[2] = l11; c[3] = l20; c[4] = l21; c[5] = l22;
    float* w = A.lb[l];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      float w0 = w[k] / l00;
      float w1 = (w[6 + k] - l10 * w0) / l11;
      float w2 = (w[12 + k] - l20 * w0 - l21 * w1) / l22;
      w[k] = w0; w[6 + k] = w1; w[12 + k] = w2;
    }
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j <= i; j++)
        A.bb[tri(i, j)] -= w[i] * w[j] + w[6 + i] * w[6 + j] + w[12 + i] * w[12 + j];
  }
#pragma unroll
  for (int j = 0; j < 6; j++) {
    float s = A.bb[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) s -= A.bb[tri(j, k)] * A.bb[tri(j, k)];
    float d = sqrtf(s);
    A.bb[tri(j, j)] = d;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      float t = A.bb[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) t -= A.bb[tri(i, k)] * A.bb[tri(j, k)];
      A.bb[tri(i, j)] = t / d;
    }
  }
}
// x = A^-1 b with the factor produced by arrow_factor
PG_INL void arrow_solve(const Arrow& F, const float* b, float* x) {
  float yl[12], rb[6];
#pragma unroll
  for (int k = 0; k < 6; k++) rb[k] = b[k];
#pragma unroll
  for (int l = 0; l < 4; l++) {
    const float* c = F.ll[l];
    float y0 = b[6 + 3 * l] / c[0];
    float y1 = (b[7 + 3 * l] - c[1] * y0) / c[2];
    float y2 = (b[8 + 3 * l] - c[3] * y0 - c[4] * y1) / c[5];
    yl[3 * l] = y0; yl[3 * l + 1] = y1; yl[3 * l + 2] = y2;
#pragma unroll
    for (int k = 0; k < 6; k++) rb[k] -= F.lb[l][k] * y0 + F.lb[l][6 + k] * y1 + F.lb[l][12 + k] * y2;
  }
  float z[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float s = rb[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= F.bb[tri(i, k)] * z[k];
    z[i] = s / F.bb[tri(i, i)];
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    float s = z[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) s -= F.bb[tri(k, i)] * x[k];
    x[i] = s / F.bb[tri(i, i)];
  }
#pragma unroll
  for (int l = 0; l < 4; l++) {
    const float* c = F.ll[l];
    float t0 = yl[3 * l], t1 = yl[3 * l + 1], t2 = yl[3 * l + 2];
#pragma unroll
    for (int k = 0; k < 6; k++) { t0 -= F.lb[l][k] * x[k]; t1 -= F.lb[l][6 + k] * x[k]; t2 -= F.lb[l][12 + k] * x[k]; }
    float x2 = t2 / c[5];
    float x1 = (t1 - c[4] * x2) / c[2];
    float x0 = (t0 - c[1] * x1 - c[3] * x2) / c[0];
    x[6 + 3 * l] = x0; x[7 + 3 * l] = x1; x[8 + 3 * l] = x2;
  }
}

// ------------------------------------------------------------------ per-env model view (nominal or domain-randomised)
struct EnvModel {
  float mass[13];
  V3 base_ipos;
  float qpos0j[12], armature[12], damping[12], gain[12], bias1[12];
  float floor_friction;
};
template <bool HAS_DR>
PG_INL void load_env_model(const PgttModel* __restrict__ m, const float* __restrict__ prm, int N, int e, EnvModel& em) {
  if (HAS_DR) {
#pragma unroll
    for (int b = 0; b < 13; b++) em.mass[b] = prm[(PGTT_P_BODY_MASS + b) * (long)N + e];
    em.base_ipos = v3(prm[(PGTT_P_BASE_IPOS + 0) * (long)N + e], prm[(PGTT_P_BASE_IPOS + 1) * (long)N + e],
                      prm[(PGTT_P_BASE_IPOS + 2) * (long)N + e]);
#pragma unroll
    for (int j = 0; j < 12; j++) {
      em.qpos0j[j] = prm[(PGTT_P_QPOS0 + j) * (long)N + e];
      em.armature[j] = prm[(PGTT_P_ARMATURE + j) * (long)N + e];
      em.damping[j] = prm[(PGTT_P_DAMPING + j) * (long)N + e];
      em.gain[j] = prm[(PGTT_P_GAIN + j) * (long)N + e];
      em.bias1[j] = prm[(PGTT_P_BIAS1 + j) * (long)N + e];
    }
    em.floor_friction = prm[PGTT_P_FLOOR_FRICTION * (long)N + e];
  } else {
#pragma unroll
    for (int b = 0; b < 13; b++) em.mass[b] = m->body_mass[b];
    em.base_ipos = v3(m->body_ipos[0][0], m->body_ipos[0][1], m->body_ipos[0][2]);
#pragma unroll
    for (int j = 0; j < 12; j++) {
      em.qpos0j[j] = m->qpos0[7 + j]; em.armature[j] = m->dof_armature[6 + j]; em.damping[j] = m->dof_damping[6 + j];
      em.gain[j] = m->act_gain[j]; em.bias1[j] = m->act_bias[j][1];
    }
    em.floor_friction = m->floor_friction[0];
  }
}

// ------------------------------------------------------------------ contact record
struct Contact {
  int leg;           // 0..3 or -1 (slot empty)
  int box;           // -1 plane, >=0 box index, -2 empty
  float dist;
  float mu;
  float D;           // common to the 4 pyramid rows
  float aref[4];
  float J[3][9];     // contact-frame Jacobian: rows normal, tangent1, tangent2; cols base(6) + own leg(3)
  bool row_active;   // dist < margin
};

struct PenPair { float dist, key; int idx; V3 pos, n; };

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
  else { ya = (1.0f / powf(mid, power - 1.0f)) * powf(x, power); yb = 1.0f - (1.0f / powf(1.0f - mid, power - 1.0f)) * powf(1.0f - x, power); }
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
  for (int k = 0; k < 4; k++) if (k == idx) { int k0 = (k + 3) & 3; au0 = fu[k0]; av0 = fv[k0]; bu0 = fu[k]; bv0 = fv[k]; }
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
  V3 spt = c + n * radius;
  dist = dn - radius;
  V3 pl = (ptv + spt) * 0.5f;
  n_w = mmul(R, n);
  pos_w = mmul(R, pl) + v3(tb.px, tb.py, tb.pz);
}

// ------------------------------------------------------------------ the per-env simulator
struct Sim {
  // state
  float qpos[19], qvel[18], warm[18], ctrl[12];
  // kinematics
  V3 p0, com; M3 R0;
  V3 anchor[12], axis[12];
  V3 footc[4], sitef[4], imu; // geom centres, foot sites, imu site (leg order)
  I10 cin[13];
  S6 cdr[3];                 // base rotational cdofs (translations are unit vectors)
  S6 cdl[12];                // leg cdofs
  Arrow M, LM;
  float meaninertia_scale;
  // velocity
  S6 cvel[13], cddr[3], cddl[12];
  float qfrc_smooth[18], qacc_smooth[18], act_force[12];
  // constraints
  bool lim_active[12]; float lim_sign[12], lim_D[12], lim_aref[12];
  Contact con[8];
  int ncon_box;
  // outputs
  float qacc[18];
  int niter, niter_max;
};

PG_INL float sel4(int l, float a, float b, float c, float d) { return l == 0 ? a : (l == 1 ? b : (l == 2 ? c : d)); }

template <bool HAS_DR>
struct Physics {
  const PgttModel* __restrict__ m;
  const EnvModel& em;
  Sim& s;
  PG_INL Physics(const PgttModel* m_, const EnvModel& em_, Sim& s_) : m(m_), em(em_), s(s_) {}

  // ---- kinematics, COM, spatial inertias, cdofs, CRB -> M, factor
  PG_INL void position_stage() {
    Q4 q0{s.qpos[3], s.qpos[4], s.qpos[5], s.qpos[6]};
    normalize4(q0);
    s.qpos[3] = q0.w; s.qpos[4] = q0.x; s.qpos[5] = q0.y; s.qpos[6] = q0.z;
    s.p0 = v3(s.qpos[0], s.qpos[1], s.qpos[2]);
    s.R0 = qmat(q0);
    V3 xipos[13]; float Iw[13][6];
    auto body_inertia = [&](int mb, V3 xp, Q4 xq, V3 ipos) {
      xipos[mb] = xp + qrot(ipos, xq);
      Q4 iq{m->body_iquat[mb][0], m->body_iquat[mb][1], m->body_iquat[mb][2], m->body_iquat[mb][3]};
      M3 xi = qmat(qmul(xq, iq));
      float d0 = m->body_inertia[mb][0], d1 = m->body_inertia[mb][1], d2 = m->body_inertia[mb][2];
      // R diag(d) R^T, order xx,yy,zz,xy,xz,yz
      Iw[mb][0] = xi.m[0] * d0 * xi.m[0] + xi.m[1] * d1 * xi.m[1] + xi.m[2] * d2 * xi.m[2];
      Iw[mb][1] = xi.m[3] * d0 * xi.m[3] + xi.m[4] * d1 * xi.m[4] + xi.m[5] * d2 * xi.m[5];
      Iw[mb][2] = xi.m[6] * d0 * xi.m[6] + xi.m[7] * d1 * xi.m[7] + xi.m[8] * d2 * xi.m[8];
      Iw[mb][3] = xi.m[0] * d0 * xi.m[3] + xi.m[1] * d1 * xi.m[4] + xi.m[2] * d2 * xi.m[5];
      Iw[mb][4] = xi.m[0] * d0 * xi.m[6] + xi.m[1] * d1 * xi.m[7] + xi.m[2] * d2 * xi.m[8];
      Iw[mb][5] = xi.m[3] * d0 * xi.m[6] + xi.m[4] * d1 * xi.m[7] + xi.m[5] * d2 * xi.m[8];
    };
    body_inertia(0, s.p0, q0, em.base_ipos);
    s.imu = s.p0 + qrot(v3(m->imu_pos[0], m->imu_pos[1], m->imu_pos[2]), q0);
#pragma unroll
    for (int l = 0; l < 4; l++) {
      V3 pp = s.p0; Q4 pq = q0;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        int j = 3 * l + k, mb = 1 + j;
        V3 pos = pp + qrot(v3(m->body_pos[mb][0], m->body_pos[mb][1], m->body_pos[mb][2]), pq);
        V3 ax = v3(m->jnt_axis[j][0], m->jnt_axis[j][1], m->jnt_axis[j][2]);
        float ang = s.qpos[7 + j] - em.qpos0j[j];
        float sn, cs; sincosf(0.5f * ang, &sn, &cs);
        Q4 ql{cs, ax.x * sn, ax.y * sn, ax.z * sn};
        s.axis[j] = qrot(ax, pq);          // body_quat is identity for every link (checked at create)
        s.anchor[j] = pos;
        Q4 xq = qmul(pq, ql);
        body_inertia(mb, pos, xq, v3(m->body_ipos[mb][0], m->body_ipos[mb][1], m->body_ipos[mb][2]));
        pp = pos; pq = xq;
      }
      s.footc[l] = pp + qrot(v3(m->foot_geom_pos[l][0], m->foot_geom_pos[l][1], m->foot_geom_pos[l][2]), pq);
      s.sitef[l] = pp + qrot(v3(m->foot_site_pos[l][0], m->foot_site_pos[l][1], m->foot_site_pos[l][2]), pq);
    }
    // subtree COM of the robot
    V3 acc = v3(0, 0, 0); float mt = 0.f;
#pragma unroll
    for (int b = 12; b >= 0; b--) { acc = acc + xipos[b] * em.mass[b]; mt += em.mass[b]; }
    s.com = acc * (1.0f / fmaxf(mt, kMinVal));
    // spatial inertias about the COM
#pragma unroll
    for (int b = 0; b < 13; b++) {
      V3 o = xipos[b] - s.com; float mb_ = em.mass[b];
      float oo = dot(o, o);
      I10& c = s.cin[b];
      c.i[0] = Iw[b][0] + mb_ * (oo - o.x * o.x); c.i[1] = Iw[b][1] + mb_ * (oo - o.y * o.y); c.i[2] = Iw[b][2] + mb_ * (oo - o.z * o.z);
      c.i[3] = Iw[b][3] - mb_ * o.x * o.y; c.i[4] = Iw[b][4] - mb_ * o.x * o.z; c.i[5] = Iw[b][5] - mb_ * o.y * o.z;
      c.i[6] = o.x * mb_; c.i[7] = o.y * mb_; c.i[8] = o.z * mb_; c.i[9] = mb_;
    }
    // cdofs
    V3 ob = s.com - s.p0;
#pragma unroll
    for (int k = 0; k < 3; k++) { V3 a = mcol(s.R0, k); s.cdr[k] = S6{a, cross(a, ob)}; }
#pragma unroll
    for (int j = 0; j < 12; j++) s.cdl[j] = S6{s.axis[j], cross(s.axis[j], s.com - s.anchor[j])};
    // composite inertias and the arrowhead M
    I10 crb_base = s.cin[0];
#pragma unroll
    for (int l = 0; l < 4; l++) {
      I10 crb = s.cin[3 + 3 * l];
#pragma unroll
      for (int k = 2; k >= 0; k--) {
        int j = 3 * l + k;
        if (k < 2) {
#pragma unroll
          for (int i = 0; i < 10; i++) crb.i[i] += s.cin[1 + j].i[i];
        }
        S6 f = inert_mul(crb, s.cdl[j]);
#pragma unroll
        for (int kk = 0; kk <= k; kk++) s.M.ll[l][tri(k, kk)] = dot6(s.cdl[3 * l + kk], f);
        s.M.lb[l][k * 6 + 0] = f.l.x; s.M.lb[l][k * 6 + 1] = f.l.y; s.M.lb[l][k * 6 + 2] = f.l.z;
#pragma unroll
        for (int r = 0; r < 3; r++) s.M.lb[l][k * 6 + 3 + r] = dot6(s.cdr[r], f);
        s.M.ll[l][tri(k, k)] += em.armature[j];
      }
#pragma unroll
      for (int i = 0; i < 10; i++) crb_base.i[i] += crb.i[i];
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
      S6 cd = k < 3 ? S6{v3(0, 0, 0), v3(k == 0, k == 1, k == 2)} : s.cdr[k - 3];
      S6 f = inert_mul(crb_base, cd);
      float fl[3] = {f.l.x, f.l.y, f.l.z};
#pragma unroll
      for (int kk = 0; kk <= k; kk++) s.M.bb[tri(k, kk)] = kk < 3 ? fl[kk] : dot6(s.cdr[kk - 3], f);
    }
    s.LM = s.M;
    arrow_factor(s.LM);
  }

  // ---- com_vel, passive, rne bias, actuation, qacc_smooth
  PG_INL void velocity_stage() {
    const float* qv = s.qvel;
    S6 cv0{v3(0, 0, 0), v3(qv[0], qv[1], qv[2])};        // after the 3 translational dofs
#pragma unroll
    for (int k = 0; k < 3; k++) s.cddr[k] = motion_cross(cv0, s.cdr[k]);
    S6 cvb = cv0;
#pragma unroll
    for (int k = 0; k < 3; k++) cvb = cvb + s.cdr[k] * qv[3 + k];
    s.cvel[0] = cvb;
    S6 cacc0{v3(0, 0, 0), v3(-m->gravity[0], -m->gravity[1], -m->gravity[2])};
    S6 caccb = cacc0;
#pragma unroll
    for (int k = 0; k < 3; k++) caccb = caccb + s.cddr[k] * qv[3 + k];
    auto body_force = [&](int mb, S6 cacc) {
      S6 f1 = inert_mul(s.cin[mb], cacc);
      S6 f2 = inert_mul(s.cin[mb], s.cvel[mb]);
      return f1 + motion_cross_force(s.cvel[mb], f2);
    };
    S6 fbase = body_force(0, caccb);
    float bias[18];
#pragma unroll
    for (int l = 0; l < 4; l++) {
      S6 cv = cvb, ca = caccb; S6 fb[3];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        int j = 3 * l + k;
        s.cddl[j] = motion_cross(cv, s.cdl[j]);
        cv = cv + s.cdl[j] * qv[6 + j];
        s.cvel[1 + j] = cv;
        ca = ca + s.cddl[j] * qv[6 + j];
        fb[k] = body_force(1 + j, ca);
      }
      fb[1] = fb[1] + fb[2]; fb[0] = fb[0] + fb[1];
#pragma unroll
      for (int k = 0; k < 3; k++) bias[6 + 3 * l + k] = dot6(s.cdl[3 * l + k], fb[k]);
      fbase = fbase + fb[0];
    }
    bias[0] = fbase.l.x; bias[1] = fbase.l.y; bias[2] = fbase.l.z;
#pragma unroll
    for (int k = 0; k < 3; k++) bias[3 + k] = dot6(s.cdr[k], fbase);
    // actuation + passive
#pragma unroll
    for (int i = 0; i < 6; i++) s.qfrc_smooth[i] = -m->dof_damping[i] * qv[i] - bias[i];
#pragma unroll
    for (int l = 0; l < 4; l++) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        int j = 3 * l + k, a = 3 * (l ^ 1) + k;          // actuators are declared FR,FL,RR,RL
        float c = fminf(fmaxf(s.ctrl[a], m->act_ctrlrange[a][0]), m->act_ctrlrange[a][1]);
        float force = em.gain[a] * c + (m->act_bias[a][0] + em.bias1[a] * s.qpos[7 + j] + m->act_bias[a][2] * qv[6 + j]);
        force = fminf(fmaxf(force, m->act_forcerange[a][0]), m->act_forcerange[a][1]);
        s.act_force[a] = force;
        s.qfrc_smooth[6 + j] = -em.damping[j] * qv[6 + j] - bias[6 + j] + force;
      }
    }
    arrow_solve(s.LM, s.qfrc_smooth, s.qacc_smooth);
  }

  // ---- contact Jacobian block (3x9) of a world point on leg l, expressed in the frame (n, t1, t2), times sign
  PG_INL void contact_jac(Contact& c, int l, V3 pos, V3 n, V3 t1, V3 t2, float sign) {
    V3 off = pos - s.com;
    V3 fr[3] = {n * sign, t1 * sign, t2 * sign};
    V3 col[9];
    col[0] = v3(1, 0, 0); col[1] = v3(0, 1, 0); col[2] = v3(0, 0, 1);
#pragma unroll
    for (int k = 0; k < 3; k++) col[3 + k] = s.cdr[k].l + cross(s.cdr[k].a, off);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      S6 cd = s.cdl[k];
      if (l == 1) cd = s.cdl[3 + k]; else if (l == 2) cd = s.cdl[6 + k]; else if (l == 3) cd = s.cdl[9 + k];
      col[6 + k] = cd.l + cross(cd.a, off);
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int k = 0; k < 9; k++) c.J[a][k] = dot(fr[a], col[k]);
  }

  PG_INL void finish_contact(Contact& c, const float* solref, const float* solimp, float includemargin, float invw_body) {
    float pos = c.dist - includemargin;
    c.row_active = pos < 0.f;
    float kimp, b, imp;
    kbi(m->timestep, solref, solimp, pos, kimp, b, imp);
    float mu = c.mu;
    float invweight = (invw_body + mu * mu * invw_body) * 2.0f * mu * mu / m->impratio;
    float r = fmaxf(invweight * (1.0f - imp) / imp, kMinVal);
    c.D = c.row_active ? 1.0f / r : 0.f;
    // J qvel per row
    float t[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      float x = 0.f;
#pragma unroll
      for (int k = 0; k < 6; k++) x += c.J[a][k] * s.qvel[k];
#pragma unroll
      for (int k = 0; k < 3; k++) x += c.J[a][6 + k] * sel4(c.leg, s.qvel[6 + k], s.qvel[9 + k], s.qvel[12 + k], s.qvel[15 + k]);
      t[a] = x;
    }
    float jv[4] = {t[0] + mu * t[1], t[0] - mu * t[1], t[0] + mu * t[2], t[0] - mu * t[2]};
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) c.aref[r4] = c.row_active ? (-b * jv[r4] - kimp * pos) : 0.f;
  }

  // ---- collision + constraint rows.  boxes: terrain table of this env's variant (nbox entries) or nullptr
  PG_INL void constraint_stage(const TerrainBox* __restrict__ boxes, int nbox, const float* __restrict__ box_fr, int N, int e) {
    // joint limits
#pragma unroll
    for (int j = 0; j < 12; j++) {
      float q = s.qpos[7 + j];
      float dmin = q - m->jnt_range[j][0], dmax = m->jnt_range[j][1] - q;
      float pos = fminf(dmin, dmax);
      bool act = pos < 0.f;
      s.lim_active[j] = act;
      s.lim_sign[j] = dmin < dmax ? 1.0f : -1.0f;
      float kimp, b, imp;
      kbi(m->timestep, m->jnt_solref, m->jnt_solimp, pos, kimp, b, imp);
      float r = fmaxf(m->dof_invweight0[6 + j] * (1.0f - imp) / imp, kMinVal);
      s.lim_D[j] = act ? 1.0f / r : 0.f;
      s.lim_aref[j] = act ? (-b * (s.lim_sign[j] * s.qvel[6 + j]) - kimp * pos) : 0.f;
    }
    // mixed contact parameters (mjx collision_driver): friction max, solref/solimp solmix-weighted, margin max
    auto mix = [&](const float* sr1, const float* si1, float sm1, const float* sr2, const float* si2, float sm2, float* sr, float* si) {
      float mixw = sm1 / (sm1 + sm2);
      if (sm1 < kMinVal && sm2 < kMinVal) mixw = 0.5f; else if (sm1 < kMinVal) mixw = 0.f; else if (sm2 < kMinVal) mixw = 1.f;
      if (sr1[0] > 0.f && sr2[0] > 0.f) { sr[0] = mixw * sr1[0] + (1 - mixw) * sr2[0]; sr[1] = mixw * sr1[1] + (1 - mixw) * sr2[1]; }
      else { sr[0] = fminf(sr1[0], sr2[0]); sr[1] = fminf(sr1[1], sr2[1]); }
#pragma unroll
      for (int i = 0; i < 5; i++) si[i] = mixw * si1[i] + (1 - mixw) * si2[i];
    };
    // plane-sphere slots 0..3 (leg order)
    {
      float sr[2], si[5];
      mix(m->floor_solref, m->floor_solimp, m->floor_solmix, m->foot_solref, m->foot_solimp, m->foot_solmix, sr, si);
      float margin = fmaxf(m->floor_margin, m->foot_margin) - fmaxf(m->floor_gap, m->foot_gap);
      float mu = fmaxf(em.floor_friction, m->foot_friction[0]);
#pragma unroll
      for (int l = 0; l < 4; l++) {
        Contact& c = s.con[l];
        float r = m->foot_radius[l];
        c.leg = l; c.box = -1; c.mu = mu;
        c.dist = s.footc[l].z - r;
        V3 pos = s.footc[l] - v3(0, 0, 1) * (r + 0.5f * c.dist);
        // frame of n = +z: t1 = (0,1,0), t2 = (-1,0,0); body2 = calf -> sign +1
        contact_jac(c, l, pos, v3(0, 0, 1), v3(0, 1, 0), v3(-1, 0, 0), 1.0f);
        finish_contact(c, sr, si, margin, m->body_invweight0[3 + 3 * l][0]);
      }
    }
    // sphere-box slots 4..7
#pragma unroll
    for (int k = 4; k < 8; k++) { s.con[k].leg = -1; s.con[k].box = -2; s.con[k].dist = 1.f; s.con[k].D = 0.f; s.con[k].row_active = false; s.con[k].mu = 0.f;
#pragma unroll
      for (int r4 = 0; r4 < 4; r4++) s.con[k].aref[r4] = 0.f;
      // an unused slot may still be visited (the slot count is wave-uniform): keep its Jacobian finite
#pragma unroll
      for (int a3 = 0; a3 < 3; a3++)
#pragma unroll
        for (int k9 = 0; k9 < 9; k9++) s.con[k].J[a3][k9] = 0.f; }
    s.ncon_box = 0;
    if (boxes == nullptr || nbox <= 0) return;
    const int maxp = m->max_geom_pairs, maxc = m->max_contact_points;
    const bool broad = maxp > -1 && 4 * nbox > maxp;
    PenPair pen[kMaxPen]; int npen = 0;
    float keyC[4];
#pragma unroll
    for (int l = 0; l < 4; l++) keyC[l] = m->foot_radius[l] + m->box_rbound;
#pragma unroll 4
    for (int b = 0; b < nbox; b++) {
      const float4 A = reinterpret_cast<const float4*>(boxes + b)[0];
#pragma unroll
      for (int l = 0; l < 4; l++) {
        V3 dv = v3(A.x, A.y, A.z) - s.footc[l];
        float dc = norm(dv);
        if (dc <= A.w + m->foot_radius[l] + 1e-5f) {        // necessary for penetration (bounding sphere)
          TerrainBox tb = boxes[b];
          float nd; V3 pw, nw;
          sphere_box(s.footc[l], m->foot_radius[l], tb, nd, pw, nw);
          if (nd < 0.f && npen < kMaxPen) {
            PenPair pp; pp.dist = nd; pp.key = dc - keyC[l]; pp.idx = l * nbox + b; pp.pos = pw; pp.n = nw;
            pen[npen++] = pp;
          }
        }
      }
    }
    if (__ballot(npen > 0) == 0ull) return;
    // exact broad-phase rank of every penetrating pair: #pairs with (key, idx) lexicographically smaller
    int rank[kMaxPen];
#pragma unroll
    for (int i = 0; i < kMaxPen; i++) rank[i] = 0;
    if (broad) {
#pragma unroll 4
      for (int b = 0; b < nbox; b++) {
        const float4 A = reinterpret_cast<const float4*>(boxes + b)[0];
#pragma unroll
        for (int l = 0; l < 4; l++) {
          V3 dv = v3(A.x, A.y, A.z) - s.footc[l];
          float key = norm(dv) - keyC[l];
          int idx = l * nbox + b;
#pragma unroll
          for (int i = 0; i < kMaxPen; i++)
            if (i < npen) rank[i] += (key < pen[i].key || (key == pen[i].key && idx < pen[i].idx)) ? 1 : 0;
        }
      }
    }
    // survivors (rank < max_geom_pairs), then the max_contact_points deepest, ties by broad-phase rank
    bool taken[kMaxPen];
#pragma unroll
    for (int i = 0; i < kMaxPen; i++) taken[i] = !(i < npen) || (broad && rank[i] >= maxp);
    float sr[2], si[5];
    mix(m->foot_solref, m->foot_solimp, m->foot_solmix, m->box_solref, m->box_solimp, m->box_solmix, sr, si);
    float margin = fmaxf(m->foot_margin, m->box_margin) - fmaxf(m->foot_gap, m->box_gap);
    int nslot = (maxc > -1 && maxc < 4) ? maxc : 4;
    for (int k = 0; k < nslot; k++) {
      int bi = -1;
#pragma unroll
      for (int i = 0; i < kMaxPen; i++) {
        if (taken[i]) continue;
        if (bi < 0 || pen[i].dist < pen[bi].dist || (pen[i].dist == pen[bi].dist && rank[i] < rank[bi])) bi = i;
      }
      if (bi < 0) break;
      taken[bi] = true;
      Contact& c = s.con[4 + k];
      int l = pen[bi].idx / nbox, b = pen[bi].idx - l * nbox;
      c.leg = l; c.box = b; c.dist = pen[bi].dist;
      float bf = box_fr ? box_fr[(long)b * N + e] : m->box_friction[0];
      c.mu = fmaxf(bf, m->foot_friction[0]);
      V3 n, t1, t2;
      make_frame(pen[bi].n, n, t1, t2);
      // geom1 = foot (calf), geom2 = box (static): jac_dif = -jac(calf)
      contact_jac(c, l, pen[bi].pos, n, t1, t2, -1.0f);
      finish_contact(c, sr, si, margin, sel4(l, m->body_invweight0[3][0], m->body_invweight0[6][0], m->body_invweight0[9][0], m->body_invweight0[12][0]));
      s.ncon_box = k + 1;
    }
  }
};

// ------------------------------------------------------------------ Newton solver on the sparse structures
struct Solver {
  const PgttModel* __restrict__ m;
  Sim& s;
  float qacc[18], Ma[18], grad[18], search[18], qfc[18];
  float jar_lim[12], jar_con[8][4];
  float gauss, cost, prev_cost;
  int nbox_slots;    // wave-uniform count of box slots that any lane uses

  PG_INL Solver(const PgttModel* m_, Sim& s_) : m(m_), s(s_) {}

  // rows of J x for contact c
  PG_INL void con_jx(const Contact& c, const float* x, float* out4) const {
    float t[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < 6; k++) v += c.J[a][k] * x[k];
#pragma unroll
      for (int k = 0; k < 3; k++) v += c.J[a][6 + k] * sel4(c.leg, x[6 + k], x[9 + k], x[12 + k], x[15 + k]);
      t[a] = v;
    }
    out4[0] = t[0] + c.mu * t[1]; out4[1] = t[0] - c.mu * t[1]; out4[2] = t[0] + c.mu * t[2]; out4[3] = t[0] - c.mu * t[2];
  }
  // out += J^T f for contact c (f: 4 pyramid-row forces)
  PG_INL void con_jtf(const Contact& c, const float* f, float* out) const {
    float g[3] = {f[0] + f[1] + f[2] + f[3], c.mu * (f[0] - f[1]), c.mu * (f[2] - f[3])};
#pragma unroll
    for (int k = 0; k < 6; k++) out[k] += c.J[0][k] * g[0] + c.J[1][k] * g[1] + c.J[2][k] * g[2];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      float v = c.J[0][6 + k] * g[0] + c.J[1][6 + k] * g[1] + c.J[2][6 + k] * g[2];
#pragma unroll
      for (int l = 0; l < 4; l++) out[6 + 3 * l + k] += (c.leg == l) ? v : 0.f;
    }
  }

  PG_INL void init(const float* q0) {
#pragma unroll
    for (int i = 0; i < 18; i++) qacc[i] = q0[i];
    arrow_mul(s.M, qacc, Ma);
#pragma unroll
    for (int j = 0; j < 12; j++) jar_lim[j] = s.lim_sign[j] * qacc[6 + j] * (s.lim_active[j] ? 1.f : 0.f) - s.lim_aref[j];
    for (int c = 0; c < 4 + nbox_slots; c++) {
      float jx[4];
      con_jx(s.con[c], qacc, jx);
#pragma unroll
      for (int r = 0; r < 4; r++) jar_con[c][r] = (s.con[c].row_active ? jx[r] : 0.f) - s.con[c].aref[r];
    }
    cost = INFINITY; prev_cost = 0.f;
  }

  PG_INL void update_constraint() {
    float csum = 0.f;
#pragma unroll
    for (int i = 0; i < 18; i++) qfc[i] = 0.f;
#pragma unroll
    for (int j = 0; j < 12; j++) {
      float ja = jar_lim[j];
      float f = ja < 0.f ? -s.lim_D[j] * ja : 0.f;
      qfc[6 + j] += s.lim_sign[j] * f;
      csum += ja < 0.f ? s.lim_D[j] * ja * ja : 0.f;
    }
    for (int c = 0; c < 4 + nbox_slots; c++) {
      float f[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float ja = jar_con[c][r];
        f[r] = ja < 0.f ? -s.con[c].D * ja : 0.f;
        csum += ja < 0.f ? s.con[c].D * ja * ja : 0.f;
      }
      con_jtf(s.con[c], f, qfc);
    }
    float g = 0.f;
#pragma unroll
    for (int i = 0; i < 18; i++) g += (Ma[i] - s.qfrc_smooth[i]) * (qacc[i] - s.qacc_smooth[i]);
    gauss = 0.5f * g;
    prev_cost = cost;
    cost = 0.5f * csum + gauss;
  }

  // grad, H = M + J^T diag(D active) J, search = -H^-1 grad
  PG_INL void update_gradient() {
#pragma unroll
    for (int i = 0; i < 18; i++) grad[i] = Ma[i] - s.qfrc_smooth[i] - qfc[i];
    Arrow H = s.M;
#pragma unroll
    for (int j = 0; j < 12; j++) H.ll[j / 3][tri(j % 3, j % 3)] += jar_lim[j] < 0.f ? s.lim_D[j] : 0.f;
    for (int c = 0; c < 4 + nbox_slots; c++) {
      const Contact& cn = s.con[c];
      float w[4];
#pragma unroll
      for (int r = 0; r < 4; r++) w[r] = jar_con[c][r] < 0.f ? cn.D : 0.f;
      // weights in the contact frame: sum_r w_r j_r j_r^T with j_r = Jn +- mu Jt
      float mu = cn.mu;
      float W00 = w[0] + w[1] + w[2] + w[3], W01 = mu * (w[0] - w[1]), W02 = mu * (w[2] - w[3]);
      float W11 = mu * mu * (w[0] + w[1]), W22 = mu * mu * (w[2] + w[3]);
      float T[3][9];
#pragma unroll
      for (int k = 0; k < 9; k++) {
        T[0][k] = W00 * cn.J[0][k] + W01 * cn.J[1][k] + W02 * cn.J[2][k];
        T[1][k] = W01 * cn.J[0][k] + W11 * cn.J[1][k];
        T[2][k] = W02 * cn.J[0][k] + W22 * cn.J[2][k];
      }
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) H.bb[tri(i, j)] += cn.J[0][i] * T[0][j] + cn.J[1][i] * T[1][j] + cn.J[2][i] * T[2][j];
      float G_lb[18], G_ll[6];
#pragma unroll
      for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int k = 0; k < 6; k++) G_lb[i * 6 + k] = cn.J[0][6 + i] * T[0][k] + cn.J[1][6 + i] * T[1][k] + cn.J[2][6 + i] * T[2][k];
#pragma unroll
        for (int j = 0; j <= i; j++) G_ll[tri(i, j)] = cn.J[0][6 + i] * T[0][6 + j] + cn.J[1][6 + i] * T[1][6 + j] + cn.J[2][6 + i] * T[2][6 + j];
      }
#pragma unroll
      for (int l = 0; l < 4; l++) {
        float on = (cn.leg == l) ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < 18; i++) H.lb[l][i] += on * G_lb[i];
#pragma unroll
        for (int i = 0; i < 6; i++) H.ll[l][i] += on * G_ll[i];
      }
    }
    arrow_factor(H);
    float mg[18];
    arrow_solve(H, grad, mg);
#pragma unroll
    for (int i = 0; i < 18; i++) search[i] = -mg[i];
  }

  struct LSPoint { float alpha, cost, d0, d1; };

  PG_INL void linesearch(bool frozen) {
    float sn = 0.f;
#pragma unroll
    for (int i = 0; i < 18; i++) sn += search[i] * search[i];
    float smag = sqrtf(sn) * m->meaninertia * 18.0f;
    float gtol = m->tolerance * m->ls_tolerance * smag;
    float mv[18];
    arrow_mul(s.M, search, mv);
    float jv_lim[12], jv_con[8][4];
#pragma unroll
    for (int j = 0; j < 12; j++) jv_lim[j] = s.lim_active[j] ? s.lim_sign[j] * search[6 + j] : 0.f;
    for (int c = 0; c < 4 + nbox_slots; c++) {
      float jx[4];
      con_jx(s.con[c], search, jx);
#pragma unroll
      for (int r = 0; r < 4; r++) jv_con[c][r] = s.con[c].row_active ? jx[r] : 0.f;
    }
    float a = 0.f, b = 0.f, e = 0.f;
#pragma unroll
    for (int i = 0; i < 18; i++) { a += search[i] * Ma[i]; b += search[i] * s.qfrc_smooth[i]; e += search[i] * mv[i]; }
    const float qg0 = gauss, qg1 = a - b, qg2 = 0.5f * e;
    auto point = [&](float alpha) {
      float q0 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
      for (int j = 0; j < 12; j++) {
        float x = jar_lim[j] + alpha * jv_lim[j];
        float d = x < 0.f ? s.lim_D[j] : 0.f;
        q0 += d * (0.5f * jar_lim[j] * jar_lim[j]); q1 += d * (jv_lim[j] * jar_lim[j]); q2 += d * (0.5f * jv_lim[j] * jv_lim[j]);
      }
      for (int c = 0; c < 4 + nbox_slots; c++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float ja = jar_con[c][r], jv = jv_con[c][r];
          float x = ja + alpha * jv;
          float d = x < 0.f ? s.con[c].D : 0.f;
          q0 += d * (0.5f * ja * ja); q1 += d * (jv * ja); q2 += d * (0.5f * jv * jv);
        }
      }
      q0 += qg0; q1 += qg1; q2 += qg2;
      LSPoint p;
      p.alpha = alpha;
      p.cost = alpha * alpha * q2 + alpha * q1 + q0;
      p.d0 = 2.0f * alpha * q2 + q1;
      p.d1 = 2.0f * q2 + (q2 == 0.f ? kMinVal : 0.f);
      return p;
    };
    auto in_bracket = [](const LSPoint& x, const LSPoint& y) {
      return ((x.d0 < y.d0) && (y.d0 < 0.f)) || ((x.d0 > y.d0) && (y.d0 > 0.f));
    };
    LSPoint p0 = point(0.f);
    LSPoint lo0 = point(p0.alpha - p0.d0 / p0.d1);
    bool lesser = lo0.d0 < p0.d0;
    LSPoint hi = lesser ? p0 : lo0, lo = lesser ? lo0 : p0;
    bool swap = true; int it = 0;
    for (;;) {
      bool done = it >= m->ls_iterations || !swap || ((lo.d0 < 0.f) && (lo.d0 > -gtol)) || ((hi.d0 > 0.f) && (hi.d0 < gtol));
      if (__ballot(!done) == 0ull) break;
      float al[3] = {lo.alpha - lo.d0 / lo.d1, hi.alpha - hi.d0 / hi.d1, 0.5f * (lo.alpha + hi.alpha)};
      LSPoint pt[3];
      for (int k = 0; k < 3; k++) pt[k] = point(al[k]);
      LSPoint nlo = lo, nhi = hi;
      bool s1 = in_bracket(nlo, pt[0]); if (s1) nlo = pt[0];
      bool s2 = in_bracket(nlo, pt[2]); if (s2) nlo = pt[2];
      bool s3 = in_bracket(nlo, pt[1]); if (s3) nlo = pt[1];
      bool t1 = in_bracket(nhi, pt[1]); if (t1) nhi = pt[1];
      bool t2 = in_bracket(nhi, pt[2]); if (t2) nhi = pt[2];
      bool t3 = in_bracket(nhi, pt[0]); if (t3) nhi = pt[0];
      if (!done) { lo = nlo; hi = nhi; swap = s1 | s2 | s3 | t1 | t2 | t3; it++; }
    }
    bool improved = (lo.cost < p0.cost) || (hi.cost < p0.cost);
    float alpha = lo.cost < hi.cost ? lo.alpha : hi.alpha;
    float ia = (improved && !frozen) ? alpha : 0.f;
#pragma unroll
    for (int i = 0; i < 18; i++) { qacc[i] += search[i] * ia; Ma[i] += mv[i] * ia; }
#pragma unroll
    for (int j = 0; j < 12; j++) jar_lim[j] += jv_lim[j] * ia;
    for (int c = 0; c < 4 + nbox_slots; c++) {
#pragma unroll
      for (int r = 0; r < 4; r++) jar_con[c][r] += jv_con[c][r] * ia;
    }
  }

  PG_INL void solve() {
    // wave-uniform number of box-contact slots in use (ballot => skip empty slots for the whole wave)
    int nb = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) if (__ballot(s.ncon_box > k) != 0ull) nb = k + 1;
    nbox_slots = nb;
    // warm start: the cheaper of qacc_warmstart and qacc_smooth
    init(s.warm); update_constraint();
    float cw = cost;
    init(s.qacc_smooth); update_constraint();
    if (cw < cost) { init(s.warm); update_constraint(); }
    update_gradient();
    const float scale = m->meaninertia * 18.0f;
    int niter = 0;
    for (;;) {
      float gn = 0.f;
#pragma unroll
      for (int i = 0; i < 18; i++) gn += grad[i] * grad[i];
      bool done = niter >= m->iterations || ((prev_cost - cost) / scale < m->tolerance) || (sqrtf(gn) / scale < m->tolerance);
      if (__ballot(!done) == 0ull) break;
      // lanes whose solve is finished take a zero-length step: every quantity is then recomputed from unchanged
      // inputs (bitwise identical), improvement becomes 0 and the lane stays finished
      linesearch(done);
      update_constraint();
      update_gradient();
      if (!done) niter++;
    }
#pragma unroll
    for (int i = 0; i < 18; i++) { s.qacc[i] = qacc[i]; s.warm[i] = qacc[i]; }
    s.niter = niter; s.niter_max = niter > s.niter_max ? niter : s.niter_max;
  }
};

}  // namespace pgtt
