/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or executed from the product
 * path (libpgtt.so / phase_guided_terrain_traversal_amd).  Allowed users: tests/, __graft_entry__.smoke(),
 * bench.py's cpu_baseline leg.
 *
 * PARITY UNPINNED at the physics boundary: the arithmetic restated here lives in MuJoCo-MJX /
 * MuJoCo Playground, which are neither vendored in /root/reference nor pinned (no requirements /
 * lockfile; snapshot 2025-07-04 => presumably mujoco/mujoco-mjx 3.3.x, playground 0.0.4-0.0.5) and
 * are not installable here.  This file restates the published MJX algorithm for the call sites
 *   go2/joystick_pgtt.py:72,78   mjx_env.init / mjx.forward
 *   go2/joystick_pgtt.py:146-148 mjx_env.step(model, data, motor_targets, n_substeps)
 *   go2/base.py:153-171          collision.geoms_colliding over data.contact
 *   go2/base.py:116-149          mjx_env.get_sensor_data (sensordata slices)
 * in the DENSE formulation MJX uses for nv=18 (<60 dofs => dense qM, dense efc_J, cho_factor).
 * It is pinned only by physics invariants and hand-derived KATs (tests/test_oracle_physics.py): no step-level vectors of the
 * reference exist.  At DISTRIBUTION level the same arithmetic (through the HIP kernels, which this file checks step by step) is held
 * to the reference's own simulator by the statistics its training run left in policy_folder/policy177's normaliser (443 M samples,
 * 215 rows; tests/test_gpu_policy.py) - that evidence is what replaced the recalled frame flip of _sphere_convex
 * for a sphere centre inside the box (sphere_box below).
 *
 * This header is included twice by pgtt_oracle.c: REAL=float (suffix _f32) and REAL=double (_f64).
 *
 * Body numbering follows MuJoCo: 0 world, 1 base, 2+3*leg+{0,1,2} = hip,thigh,calf; legs FL,FR,RL,RR.
 * Spatial vectors are [angular(3), linear(3)] about the robot's subtree COM (MJX convention).
 */

#define OB 14          /* bodies incl. world */
#define ONV 18
#define ONEFC 44
#define ONCON 8
#define MJ_MINVAL 1e-15
#define MJ_MINIMP 0.0001
#define MJ_MAXIMP 0.9999

typedef struct F(OParams) {       /* per-env (domain-randomised) model fields, randomize.py:23-171 */
  R body_mass[13];
  R base_ipos[3];
  R qpos0j[12];
  R armature[12];
  R damping[12];
  R gain[12];
  R bias1[12];
  R floor_friction;
} F(OParams);

typedef struct F(OContact) {
  R dist;
  R pos[3];
  R frame[9];      /* rows: normal, tangent1, tangent2 */
  R friction[5];
  R solref[2];
  R solimp[5];
  R includemargin;
  int geom1, geom2; /* plane contacts: (-1, foot leg); box contacts: (foot leg, box index) */
  int foot;         /* leg index 0..3 (FL,FR,RL,RR) */
  int box;          /* -1 plane, else box index */
} F(OContact);

typedef struct F(OData) {
  /* state (in/out) */
  R qpos[19], qvel[18], qacc_warmstart[18], ctrl[12];
  /* position-dependent */
  R xpos[OB][3], xquat[OB][4], xmat[OB][9], xipos[OB][3], ximat[OB][9];
  R com[3];
  R cinert[OB][10];
  R cdof[ONV][6];
  R crb[OB][10];
  R qM[ONV][ONV];
  R qLD[ONV][ONV];               /* lower Cholesky factor of qM */
  R foot_xpos[4][3];             /* geom centres, leg order */
  R site_foot[4][3];
  R site_imu[3];
  R site_imu_mat[9];
  F(OContact) contact[ONCON];
  /* velocity-dependent */
  R cvel[OB][6], cdof_dot[ONV][6];
  R qfrc_bias[ONV], qfrc_passive[ONV], qfrc_actuator[ONV], actuator_force[12];
  R qfrc_smooth[ONV], qacc_smooth[ONV];
  /* constraints */
  R efc_J[ONEFC][ONV], efc_D[ONEFC], efc_aref[ONEFC], efc_pos[ONEFC];
  int efc_active_row[ONEFC];     /* row instantiated (limit violated / dist < margin) */
  /* solver outputs */
  R qacc[ONV], efc_force[ONEFC], qfrc_constraint[ONV];
  int solver_niter;
  int solver_niter_max;          /* max over the substeps of env_step */
  int diag_flags;                /* OR over the substeps of env_step: 1 = stale and fresh rbound pick different pair sets, 2 = a sphere centre inside a box, 4 = the cut was active, 8 = a sphere centre on a box surface to 1e-6 */
  R solver_resid, solver_resid_max;   /* scaled gradient norm at the solver's exit (diagnostic: how far from the minimiser the cut solve is) */
  R cacc_base[6];
  R sensordata[49];
} F(OData);

/* ------------------------------------------------------------------ small math (mjx/_src/math.py) */
static inline R F(dot3)(const R* a, const R* b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
static inline void F(cross3)(const R* a, const R* b, R* o) {
  R x = a[1]*b[2] - a[2]*b[1], y = a[2]*b[0] - a[0]*b[2], z = a[0]*b[1] - a[1]*b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline R F(norm3)(const R* a) { return SQRT(a[0]*a[0] + a[1]*a[1] + a[2]*a[2]); }

static inline void F(quat_mul)(const R* u, const R* v, R* o) {
  R w = u[0]*v[0] - u[1]*v[1] - u[2]*v[2] - u[3]*v[3];
  R x = u[0]*v[1] + u[1]*v[0] + u[2]*v[3] - u[3]*v[2];
  R y = u[0]*v[2] - u[1]*v[3] + u[2]*v[0] + u[3]*v[1];
  R z = u[0]*v[3] + u[1]*v[2] - u[2]*v[1] + u[3]*v[0];
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
static inline void F(quat_to_mat)(const R* q, R* m) {
  R q00 = q[0]*q[0], q01 = q[0]*q[1], q02 = q[0]*q[2], q03 = q[0]*q[3];
  R q11 = q[1]*q[1], q12 = q[1]*q[2], q13 = q[1]*q[3];
  R q22 = q[2]*q[2], q23 = q[2]*q[3], q33 = q[3]*q[3];
  m[0] = q00 + q11 - q22 - q33; m[1] = 2*(q12 - q03);         m[2] = 2*(q13 + q02);
  m[3] = 2*(q12 + q03);         m[4] = q00 - q11 + q22 - q33; m[5] = 2*(q23 - q01);
  m[6] = 2*(q13 - q02);         m[7] = 2*(q23 + q01);         m[8] = q00 - q11 - q22 + q33;
}
/* math.rotate(vec, quat) */
static inline void F(rotate)(const R* v, const R* q, R* o) {
  R s = q[0]; const R* u = q + 1;
  R uv = F(dot3)(u, v), uu = F(dot3)(u, u);
  R c[3]; F(cross3)(u, v, c);
  for (int i = 0; i < 3; i++) o[i] = 2*(uv*u[i]) + (s*s - uu)*v[i] + 2*s*c[i];
}
static inline void F(mat_mul_vec)(const R* m, const R* v, R* o) {
  R x = m[0]*v[0] + m[1]*v[1] + m[2]*v[2], y = m[3]*v[0] + m[4]*v[1] + m[5]*v[2], z = m[6]*v[0] + m[7]*v[1] + m[8]*v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void F(matT_mul_vec)(const R* m, const R* v, R* o) {
  R x = m[0]*v[0] + m[3]*v[1] + m[6]*v[2], y = m[1]*v[0] + m[4]*v[1] + m[7]*v[2], z = m[2]*v[0] + m[5]*v[1] + m[8]*v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
/* math.normalize_with_norm: zero vectors (allclose to 0, atol 1e-8) are left untouched with norm 0 */
static inline R F(normalize_n)(R* x, int n) {
  int is_zero = 1;
  for (int i = 0; i < n; i++) if (FABS(x[i]) > (R)1e-8) is_zero = 0;
  if (is_zero) { for (int i = 0; i < n; i++) x[i] = 1; }
  R s = 0; for (int i = 0; i < n; i++) s += x[i]*x[i];
  R nn = SQRT(s);
  for (int i = 0; i < n; i++) x[i] = x[i] / (nn + (is_zero ? (R)1 : (R)0));
  return is_zero ? (R)0 : nn;
}
static inline void F(axis_angle_to_quat)(const R* axis, R angle, R* q) {
  R s = SIN(angle * (R)0.5), c = COS(angle * (R)0.5);
  q[0] = c; q[1] = axis[0]*s; q[2] = axis[1]*s; q[3] = axis[2]*s;
}
/* math.inert_mul: inertia (10) x motion (6) -> force (6) */
static inline void F(inert_mul)(const R* i, const R* v, R* o) {
  /* tri_id = [[0,3,4],[3,1,5],[4,5,2]] */
  R a0 = i[0]*v[0] + i[3]*v[1] + i[4]*v[2];
  R a1 = i[3]*v[0] + i[1]*v[1] + i[5]*v[2];
  R a2 = i[4]*v[0] + i[5]*v[1] + i[2]*v[2];
  R c[3]; F(cross3)(i + 6, v + 3, c);
  R d[3]; F(cross3)(i + 6, v, d);
  o[0] = a0 + c[0]; o[1] = a1 + c[1]; o[2] = a2 + c[2];
  o[3] = i[9]*v[3] - d[0]; o[4] = i[9]*v[4] - d[1]; o[5] = i[9]*v[5] - d[2];
}
static inline void F(motion_cross)(const R* u, const R* v, R* o) {
  R a[3], b[3], c[3];
  F(cross3)(u, v, a); F(cross3)(u + 3, v, b); F(cross3)(u, v + 3, c);
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = b[0] + c[0]; o[4] = b[1] + c[1]; o[5] = b[2] + c[2];
}
static inline void F(motion_cross_force)(const R* v, const R* f, R* o) {
  R a[3], b[3], c[3];
  F(cross3)(v, f, a); F(cross3)(v + 3, f + 3, b); F(cross3)(v, f + 3, c);
  o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; o[3] = c[0]; o[4] = c[1]; o[5] = c[2];
}
/* math.make_frame */
static inline void F(make_frame)(const R* a_in, R* frame) {
  R a[3] = {a_in[0], a_in[1], a_in[2]};
  F(normalize_n)(a, 3);
  R y[3] = {0, 0, 0};
  if (a[1] > (R)-0.5 && a[1] < (R)0.5) y[1] = 1; else y[2] = 1;
  R ay = F(dot3)(a, y);
  R b[3] = {y[0] - a[0]*ay, y[1] - a[1]*ay, y[2] - a[2]*ay};
  F(normalize_n)(b, 3);
  R c[3]; F(cross3)(a, b, c);
  for (int i = 0; i < 3; i++) { frame[i] = a[i]; frame[3 + i] = b[i]; frame[6 + i] = c[i]; }
}

static inline int F(body_parent)(int b) { return b <= 1 ? 0 : (((b - 2) % 3 == 0) ? 1 : b - 1); }
static inline int F(dof_body)(int d) { return d < 6 ? 1 : d - 4; }
static inline int F(dof_parent)(int d) { return d == 0 ? -1 : (d < 6 ? d - 1 : (((d - 6) % 3 == 0) ? 5 : d - 1)); }

/* ------------------------------------------------------------------ mjx.smooth.kinematics + com_pos */
static void F(kinematics)(const PgttModel* m, const F(OParams)* p, F(OData)* d) {
  R* q = d->qpos;
  /* world */
  for (int i = 0; i < 3; i++) d->xpos[0][i] = 0;
  d->xquat[0][0] = 1; d->xquat[0][1] = d->xquat[0][2] = d->xquat[0][3] = 0;
  /* base: free joint; qpos quaternion is normalised in place */
  for (int i = 0; i < 3; i++) d->xpos[1][i] = q[i];
  F(normalize_n)(q + 3, 4);
  for (int i = 0; i < 4; i++) d->xquat[1][i] = q[3 + i];
  for (int b = 2; b < OB; b++) {
    int par = F(body_parent)(b), mb = b - 1, j = b - 2;
    R bp[3] = {(R)m->body_pos[mb][0], (R)m->body_pos[mb][1], (R)m->body_pos[mb][2]};
    R bq[4] = {(R)m->body_quat[mb][0], (R)m->body_quat[mb][1], (R)m->body_quat[mb][2], (R)m->body_quat[mb][3]};
    R pos[3], quat[4], r[3];
    F(rotate)(bp, d->xquat[par], r);
    for (int i = 0; i < 3; i++) pos[i] = d->xpos[par][i] + r[i];
    F(quat_mul)(d->xquat[par], bq, quat);
    R axis[3] = {(R)m->jnt_axis[j][0], (R)m->jnt_axis[j][1], (R)m->jnt_axis[j][2]};
    R angle = q[7 + j] - p->qpos0j[j];
    R qloc[4];
    F(axis_angle_to_quat)(axis, angle, qloc);
    F(quat_mul)(quat, qloc, d->xquat[b]);
    /* pos = anchor - rotate(jnt_pos=0, quat) = pos */
    for (int i = 0; i < 3; i++) d->xpos[b][i] = pos[i];
  }
  for (int b = 0; b < OB; b++) {
    F(quat_to_mat)(d->xquat[b], d->xmat[b]);
    if (b == 0) {
      for (int i = 0; i < 3; i++) d->xipos[0][i] = 0;
      F(quat_to_mat)(d->xquat[0], d->ximat[0]);
      continue;
    }
    int mb = b - 1;
    R ip[3] = {(R)m->body_ipos[mb][0], (R)m->body_ipos[mb][1], (R)m->body_ipos[mb][2]};
    if (b == 1) { ip[0] = p->base_ipos[0]; ip[1] = p->base_ipos[1]; ip[2] = p->base_ipos[2]; }
    R iq[4] = {(R)m->body_iquat[mb][0], (R)m->body_iquat[mb][1], (R)m->body_iquat[mb][2], (R)m->body_iquat[mb][3]};
    R r[3], qq[4];
    F(rotate)(ip, d->xquat[b], r);
    for (int i = 0; i < 3; i++) d->xipos[b][i] = d->xpos[b][i] + r[i];
    F(quat_mul)(d->xquat[b], iq, qq);
    F(quat_to_mat)(qq, d->ximat[b]);
  }
  /* geoms / sites: support.local_to_global */
  for (int l = 0; l < 4; l++) {
    int b = 4 + 3*l;
    R gp[3] = {(R)m->foot_geom_pos[l][0], (R)m->foot_geom_pos[l][1], (R)m->foot_geom_pos[l][2]};
    R sp[3] = {(R)m->foot_site_pos[l][0], (R)m->foot_site_pos[l][1], (R)m->foot_site_pos[l][2]};
    R r[3];
    F(rotate)(gp, d->xquat[b], r);
    for (int i = 0; i < 3; i++) d->foot_xpos[l][i] = d->xpos[b][i] + r[i];
    F(rotate)(sp, d->xquat[b], r);
    for (int i = 0; i < 3; i++) d->site_foot[l][i] = d->xpos[b][i] + r[i];
  }
  {
    R ip[3] = {(R)m->imu_pos[0], (R)m->imu_pos[1], (R)m->imu_pos[2]}, r[3], one[4] = {1, 0, 0, 0}, qq[4];
    F(rotate)(ip, d->xquat[1], r);
    for (int i = 0; i < 3; i++) d->site_imu[i] = d->xpos[1][i] + r[i];
    F(quat_mul)(d->xquat[1], one, qq);
    F(quat_to_mat)(qq, d->site_imu_mat);
  }
}

static void F(com_pos)(const PgttModel* m, const F(OParams)* p, F(OData)* d) {
  (void)m;
  /* subtree sums in MJX's reverse body_tree order: calf -> thigh -> hip -> base (children summed by id) */
  R carry[3] = {0, 0, 0}, cmass = 0;
  for (int l = 0; l < 4; l++) {
    R sp[3] = {0, 0, 0}, sm = 0;
    for (int k = 2; k >= 0; k--) {
      int b = 2 + 3*l + k; R mass = p->body_mass[b - 1];
      R own[3] = {d->xipos[b][0]*mass, d->xipos[b][1]*mass, d->xipos[b][2]*mass};
      if (k == 2) { sp[0] = own[0]; sp[1] = own[1]; sp[2] = own[2]; sm = mass; }
      else { sp[0] = own[0] + sp[0]; sp[1] = own[1] + sp[1]; sp[2] = own[2] + sp[2]; sm = mass + sm; }
    }
    if (l == 0) { carry[0] = sp[0]; carry[1] = sp[1]; carry[2] = sp[2]; cmass = sm; }
    else { carry[0] += sp[0]; carry[1] += sp[1]; carry[2] += sp[2]; cmass += sm; }
  }
  R bm = p->body_mass[0];
  R tot[3] = {d->xipos[1][0]*bm + carry[0], d->xipos[1][1]*bm + carry[1], d->xipos[1][2]*bm + carry[2]};
  R tm = bm + cmass;
  R den = tm > (R)MJ_MINVAL ? tm : (R)MJ_MINVAL;
  for (int i = 0; i < 3; i++) d->com[i] = (tm < (R)MJ_MINVAL) ? d->xipos[1][i] : tot[i] / den;

  /* cinert about the root subtree COM */
  for (int i = 0; i < 10; i++) d->cinert[0][i] = 0;
  for (int b = 1; b < OB; b++) {
    R mass = p->body_mass[b - 1];
    R off[3] = {d->xipos[b][0] - d->com[0], d->xipos[b][1] - d->com[1], d->xipos[b][2] - d->com[2]};
    const R* xm = d->ximat[b];
    R in[3] = {(R)m->body_inertia[b - 1][0], (R)m->body_inertia[b - 1][1], (R)m->body_inertia[b - 1][2]};
    /* h = cross(off, -eye(3)) rows; inert = (ximat*inert) @ ximat.T + h @ h.T * mass */
    R h[3][3];
    for (int r = 0; r < 3; r++) { R e[3] = {0, 0, 0}; e[r] = -1; F(cross3)(off, e, h[r]); }
    R I[3][3];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
      R s = 0;
      for (int k = 0; k < 3; k++) s += (xm[3*r + k]*in[k]) * xm[3*c + k];
      R hh = 0;
      for (int k = 0; k < 3; k++) hh += h[r][k]*h[c][k];
      I[r][c] = s + hh*mass;
    }
    R* ci = d->cinert[b];
    ci[0] = I[0][0]; ci[1] = I[1][1]; ci[2] = I[2][2]; ci[3] = I[0][1]; ci[4] = I[0][2]; ci[5] = I[1][2];
    ci[6] = off[0]*mass; ci[7] = off[1]*mass; ci[8] = off[2]*mass; ci[9] = mass;
  }
  /* cdof */
  for (int i = 0; i < 3; i++) { for (int k = 0; k < 6; k++) d->cdof[i][k] = 0; d->cdof[i][3 + i] = 1; }
  {
    R off[3] = {d->com[0] - d->qpos[0], d->com[1] - d->qpos[1], d->com[2] - d->qpos[2]};
    for (int i = 0; i < 3; i++) {
      R a[3] = {d->xmat[1][i], d->xmat[1][3 + i], d->xmat[1][6 + i]}, c[3];  /* row i of xmat.T */
      F(cross3)(a, off, c);
      for (int k = 0; k < 3; k++) { d->cdof[3 + i][k] = a[k]; d->cdof[3 + i][3 + k] = c[k]; }
    }
  }
  for (int j = 0; j < 12; j++) {
    int b = 2 + j, par = F(body_parent)(b);
    /* axis rotated by the body quaternion BEFORE the joint rotation (parent_quat * body_quat) */
    R bq[4] = {(R)m->body_quat[b - 1][0], (R)m->body_quat[b - 1][1], (R)m->body_quat[b - 1][2], (R)m->body_quat[b - 1][3]};
    R quat[4]; F(quat_mul)(d->xquat[par], bq, quat);
    R axis[3] = {(R)m->jnt_axis[j][0], (R)m->jnt_axis[j][1], (R)m->jnt_axis[j][2]}, a[3], c[3];
    F(rotate)(axis, quat, a);
    R off[3] = {d->com[0] - d->xpos[b][0], d->com[1] - d->xpos[b][1], d->com[2] - d->xpos[b][2]};
    F(cross3)(a, off, c);
    for (int k = 0; k < 3; k++) { d->cdof[6 + j][k] = a[k]; d->cdof[6 + j][3 + k] = c[k]; }
  }
}

/* ------------------------------------------------------------------ crb + factor_m */
static int F(cholesky)(const R A[ONV][ONV], R L[ONV][ONV]) {
  for (int i = 0; i < ONV; i++) for (int j = 0; j < ONV; j++) L[i][j] = 0;
  for (int j = 0; j < ONV; j++) {
    R s = A[j][j];
    for (int k = 0; k < j; k++) s -= L[j][k]*L[j][k];
    if (!(s > 0)) return -1;
    R ljj = SQRT(s);
    L[j][j] = ljj;
    for (int i = j + 1; i < ONV; i++) {
      R t = A[i][j];
      for (int k = 0; k < j; k++) t -= L[i][k]*L[j][k];
      L[i][j] = t / ljj;
    }
  }
  return 0;
}
static void F(cho_solve)(const R L[ONV][ONV], const R* b, R* x) {
  R y[ONV];
  for (int i = 0; i < ONV; i++) { R s = b[i]; for (int k = 0; k < i; k++) s -= L[i][k]*y[k]; y[i] = s / L[i][i]; }
  for (int i = ONV - 1; i >= 0; i--) { R s = y[i]; for (int k = i + 1; k < ONV; k++) s -= L[k][i]*x[k]; x[i] = s / L[i][i]; }
}

static void F(crb)(const PgttModel* m, const F(OParams)* p, F(OData)* d) {
  (void)m;
  for (int b = 0; b < OB; b++) for (int i = 0; i < 10; i++) d->crb[b][i] = d->cinert[b][i];
  for (int l = 0; l < 4; l++) {
    for (int k = 2; k >= 1; k--) { int b = 2 + 3*l + k; for (int i = 0; i < 10; i++) d->crb[b - 1][i] += d->crb[b][i]; }
  }
  {
    R carry[10];
    for (int i = 0; i < 10; i++) carry[i] = d->crb[2][i];
    for (int l = 1; l < 4; l++) for (int i = 0; i < 10; i++) carry[i] += d->crb[2 + 3*l][i];
    for (int i = 0; i < 10; i++) d->crb[1][i] += carry[i];
  }
  for (int i = 0; i < 10; i++) d->crb[0][i] = 0;
  R crb_cdof[ONV][6];
  for (int i = 0; i < ONV; i++) F(inert_mul)(d->crb[F(dof_body)(i)], d->cdof[i], crb_cdof[i]);
  for (int i = 0; i < ONV; i++) for (int j = 0; j < ONV; j++) d->qM[i][j] = 0;
  for (int i = 0; i < ONV; i++) {
    int j = i;
    while (j > -1) {
      R s = 0;
      for (int k = 0; k < 6; k++) s += crb_cdof[i][k]*d->cdof[j][k];
      d->qM[i][j] = s;
      if (i != j) d->qM[j][i] = s;
      j = F(dof_parent)(j);
    }
  }
  for (int j = 0; j < 12; j++) d->qM[6 + j][6 + j] += p->armature[j];
  F(cholesky)(d->qM, d->qLD);
}

/* ------------------------------------------------------------------ collision (mjx collision_driver / collision_primitive / collision_convex) */
static void F(mix_params)(const float* fr1, const float* sr1, const float* si1, float mg1, float gp1, float sm1,
                          const R* fr2, const float* sr2, const float* si2, float mg2, float gp2, float sm2,
                          F(OContact)* c) {
  R f0 = (R)fr1[0] > fr2[0] ? (R)fr1[0] : fr2[0];
  R f1 = (R)fr1[1] > fr2[1] ? (R)fr1[1] : fr2[1];
  R f2 = (R)fr1[2] > fr2[2] ? (R)fr1[2] : fr2[2];
  c->friction[0] = f0; c->friction[1] = f0; c->friction[2] = f1; c->friction[3] = f2; c->friction[4] = f2;
  R s1 = sm1, s2 = sm2, mix = s1 / (s1 + s2);
  if (s1 < (R)MJ_MINVAL && s2 < (R)MJ_MINVAL) mix = (R)0.5;
  else if (s1 < (R)MJ_MINVAL) mix = 0;
  else if (s2 < (R)MJ_MINVAL) mix = 1;
  if (sr1[0] > 0 && sr2[0] > 0) { for (int i = 0; i < 2; i++) c->solref[i] = mix*(R)sr1[i] + (1 - mix)*(R)sr2[i]; }
  else { for (int i = 0; i < 2; i++) c->solref[i] = (R)(sr1[i] < sr2[i] ? sr1[i] : sr2[i]); }
  for (int i = 0; i < 5; i++) c->solimp[i] = mix*(R)si1[i] + (1 - mix)*(R)si2[i];
  R margin = (R)(mg1 > mg2 ? mg1 : mg2), gap = (R)(gp1 > gp2 ? gp1 : gp2);
  c->includemargin = margin - gap;
}

/* sphere (in world) vs box given by (pos, mat row-major, half-size): collision_convex._sphere_convex */
static void F(sphere_box)(const R* spos, R radius, const R* bpos, const R* bmat, const R* size,
                          R* dist_out, R* pos_out, R* n_out) {
  static const int vsign[8][3] = {{-1,-1,-1},{-1,-1,1},{-1,1,-1},{-1,1,1},{1,-1,-1},{1,-1,1},{1,1,-1},{1,1,1}};
  static const int faces[6][4] = {{0,4,5,1},{0,2,6,4},{6,7,5,4},{2,3,7,6},{1,5,7,3},{0,1,3,2}};
  static const R normals[6][3] = {{0,-1,0},{0,0,-1},{1,0,0},{0,1,0},{0,0,1},{-1,0,0}};
  R rel[3] = {spos[0] - bpos[0], spos[1] - bpos[1], spos[2] - bpos[2]}, c[3];
  F(matT_mul_vec)(bmat, rel, c);
  R support[6]; int best = 0;
  for (int f = 0; f < 6; f++) {
    const int* v0 = vsign[faces[f][0]];
    R p[3];
    for (int i = 0; i < 3; i++) p[i] = (c[i] - normals[f][i]*radius) - (R)v0[i]*size[i];
    support[f] = F(dot3)(p, normals[f]);
    if (support[f] >= 0) support[f] = (R)-1e12;
  }
  for (int f = 1; f < 6; f++) if (support[f] > support[best]) best = f;
  const R* fn = normals[best];
  R face[4][3];
  for (int k = 0; k < 4; k++) for (int i = 0; i < 3; i++) face[k][i] = (R)vsign[faces[best][k]][i]*size[i];
  /* project centre onto the face plane */
  R rp[3] = {c[0] - face[0][0], c[1] - face[0][1], c[2] - face[0][2]};
  R dd = F(dot3)(rp, fn);
  R pt[3] = {c[0] - dd*fn[0], c[1] - dd*fn[1], c[2] - dd*fn[2]};
  R edge_dist[4]; R en[4][3]; int inside = 1;
  for (int k = 0; k < 4; k++) {
    const R* p0 = face[(k + 3) % 4]; const R* p1 = face[k];
    R e[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    F(cross3)(e, fn, en[k]);
    R r[3] = {pt[0] - p0[0], pt[1] - p0[1], pt[2] - p0[2]};
    edge_dist[k] = F(dot3)(r, en[k]);
    if (!(edge_dist[k] <= 0)) inside = 0;
  }
  int idx = 0; R bestd = 0;
  for (int k = 0; k < 4; k++) {
    int degenerate = (en[k][0] == 0 && en[k][1] == 0 && en[k][2] == 0);
    R ed = (degenerate || edge_dist[k] < 0) ? (R)1e12 : edge_dist[k];
    if (k == 0 || ed < bestd) { bestd = ed; idx = k; }
  }
  /* math.closest_segment_point(a, b, pt) */
  const R* a = face[(idx + 3) % 4]; const R* b = face[idx];
  R ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
  R ap[3] = {pt[0] - a[0], pt[1] - a[1], pt[2] - a[2]};
  R t = F(dot3)(ap, ab) / (F(dot3)(ab, ab) + (R)1e-6);
  t = t < 0 ? 0 : (t > 1 ? 1 : t);
  R ept[3] = {a[0] + t*ab[0], a[1] + t*ab[1], a[2] + t*ab[2]};
  if (!inside) { pt[0] = ept[0]; pt[1] = ept[1]; pt[2] = ept[2]; }
  R n[3] = {pt[0] - c[0], pt[1] - c[1], pt[2] - c[2]};
  R dn = F(normalize_n)(n, 3);
#ifndef PGTT_SPHERE_CONVEX_FLIP
  /* centre INSIDE the box (deeper than one radius): keep the inward normal of the least-penetrated face and a growing depth.  The recalled
     `normalize(pt - c)` would flip the frame and lose the contact; the reference's own training statistics (policy177's normaliser)
     rule that out - see DESIGN.md 2 / 9.  -DPGTT_SPHERE_CONVEX_FLIP restores the recalled variant. */
  if (FABS(c[0]) <= size[0] && FABS(c[1]) <= size[1] && FABS(c[2]) <= size[2]) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; dn = -dn; }
#endif
  R spt[3] = {c[0] + n[0]*radius, c[1] + n[1]*radius, c[2] + n[2]*radius};
  *dist_out = dn - radius;
  R pl[3] = {(pt[0] + spt[0])*(R)0.5, (pt[1] + spt[1])*(R)0.5, (pt[2] + spt[2])*(R)0.5};
  F(mat_mul_vec)(bmat, n, n_out);
  R pw[3]; F(mat_mul_vec)(bmat, pl, pw);
  for (int i = 0; i < 3; i++) pos_out[i] = pw[i] + bpos[i];
}

/* broad phase of the max_geom_pairs cut: dist = |pos2 - pos1| - (rbound1 + rbound2), top_k(-dist, maxp), ties -> lower index first (lax.top_k).
   fresh = 0: the STALE compiled rbound of the placeholder boxes (what MJX sees after go2/randomize.py:97-108 replaced geom_size but not geom_rbound:
   the reading the product and this oracle use); fresh = 1: every box's own bounding radius |half-size| (diagnostic switch only) */
static int F(broad_phase)(const PgttModel* m, const F(OData)* d, const float* boxes, int nbox, int maxp, int fresh, int* sel) {
  int npair = 4*nbox, nsel = 0;
  R key[4*PGTT_MAX_BOX]; unsigned char used[4*PGTT_MAX_BOX];
  for (int l = 0; l < 4; l++) for (int b = 0; b < nbox; b++) {
    const float* bx = boxes + 10*b;
    R dv[3] = {(R)bx[0] - d->foot_xpos[l][0], (R)bx[1] - d->foot_xpos[l][1], (R)bx[2] - d->foot_xpos[l][2]};
    R hs[3] = {(R)bx[7], (R)bx[8], (R)bx[9]};
    key[l*nbox + b] = F(norm3)(dv) - ((R)m->foot_radius[l] + (fresh ? F(norm3)(hs) : (R)m->box_rbound));
    used[l*nbox + b] = 0;
  }
  for (int s = 0; s < maxp; s++) {
    int bi = -1;
    for (int i = 0; i < npair; i++) if (!used[i] && (bi < 0 || key[i] < key[bi])) bi = i;
    used[bi] = 1; sel[nsel++] = bi;
  }
  return nsel;
}

/* boxes: nbox x 10 [pos, quat, half-size]; box_friction: nbox sliding frictions or NULL */
static void F(collision)(const PgttModel* m, const F(OParams)* p, const float* boxes, const float* box_friction,
                         int nbox, F(OData)* d) {
  /* group 1: plane-sphere, pairs (floor, FL), (floor, FR), (floor, RL), (floor, RR); 4 <= max_contact_points */
  for (int l = 0; l < 4; l++) {
    F(OContact)* c = &d->contact[l];
    R n[3] = {0, 0, 1};
    R r = (R)m->foot_radius[l];
    R dist = F(dot3)(d->foot_xpos[l], n) - r;   /* plane at the origin, normal +z */
    for (int i = 0; i < 3; i++) c->pos[i] = d->foot_xpos[l][i] - n[i]*(r + (R)0.5*dist);
    c->dist = dist;
    F(make_frame)(n, c->frame);
    R ff[3] = {p->floor_friction, (R)m->floor_friction[1], (R)m->floor_friction[2]};
    /* geom1 = plane, geom2 = sphere */
    float fr1[3] = {0, 0, 0};  /* placeholder, plane friction passed through fr2 max below */
    (void)fr1;
    R footf[3] = {(R)m->foot_friction[0], (R)m->foot_friction[1], (R)m->foot_friction[2]};
    R mx[3] = {ff[0] > footf[0] ? ff[0] : footf[0], ff[1] > footf[1] ? ff[1] : footf[1], ff[2] > footf[2] ? ff[2] : footf[2]};
    float zero3[3] = {0, 0, 0};
    F(mix_params)(zero3, m->floor_solref, m->floor_solimp, m->floor_margin, m->floor_gap, m->floor_solmix,
                  mx, m->foot_solref, m->foot_solimp, m->foot_margin, m->foot_gap, m->foot_solmix, c);
    c->geom1 = -1; c->geom2 = l; c->foot = l; c->box = -1;
  }
  /* group 2: sphere-box */
  for (int k = 4; k < ONCON; k++) {
    F(OContact)* c = &d->contact[k];
    c->dist = 1; c->foot = -1; c->box = -2; c->geom1 = -2; c->geom2 = -2;
    for (int i = 0; i < 3; i++) c->pos[i] = 0;
    for (int i = 0; i < 9; i++) c->frame[i] = 0;
    for (int i = 0; i < 5; i++) { c->friction[i] = 0; c->solimp[i] = 0; }
    c->solref[0] = c->solref[1] = 0; c->includemargin = 0;
  }
  if (nbox <= 0) return;
  int npair = 4*nbox;
  int maxp = m->max_geom_pairs, maxc = m->max_contact_points;
  int sel[4*PGTT_MAX_BOX]; int nsel = 0;
  if (maxp > -1 && npair > maxp) {
    extern int g_oracle_fresh_rbound;
    nsel = F(broad_phase)(m, d, boxes, nbox, maxp, g_oracle_fresh_rbound, sel);
    {  /* diagnostic (tools/gpu_model_switch_relevance.py): would the other reading of rbound pick another set of pairs? */
      int alt[4*PGTT_MAX_BOX]; unsigned char in[4*PGTT_MAX_BOX];
      int na = F(broad_phase)(m, d, boxes, nbox, maxp, !g_oracle_fresh_rbound, alt);
      for (int i = 0; i < npair; i++) in[i] = 0;
      for (int i = 0; i < nsel; i++) in[sel[i]] = 1;
      for (int i = 0; i < na; i++) if (!in[alt[i]]) d->diag_flags |= 1;
      d->diag_flags |= 4;
    }
  } else {
    for (int i = 0; i < npair; i++) sel[nsel++] = i;
  }
  R ndist[4*PGTT_MAX_BOX], npos[4*PGTT_MAX_BOX][3], nn[4*PGTT_MAX_BOX][3];
  for (int s = 0; s < nsel; s++) {
    int l = sel[s] / nbox, b = sel[s] % nbox;
    const float* bx = boxes + 10*b;
    R bpos[3] = {(R)bx[0], (R)bx[1], (R)bx[2]}, bq[4] = {(R)bx[3], (R)bx[4], (R)bx[5], (R)bx[6]};
    R size[3] = {(R)bx[7], (R)bx[8], (R)bx[9]}, bmat[9];
    F(quat_to_mat)(bq, bmat);
    F(sphere_box)(d->foot_xpos[l], (R)m->foot_radius[l], bpos, bmat, size, &ndist[s], npos[s], nn[s]);
  }
  int keep[4*PGTT_MAX_BOX]; int nkeep = 0;
  if (maxc > -1 && nsel > maxc) {
    unsigned char used[4*PGTT_MAX_BOX];
    for (int s = 0; s < nsel; s++) used[s] = 0;
    for (int k = 0; k < maxc; k++) {
      int bi = -1;
      for (int s = 0; s < nsel; s++) if (!used[s] && (bi < 0 || ndist[s] < ndist[bi])) bi = s;
      used[bi] = 1; keep[nkeep++] = bi;
    }
  } else {
    for (int s = 0; s < nsel; s++) keep[nkeep++] = s;
  }
  for (int k = 0; k < nkeep && k < 4; k++) {
    int s = keep[k], l = sel[s] / nbox, b = sel[s] % nbox;
    F(OContact)* c = &d->contact[4 + k];
    c->dist = ndist[s];
    for (int i = 0; i < 3; i++) c->pos[i] = npos[s][i];
    F(make_frame)(nn[s], c->frame);
    R bf[3] = {box_friction ? (R)box_friction[b] : (R)m->box_friction[0], (R)m->box_friction[1], (R)m->box_friction[2]};
    R footf[3] = {(R)m->foot_friction[0], (R)m->foot_friction[1], (R)m->foot_friction[2]};
    R mx[3] = {bf[0] > footf[0] ? bf[0] : footf[0], bf[1] > footf[1] ? bf[1] : footf[1], bf[2] > footf[2] ? bf[2] : footf[2]};
    float zero3[3] = {0, 0, 0};
    F(mix_params)(zero3, m->foot_solref, m->foot_solimp, m->foot_margin, m->foot_gap, m->foot_solmix,
                  mx, m->box_solref, m->box_solimp, m->box_margin, m->box_gap, m->box_solmix, c);
    c->geom1 = l; c->geom2 = b; c->foot = l; c->box = b;
    if (c->dist < -(R)m->foot_radius[l]) d->diag_flags |= 2;      /* an ACTIVE contact with the sphere centre inside the box: where the recalled frame flip would bite */
    if (c->dist < 0 && FABS(c->dist + (R)m->foot_radius[l]) < (R)1e-6) d->diag_flags |= 8;   /* ... ON the surface of the box to 1e-6: the normal is normalize(~0) (the seam attractor) */
  }
}

/* ------------------------------------------------------------------ velocity stage: com_vel, passive, rne, actuation */
static void F(fwd_velocity_actuation)(const PgttModel* m, const F(OParams)* p, F(OData)* d) {
  const R* qv = d->qvel;
  for (int i = 0; i < 6; i++) d->cvel[0][i] = 0;
  {  /* base: free joint */
    R cv[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; i++) for (int k = 0; k < 6; k++) cv[k] += d->cdof[i][k]*qv[i];
    for (int i = 0; i < 3; i++) for (int k = 0; k < 6; k++) d->cdof_dot[i][k] = 0;
    for (int i = 3; i < 6; i++) F(motion_cross)(cv, d->cdof[i], d->cdof_dot[i]);
    for (int i = 3; i < 6; i++) for (int k = 0; k < 6; k++) cv[k] += d->cdof[i][k]*qv[i];
    for (int k = 0; k < 6; k++) d->cvel[1][k] = cv[k];
  }
  for (int b = 2; b < OB; b++) {
    int par = F(body_parent)(b), dof = 4 + b;
    R cv[6];
    for (int k = 0; k < 6; k++) cv[k] = d->cvel[par][k];
    F(motion_cross)(cv, d->cdof[dof], d->cdof_dot[dof]);
    for (int k = 0; k < 6; k++) d->cvel[b][k] = cv[k] + d->cdof[dof][k]*qv[dof];
  }
  /* passive */
  for (int i = 0; i < 6; i++) d->qfrc_passive[i] = -(R)m->dof_damping[i]*qv[i];
  for (int j = 0; j < 12; j++) d->qfrc_passive[6 + j] = -p->damping[j]*qv[6 + j];
  /* rne */
  R cacc[OB][6], cfrc[OB][6];
  for (int k = 0; k < 3; k++) { cacc[0][k] = 0; cacc[0][3 + k] = -(R)m->gravity[k]; }
  for (int b = 1; b < OB; b++) {
    int par = F(body_parent)(b);
    for (int k = 0; k < 6; k++) cacc[b][k] = cacc[par][k];
    if (b == 1) { for (int i = 0; i < 6; i++) for (int k = 0; k < 6; k++) cacc[b][k] += d->cdof_dot[i][k]*qv[i]; }
    else { int dof = 4 + b; for (int k = 0; k < 6; k++) cacc[b][k] += d->cdof_dot[dof][k]*qv[dof]; }
  }
  for (int b = 0; b < OB; b++) {
    R f1[6], f2[6], f3[6];
    F(inert_mul)(d->cinert[b], cacc[b], f1);
    F(inert_mul)(d->cinert[b], d->cvel[b], f2);
    F(motion_cross_force)(d->cvel[b], f2, f3);
    for (int k = 0; k < 6; k++) cfrc[b][k] = f1[k] + f3[k];
  }
  for (int l = 0; l < 4; l++) for (int k = 2; k >= 1; k--) { int b = 2 + 3*l + k; for (int i = 0; i < 6; i++) cfrc[b - 1][i] += cfrc[b][i]; }
  {
    R carry[6];
    for (int i = 0; i < 6; i++) carry[i] = cfrc[2][i];
    for (int l = 1; l < 4; l++) for (int i = 0; i < 6; i++) carry[i] += cfrc[2 + 3*l][i];
    for (int i = 0; i < 6; i++) cfrc[1][i] += carry[i];
  }
  for (int i = 0; i < ONV; i++) {
    R s = 0; const R* f = cfrc[F(dof_body)(i)];
    for (int k = 0; k < 6; k++) s += d->cdof[i][k]*f[k];
    d->qfrc_bias[i] = s;
  }
  /* actuation (mjx.forward.fwd_actuation): position actuators, affine bias, ctrl + force clamping */
  for (int i = 0; i < ONV; i++) d->qfrc_actuator[i] = 0;
  for (int a = 0; a < 12; a++) {
    int dof = m->act_dof[a];
    R ctrl = d->ctrl[a];
    R lo = (R)m->act_ctrlrange[a][0], hi = (R)m->act_ctrlrange[a][1];
    ctrl = ctrl < lo ? lo : (ctrl > hi ? hi : ctrl);
    R length = d->qpos[7 + (dof - 6)], velocity = qv[dof];
    R force = p->gain[a]*ctrl + ((R)m->act_bias[a][0] + p->bias1[a]*length + (R)m->act_bias[a][2]*velocity);
    R flo = (R)m->act_forcerange[a][0], fhi = (R)m->act_forcerange[a][1];
    force = force < flo ? flo : (force > fhi ? fhi : force);
    d->actuator_force[a] = force;
    d->qfrc_actuator[dof] += force;
  }
  for (int i = 0; i < ONV; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
  F(cho_solve)(d->qLD, d->qfrc_smooth, d->qacc_smooth);
}

/* ------------------------------------------------------------------ make_constraint */
static void F(kbi)(const PgttModel* m, const R* solref, const R* solimp, R pos, R* k_out, R* b_out, R* imp_out) {
  R timeconst = solref[0], dampratio = solref[1];
  R ts2 = 2*(R)m->timestep;
  if (timeconst < ts2) timeconst = ts2;     /* refsafe */
  R dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  dmin = dmin < (R)MJ_MINIMP ? (R)MJ_MINIMP : (dmin > (R)MJ_MAXIMP ? (R)MJ_MAXIMP : dmin);
  dmax = dmax < (R)MJ_MINIMP ? (R)MJ_MINIMP : (dmax > (R)MJ_MAXIMP ? (R)MJ_MAXIMP : dmax);
  width = width > (R)MJ_MINVAL ? width : (R)MJ_MINVAL;
  mid = mid < (R)MJ_MINIMP ? (R)MJ_MINIMP : (mid > (R)MJ_MAXIMP ? (R)MJ_MAXIMP : mid);
  power = power > 1 ? power : 1;
  R k = 1 / (dmax*dmax*timeconst*timeconst*dampratio*dampratio);
  R b = 2 / (dmax*timeconst);
  if (solref[0] <= 0) k = -solref[0] / (dmax*dmax);
  if (solref[1] <= 0) b = -solref[1] / dmax;
  R imp_x = FABS(pos) / width;
  R imp_a = (1 / POW(mid, power - 1))*POW(imp_x, power);
  R imp_b = 1 - (1 / POW(1 - mid, power - 1))*POW(1 - imp_x, power);
  R imp_y = imp_x < mid ? imp_a : imp_b;
  R imp = dmin + imp_y*(dmax - dmin);
  imp = imp < dmin ? dmin : (imp > dmax ? dmax : imp);
  if (imp_x > 1) imp = dmax;
  *k_out = k; *b_out = b; *imp_out = imp;
}

static void F(efc_row)(const PgttModel* m, F(OData)* d, int row, const R* j, R pos, R invweight,
                       const R* solref, const R* solimp, int active) {
  R k, b, imp;
  F(kbi)(m, solref, solimp, pos, &k, &b, &imp);
  R r = invweight*(1 - imp)/imp;
  if (r < (R)MJ_MINVAL) r = (R)MJ_MINVAL;
  R jv = 0;
  for (int i = 0; i < ONV; i++) jv += j[i]*d->qvel[i];
  R aref = -b*jv - k*imp*pos;
  d->efc_active_row[row] = active;
  for (int i = 0; i < ONV; i++) d->efc_J[row][i] = active ? j[i] : 0;
  d->efc_pos[row] = active ? pos : 0;
  d->efc_D[row] = active ? 1 / r : 0;
  d->efc_aref[row] = active ? aref : 0;
}

static void F(make_constraint)(const PgttModel* m, F(OData)* d) {
  int row = 0;
  R jsr[2] = {(R)m->jnt_solref[0], (R)m->jnt_solref[1]};
  R jsi[5]; for (int i = 0; i < 5; i++) jsi[i] = (R)m->jnt_solimp[i];
  for (int j = 0; j < 12; j++, row++) {
    R q = d->qpos[7 + j];
    R dmin = q - (R)m->jnt_range[j][0], dmax = (R)m->jnt_range[j][1] - q;
    R pos = dmin < dmax ? dmin : dmax;
    int active = pos < 0;
    R jr[ONV]; for (int i = 0; i < ONV; i++) jr[i] = 0;
    jr[6 + j] = (R)((dmin < dmax)*2 - 1);
    F(efc_row)(m, d, row, jr, pos, (R)m->dof_invweight0[6 + j], jsr, jsi, active);
  }
  for (int c = 0; c < ONCON; c++) {
    const F(OContact)* con = &d->contact[c];
    R dist = con->dist - con->includemargin;
    int active = (con->box != -2) && (dist < 0);
    /* jac_dif_pair = jacp(body2) - jacp(body1) at con->pos; static bodies have zero Jacobian */
    R sign = (con->box == -1) ? (R)1 : (R)-1;        /* plane contacts: body2 = calf; box contacts: body1 = calf */
    R diff[ONV][3];
    for (int i = 0; i < ONV; i++) diff[i][0] = diff[i][1] = diff[i][2] = 0;
    R t = 0;
    if (con->foot >= 0) {
      int calf = 4 + 3*con->foot;
      R off[3] = {con->pos[0] - d->com[0], con->pos[1] - d->com[1], con->pos[2] - d->com[2]};
      for (int i = 0; i < ONV; i++) {
        int on_chain = (i < 6) || (i >= 6 + 3*con->foot && i < 9 + 3*con->foot);
        if (!on_chain) continue;
        R cr[3]; F(cross3)(d->cdof[i], off, cr);
        for (int k = 0; k < 3; k++) diff[i][k] = sign*(d->cdof[i][3 + k] + cr[k]);
      }
      t = (R)m->body_invweight0[calf - 1][0];        /* + 0 for the static body */
    }
    R dc[3][ONV];
    for (int a = 0; a < 3; a++) for (int i = 0; i < ONV; i++)
      dc[a][i] = con->frame[3*a]*diff[i][0] + con->frame[3*a + 1]*diff[i][1] + con->frame[3*a + 2]*diff[i][2];
    for (int a = 1; a <= 2; a++) {
      R fr = con->friction[a - 1];
      for (int s = 0; s < 2; s++, row++) {
        R f = s == 0 ? fr : -fr;
        R j[ONV];
        for (int i = 0; i < ONV; i++) j[i] = dc[0][i] + dc[a][i]*f;
        R invweight = (t + f*f*t)*2*f*f / (R)m->impratio;
        F(efc_row)(m, d, row, j, dist, invweight, con->solref, con->solimp, active);
      }
    }
  }
}

/* ------------------------------------------------------------------ solver (mjx solver.py, Newton, dense, pyramidal) */
typedef struct F(OCtx) {
  R qacc[ONV], Jaref[ONEFC], Ma[ONV], grad[ONV], Mgrad[ONV], search[ONV];
  R efc_force[ONEFC], qfrc_constraint[ONV];
  R gauss, cost, prev_cost;
  int active[ONEFC];
  int niter;
} F(OCtx);

static void F(update_constraint)(const F(OData)* d, F(OCtx)* c) {
  for (int r = 0; r < ONEFC; r++) {
    c->active[r] = c->Jaref[r] < 0;
    c->efc_force[r] = d->efc_D[r]*(-c->Jaref[r])*(R)c->active[r];
  }
  for (int i = 0; i < ONV; i++) { R s = 0; for (int r = 0; r < ONEFC; r++) s += d->efc_J[r][i]*c->efc_force[r]; c->qfrc_constraint[i] = s; }
  R g = 0;
  for (int i = 0; i < ONV; i++) g += (c->Ma[i] - d->qfrc_smooth[i])*(c->qacc[i] - d->qacc_smooth[i]);
  c->gauss = (R)0.5*g;
  R s = 0;
  for (int r = 0; r < ONEFC; r++) s += d->efc_D[r]*c->Jaref[r]*c->Jaref[r]*(R)c->active[r];
  c->prev_cost = c->cost;
  c->cost = (R)0.5*s + c->gauss;
}
static void F(update_gradient)(const F(OData)* d, F(OCtx)* c) {
  for (int i = 0; i < ONV; i++) c->grad[i] = c->Ma[i] - d->qfrc_smooth[i] - c->qfrc_constraint[i];
  R Hl[ONV][ONV], Ll[ONV][ONV];
  for (int i = 0; i < ONV; i++) for (int j = 0; j < ONV; j++) {
    R s = 0;
    for (int r = 0; r < ONEFC; r++) s += (d->efc_J[r][i]*d->efc_D[r]*(R)c->active[r])*d->efc_J[r][j];
    Hl[i][j] = d->qM[i][j] + s;
  }
  F(cholesky)(Hl, Ll);
  F(cho_solve)(Ll, c->grad, c->Mgrad);
}
static void F(ctx_create)(const F(OData)* d, const R* qacc, int grad, F(OCtx)* c) {
  for (int i = 0; i < ONV; i++) c->qacc[i] = qacc[i];
  for (int r = 0; r < ONEFC; r++) { R s = 0; for (int i = 0; i < ONV; i++) s += d->efc_J[r][i]*qacc[i]; c->Jaref[r] = s - d->efc_aref[r]; }
  for (int i = 0; i < ONV; i++) { R s = 0; for (int j = 0; j < ONV; j++) s += d->qM[i][j]*qacc[j]; c->Ma[i] = s; }
  for (int i = 0; i < ONV; i++) c->grad[i] = c->Mgrad[i] = c->search[i] = 0;
  c->gauss = 0; c->cost = (R)INFINITY; c->prev_cost = 0; c->niter = 0;
  F(update_constraint)(d, c);
  if (grad) {
    F(update_gradient)(d, c);
    for (int i = 0; i < ONV; i++) c->search[i] = -c->Mgrad[i];
  }
}

typedef struct F(OLSPoint) { R alpha, cost, deriv0, deriv1; } F(OLSPoint);
static F(OLSPoint) F(ls_point)(const F(OCtx)* c, R alpha, const R* jv, const R (*quad)[3], const R* quad_gauss) {
  R qt[3] = {quad_gauss[0], quad_gauss[1], quad_gauss[2]};
  R s0 = 0, s1 = 0, s2 = 0;
  for (int r = 0; r < ONEFC; r++) {
    R x = c->Jaref[r] + alpha*jv[r];
    R act = (x < 0) ? (R)1 : (R)0;
    s0 += quad[r][0]*act; s1 += quad[r][1]*act; s2 += quad[r][2]*act;
  }
  qt[0] += s0; qt[1] += s1; qt[2] += s2;
  F(OLSPoint) p;
  p.alpha = alpha;
  p.cost = alpha*alpha*qt[2] + alpha*qt[1] + qt[0];
  p.deriv0 = 2*alpha*qt[2] + qt[1];
  p.deriv1 = 2*qt[2] + (qt[2] == 0 ? (R)MJ_MINVAL : (R)0);
  return p;
}
static inline int F(in_bracket)(const F(OLSPoint)* x, const F(OLSPoint)* y) {
  return ((x->deriv0 < y->deriv0) && (y->deriv0 < 0)) || ((x->deriv0 > y->deriv0) && (y->deriv0 > 0));
}
static void F(linesearch)(const PgttModel* m, const F(OData)* d, F(OCtx)* c) {
  extern double* g_oracle_trace; extern int g_oracle_trace_n;
  R snorm = 0; for (int i = 0; i < ONV; i++) snorm += c->search[i]*c->search[i];
  R smag = SQRT(snorm)*(R)m->meaninertia*(R)(ONV > 1 ? ONV : 1);
  R gtol = (R)m->tolerance*(R)m->ls_tolerance*smag;
  R mv[ONV], jv[ONEFC];
  for (int i = 0; i < ONV; i++) { R s = 0; for (int j = 0; j < ONV; j++) s += d->qM[i][j]*c->search[j]; mv[i] = s; }
  for (int r = 0; r < ONEFC; r++) { R s = 0; for (int i = 0; i < ONV; i++) s += d->efc_J[r][i]*c->search[i]; jv[r] = s; }
  R a = 0, b = 0, e = 0;
  for (int i = 0; i < ONV; i++) { a += c->search[i]*c->Ma[i]; b += c->search[i]*d->qfrc_smooth[i]; e += c->search[i]*mv[i]; }
  R quad_gauss[3] = {c->gauss, a - b, (R)0.5*e};
  R quad[ONEFC][3];
  for (int r = 0; r < ONEFC; r++) {
    quad[r][0] = ((R)0.5*c->Jaref[r]*c->Jaref[r])*d->efc_D[r];
    quad[r][1] = (jv[r]*c->Jaref[r])*d->efc_D[r];
    quad[r][2] = ((R)0.5*jv[r]*jv[r])*d->efc_D[r];
  }
  F(OLSPoint) p0 = F(ls_point)(c, 0, jv, quad, quad_gauss);
  F(OLSPoint) lo0 = F(ls_point)(c, p0.alpha - p0.deriv0/p0.deriv1, jv, quad, quad_gauss);
  int lesser = lo0.deriv0 < p0.deriv0;
  F(OLSPoint) hi = lesser ? p0 : lo0;
  F(OLSPoint) lo = lesser ? lo0 : p0;
  int swap = 1, it = 0;
  for (;;) {
    int done = it >= m->ls_iterations;
    done |= !swap;
    done |= (lo.deriv0 < 0) && (lo.deriv0 > -gtol);
    done |= (hi.deriv0 > 0) && (hi.deriv0 < gtol);
    if (done) break;
    F(OLSPoint) lo_next = F(ls_point)(c, lo.alpha - lo.deriv0/lo.deriv1, jv, quad, quad_gauss);
    F(OLSPoint) hi_next = F(ls_point)(c, hi.alpha - hi.deriv0/hi.deriv1, jv, quad, quad_gauss);
    F(OLSPoint) mid = F(ls_point)(c, (R)0.5*(lo.alpha + hi.alpha), jv, quad, quad_gauss);
    int s1 = F(in_bracket)(&lo, &lo_next); if (s1) lo = lo_next;
    int s2 = F(in_bracket)(&lo, &mid);     if (s2) lo = mid;
    int s3 = F(in_bracket)(&lo, &hi_next); if (s3) lo = hi_next;
    int t1 = F(in_bracket)(&hi, &hi_next); if (t1) hi = hi_next;
    int t2 = F(in_bracket)(&hi, &mid);     if (t2) hi = mid;
    int t3 = F(in_bracket)(&hi, &lo_next); if (t3) hi = lo_next;
    swap = s1 | s2 | s3 | t1 | t2 | t3;
    it++;
  }
  int improved = (lo.cost < p0.cost) || (hi.cost < p0.cost);
  R alpha = lo.cost < hi.cost ? lo.alpha : hi.alpha;
  R ia = improved ? alpha : 0;
  if (g_oracle_trace && g_oracle_trace_n < 4000) { g_oracle_trace[g_oracle_trace_n++] = (double)ia; g_oracle_trace[g_oracle_trace_n++] = (double)it; g_oracle_trace[g_oracle_trace_n++] = (double)(lo.cost < hi.cost ? lo.deriv0 : hi.deriv0); }
  for (int i = 0; i < ONV; i++) { c->qacc[i] += c->search[i]*ia; c->Ma[i] += mv[i]*ia; }
  for (int r = 0; r < ONEFC; r++) c->Jaref[r] += jv[r]*ia;
}

static void F(solve)(const PgttModel* m, F(OData)* d) {
  extern double* g_oracle_trace; extern int g_oracle_trace_n;
  F(OCtx) warm, smth, ctx;
  F(ctx_create)(d, d->qacc_warmstart, 0, &warm);
  F(ctx_create)(d, d->qacc_smooth, 0, &smth);
  const R* q0 = warm.cost < smth.cost ? d->qacc_warmstart : d->qacc_smooth;
  F(ctx_create)(d, q0, 1, &ctx);
  R scale = (R)m->meaninertia*(R)(ONV > 1 ? ONV : 1);
  for (;;) {
    R improvement = (ctx.prev_cost - ctx.cost)/scale;
    R gn = 0; for (int i = 0; i < ONV; i++) gn += ctx.grad[i]*ctx.grad[i];
    R gradient = SQRT(gn)/scale;
    if (g_oracle_trace && g_oracle_trace_n < 4000) { g_oracle_trace[g_oracle_trace_n++] = 1000 + ctx.niter; g_oracle_trace[g_oracle_trace_n++] = (double)ctx.cost; g_oracle_trace[g_oracle_trace_n++] = (double)gradient; }
    int done = ctx.niter >= m->iterations;
    done |= improvement < (R)m->tolerance;
    done |= gradient < (R)m->tolerance;
    if (done) break;
    F(linesearch)(m, d, &ctx);
    F(update_constraint)(d, &ctx);
    F(update_gradient)(d, &ctx);
    for (int i = 0; i < ONV; i++) ctx.search[i] = -ctx.Mgrad[i];
    ctx.niter++;
  }
  for (int i = 0; i < ONV; i++) { d->qacc[i] = ctx.qacc[i]; d->qacc_warmstart[i] = ctx.qacc[i]; d->qfrc_constraint[i] = ctx.qfrc_constraint[i]; }
  for (int r = 0; r < ONEFC; r++) d->efc_force[r] = ctx.efc_force[r];
  d->solver_niter = ctx.niter;
  { R gn = 0; for (int i = 0; i < ONV; i++) gn += ctx.grad[i]*ctx.grad[i]; d->solver_resid = SQRT(gn)/scale; }
}

/* ------------------------------------------------------------------ sensors (mjx sensor.py) */
static void F(sensors)(const PgttModel* m, F(OData)* d) {
  (void)m;
  R* s = d->sensordata;
  const R* rot = d->site_imu_mat;
  const R* cv = d->cvel[1];
  R dif[3] = {d->site_imu[0] - d->com[0], d->site_imu[1] - d->com[1], d->site_imu[2] - d->com[2]};
  /* gyro */
  F(matT_mul_vec)(rot, cv, s + 0);
  /* accelerometer: needs cacc of the base from rne_postconstraint */
  R cacc[6] = {0, 0, 0, -(R)m->gravity[0], -(R)m->gravity[1], -(R)m->gravity[2]};
  for (int i = 0; i < 6; i++) for (int k = 0; k < 6; k++) cacc[k] += d->cdof_dot[i][k]*d->qvel[i] + d->cdof[i][k]*d->qacc[i];
  for (int k = 0; k < 6; k++) d->cacc_base[k] = cacc[k];
  {
    R ang[3], lin[3], acc[3], t[3], c1[3], c2[3], corr[3];
    F(matT_mul_vec)(rot, cv, ang);
    F(cross3)(dif, cv, c1);
    t[0] = cv[3] - c1[0]; t[1] = cv[4] - c1[1]; t[2] = cv[5] - c1[2];
    F(matT_mul_vec)(rot, t, lin);
    F(cross3)(dif, cacc, c2);
    t[0] = cacc[3] - c2[0]; t[1] = cacc[4] - c2[1]; t[2] = cacc[5] - c2[2];
    F(matT_mul_vec)(rot, t, acc);
    F(cross3)(ang, lin, corr);
    for (int i = 0; i < 3; i++) s[3 + i] = acc[i] + corr[i];
  }
  /* framequat (site imu) */
  { R one[4] = {1, 0, 0, 0}; F(quat_mul)(d->xquat[1], one, s + 6); }
  /* framepos global_position */
  for (int i = 0; i < 3; i++) s[10 + i] = d->site_imu[i];
  /* framelinvel / frameangvel (world frame) */
  { R c1[3]; F(cross3)(dif, cv, c1); for (int i = 0; i < 3; i++) { s[13 + i] = cv[3 + i] - c1[i]; s[16 + i] = cv[i]; }
    /* velocimeter */
    R t[3] = {s[13], s[14], s[15]}; F(matT_mul_vec)(rot, t, s + 19); }
  /* framezaxis */
  s[22] = rot[2]; s[23] = rot[5]; s[24] = rot[8];
  /* feet: sensor order FR,FL,RR,RL = legs 1,0,3,2 */
  static const int leg_of_foot[4] = {1, 0, 3, 2};
  for (int f = 0; f < 4; f++) {
    int l = leg_of_foot[f], b = 4 + 3*l;
    R rel[3] = {d->site_foot[l][0] - d->site_imu[0], d->site_foot[l][1] - d->site_imu[1], d->site_foot[l][2] - d->site_imu[2]};
    F(matT_mul_vec)(rot, rel, s + 25 + 3*f);
    R off[3] = {d->site_foot[l][0] - d->com[0], d->site_foot[l][1] - d->com[1], d->site_foot[l][2] - d->com[2]}, c1[3];
    F(cross3)(off, d->cvel[b], c1);
    for (int i = 0; i < 3; i++) s[37 + 3*f + i] = d->cvel[b][3 + i] - c1[i];
  }
}

/* ------------------------------------------------------------------ mjx.forward / mjx.step */
static void F(forward)(const PgttModel* m, const F(OParams)* p, const float* boxes, const float* box_friction, int nbox, F(OData)* d) {
  F(kinematics)(m, p, d);
  F(com_pos)(m, p, d);
  F(crb)(m, p, d);
  F(collision)(m, p, boxes, box_friction, nbox, d);
  F(fwd_velocity_actuation)(m, p, d);
  F(make_constraint)(m, d);
  F(solve)(m, d);
  F(sensors)(m, d);
}

static void F(euler)(const PgttModel* m, F(OData)* d) {
  R dt = (R)m->timestep;
  for (int i = 0; i < ONV; i++) d->qvel[i] = d->qvel[i] + d->qacc[i]*dt;
  for (int i = 0; i < 3; i++) d->qpos[i] = d->qpos[i] + dt*d->qvel[i];
  {
    R v[3] = {d->qvel[3], d->qvel[4], d->qvel[5]};
    R nn = F(normalize_n)(v, 3);
    R angle = dt*nn, qr[4], q2[4];
    F(axis_angle_to_quat)(v, angle, qr);
    F(quat_mul)(d->qpos + 3, qr, q2);
    F(normalize_n)(q2, 4);
    for (int i = 0; i < 4; i++) d->qpos[3 + i] = q2[i];
  }
  for (int j = 0; j < 12; j++) d->qpos[7 + j] = d->qpos[7 + j] + dt*d->qvel[6 + j];
}

/* mjx_env.step(model, data, action, n_substeps): scan of { ctrl <- action ; mjx.step } */
static void F(env_step)(const PgttModel* m, const F(OParams)* p, const float* boxes, const float* box_friction, int nbox,
                        F(OData)* d, const R* ctrl, int nsub) {
  d->solver_niter_max = 0; d->solver_resid_max = 0; d->diag_flags = 0;
  for (int s = 0; s < nsub; s++) {
    for (int a = 0; a < 12; a++) d->ctrl[a] = ctrl[a];
    F(forward)(m, p, boxes, box_friction, nbox, d);
    if (d->solver_niter > d->solver_niter_max) d->solver_niter_max = d->solver_niter;
    if (d->solver_resid > d->solver_resid_max) d->solver_resid_max = d->solver_resid;
    F(euler)(m, d);
  }
}

static void F(params_nominal)(const PgttModel* m, F(OParams)* p) {
  for (int b = 0; b < 13; b++) p->body_mass[b] = (R)m->body_mass[b];
  for (int i = 0; i < 3; i++) p->base_ipos[i] = (R)m->body_ipos[0][i];
  for (int j = 0; j < 12; j++) {
    p->qpos0j[j] = (R)m->qpos0[7 + j];
    p->armature[j] = (R)m->dof_armature[6 + j];
    p->damping[j] = (R)m->dof_damping[6 + j];
    p->gain[j] = (R)m->act_gain[j];
    p->bias1[j] = (R)m->act_bias[j][1];
  }
  p->floor_friction = (R)m->floor_friction[0];
}
