/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's joystick_pgtt hot path
 * (see physics_impl.h / task_impl.h for the per-function reference citations and the PARITY UNPINNED
 * statement for the un-vendored MJX physics).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load liboracle.so; the product (libpgtt.so) never does.
 *
 * Build: make -C oracle   (gcc, -O2, -ffp-contract=off so the f32 build is plain IEEE fp32)
 */
/* per-iteration solver trace for single-env replays (tools only): records of 3 doubles, (1000 + iter, cost, |grad|/scale) at the top of
   each Newton trip and (alpha, ls rounds, d0 at the accepted point) after each line search */
double* g_oracle_trace = 0; int g_oracle_trace_n = 0;
void pgtt_oracle_set_trace(double* buf4000_or_null) { g_oracle_trace = buf4000_or_null; g_oracle_trace_n = 0; }
int pgtt_oracle_trace_len(void) { return g_oracle_trace_n; }
/* diagnostic switch (tools/gpu_model_switch_relevance.py only): rank the max_geom_pairs cut with every box's own bounding radius instead of the stale compiled one */
int g_oracle_fresh_rbound = 0;
void pgtt_oracle_set_fresh_rbound(int on) { g_oracle_fresh_rbound = on; }
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/pgtt.h"

/* ---------------------------------------------------------------- Philox4x32-10 (Salmon et al. 2011) */
static inline void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c[4]) {
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
static int g_rng_override = 0;   /* tests only: every uniform draw returns g_rng_value (fixtures generated with stubbed jax.random) */
static float g_rng_value = 0.5f;
void pgtt_oracle_set_rng_override(int on) { g_rng_override = on; g_rng_value = 0.5f; }
void pgtt_oracle_set_rng_override_value(float v) { g_rng_override = 1; g_rng_value = v; }
static inline float pgtt_philox_uniform(uint64_t seed, uint32_t env, uint32_t epoch, uint32_t stream, int idx) {
  if (g_rng_override) return g_rng_value;
  uint32_t c[4] = {env, epoch, stream, (uint32_t)(idx >> 2)};
  philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), c);
  return (float)(c[idx & 3] >> 8) * (1.0f / 16777216.0f);
}
float pgtt_oracle_uniform(uint64_t seed, uint32_t env, uint32_t epoch, uint32_t stream, int idx) {
  return pgtt_philox_uniform(seed, env, epoch, stream, idx);
}

#define PGTT_CAT2(a, b) a##b
#define PGTT_CAT(a, b) PGTT_CAT2(a, b)

/* ---- float32 instantiation */
#define R float
#define F(name) PGTT_CAT(name, _f32)
#define SQRT sqrtf
#define FABS fabsf
#define SIN sinf
#define COS cosf
#define POW powf
#define EXP expf
#define ATAN2 atan2f
#define FMOD fmodf
#include "physics_impl.h"
#include "task_impl.h"
#undef R
#undef F
#undef SQRT
#undef FABS
#undef SIN
#undef COS
#undef POW
#undef EXP
#undef ATAN2
#undef FMOD
#undef OB
#undef ONV
#undef ONEFC
#undef ONCON
#undef MJ_MINVAL
#undef MJ_MINIMP
#undef MJ_MAXIMP

/* ---- float64 instantiation */
#define R double
#define F(name) PGTT_CAT(name, _f64)
#define SQRT sqrt
#define FABS fabs
#define SIN sin
#define COS cos
#define POW pow
#define EXP exp
#define ATAN2 atan2
#define FMOD fmod
#include "physics_impl.h"
#include "task_impl.h"

/* ================================================================ exported debug entry: one forward (+ euler) */
typedef struct PgttOracleDump {
  double qpos_in_normalized[19];
  double xpos[14*3], xquat[14*4], xipos[14*3], com[3], cdof[18*6], cinert[14*10];
  double qM[18*18], qfrc_bias[18], qfrc_passive[18], qfrc_actuator[18], actuator_force[12];
  double qfrc_smooth[18], qacc_smooth[18];
  double con_dist[8], con_pos[8*3], con_frame[8*9], con_friction[8], con_solimp[8*5], con_solref[8*2];
  int32_t con_foot[8], con_box[8];
  double efc_J[44*18], efc_D[44], efc_aref[44], efc_pos[44], efc_force[44];
  int32_t efc_active[44];
  double qacc[18], qfrc_constraint[18];
  int32_t niter;
  double sensordata[49], site_imu_mat[9], site_foot[12], foot_xpos[12];
  double qpos_next[19], qvel_next[18];
} PgttOracleDump;

int pgtt_oracle_sizeof_dump(void) { return (int)sizeof(PgttOracleDump); }

#define DUMP_BODY(SUF, RT)                                                                               \
  static void dump_##SUF(const OData_##SUF* d, PgttOracleDump* o) {                                      \
    for (int i = 0; i < 19; i++) o->qpos_in_normalized[i] = d->qpos[i];                                  \
    for (int b = 0; b < 14; b++) {                                                                       \
      for (int i = 0; i < 3; i++) { o->xpos[3*b + i] = d->xpos[b][i]; o->xipos[3*b + i] = d->xipos[b][i]; } \
      for (int i = 0; i < 4; i++) o->xquat[4*b + i] = d->xquat[b][i];                                    \
      for (int i = 0; i < 10; i++) o->cinert[10*b + i] = d->cinert[b][i];                                \
    }                                                                                                    \
    for (int i = 0; i < 3; i++) o->com[i] = d->com[i];                                                   \
    for (int i = 0; i < 18; i++) {                                                                       \
      for (int k = 0; k < 6; k++) o->cdof[6*i + k] = d->cdof[i][k];                                      \
      for (int j = 0; j < 18; j++) o->qM[18*i + j] = d->qM[i][j];                                        \
      o->qfrc_bias[i] = d->qfrc_bias[i]; o->qfrc_passive[i] = d->qfrc_passive[i];                        \
      o->qfrc_actuator[i] = d->qfrc_actuator[i]; o->qfrc_smooth[i] = d->qfrc_smooth[i];                  \
      o->qacc_smooth[i] = d->qacc_smooth[i]; o->qacc[i] = d->qacc[i];                                    \
      o->qfrc_constraint[i] = d->qfrc_constraint[i];                                                     \
    }                                                                                                    \
    for (int a = 0; a < 12; a++) o->actuator_force[a] = d->actuator_force[a];                            \
    for (int c = 0; c < 8; c++) {                                                                        \
      o->con_dist[c] = d->contact[c].dist; o->con_foot[c] = d->contact[c].foot; o->con_box[c] = d->contact[c].box; \
      o->con_friction[c] = d->contact[c].friction[0];                                                    \
      for (int i = 0; i < 3; i++) o->con_pos[3*c + i] = d->contact[c].pos[i];                            \
      for (int i = 0; i < 9; i++) o->con_frame[9*c + i] = d->contact[c].frame[i];                        \
      for (int i = 0; i < 5; i++) o->con_solimp[5*c + i] = d->contact[c].solimp[i];                      \
      for (int i = 0; i < 2; i++) o->con_solref[2*c + i] = d->contact[c].solref[i];                      \
    }                                                                                                    \
    for (int r = 0; r < 44; r++) {                                                                       \
      for (int i = 0; i < 18; i++) o->efc_J[18*r + i] = d->efc_J[r][i];                                  \
      o->efc_D[r] = d->efc_D[r]; o->efc_aref[r] = d->efc_aref[r]; o->efc_pos[r] = d->efc_pos[r];         \
      o->efc_force[r] = d->efc_force[r]; o->efc_active[r] = d->efc_active_row[r];                        \
    }                                                                                                    \
    o->niter = d->solver_niter;                                                                          \
    for (int i = 0; i < 49; i++) o->sensordata[i] = d->sensordata[i];                                    \
    for (int i = 0; i < 9; i++) o->site_imu_mat[i] = d->site_imu_mat[i];                                 \
    for (int l = 0; l < 4; l++) for (int i = 0; i < 3; i++) {                                            \
      o->site_foot[3*l + i] = d->site_foot[l][i]; o->foot_xpos[3*l + i] = d->foot_xpos[l][i]; }          \
  }
DUMP_BODY(f32, float)
DUMP_BODY(f64, double)

#define PARAMS_FROM(SUF, RT)                                                                             \
  static void params_from_##SUF(const PgttModel* m, const float* prm, long stride, OParams_##SUF* p) {   \
    params_nominal_##SUF(m, p);                                                                          \
    if (!prm) return;                                                                                    \
    for (int b = 0; b < 13; b++) p->body_mass[b] = (RT)prm[(PGTT_P_BODY_MASS + b)*stride];               \
    for (int i = 0; i < 3; i++) p->base_ipos[i] = (RT)prm[(PGTT_P_BASE_IPOS + i)*stride];                \
    for (int j = 0; j < 12; j++) {                                                                       \
      p->qpos0j[j] = (RT)prm[(PGTT_P_QPOS0 + j)*stride];                                                 \
      p->armature[j] = (RT)prm[(PGTT_P_ARMATURE + j)*stride];                                            \
      p->damping[j] = (RT)prm[(PGTT_P_DAMPING + j)*stride];                                              \
      p->gain[j] = (RT)prm[(PGTT_P_GAIN + j)*stride];                                                    \
      p->bias1[j] = (RT)prm[(PGTT_P_BIAS1 + j)*stride];                                                  \
    }                                                                                                    \
    p->floor_friction = (RT)prm[PGTT_P_FLOOR_FRICTION*stride];                                           \
  }
PARAMS_FROM(f32, float)
PARAMS_FROM(f64, double)

/* one mjx.forward (+ mjx euler) from (qpos,qvel,warmstart,ctrl); params: float[PGTT_NPARAM] or NULL */
int pgtt_oracle_forward(const PgttModel* m, const float* params, const float* boxes, const float* box_friction, int nbox,
                        const double* qpos, const double* qvel, const double* warm, const double* ctrl,
                        int fp64, PgttOracleDump* out) {
  if (fp64) {
    OData_f64* d = (OData_f64*)calloc(1, sizeof(OData_f64)); OParams_f64 p; params_from_f64(m, params, 1, &p);
    for (int i = 0; i < 19; i++) d->qpos[i] = qpos[i];
    for (int i = 0; i < 18; i++) { d->qvel[i] = qvel[i]; d->qacc_warmstart[i] = warm[i]; }
    for (int i = 0; i < 12; i++) d->ctrl[i] = ctrl[i];
    forward_f64(m, &p, boxes, box_friction, nbox, d);
    dump_f64(d, out);
    euler_f64(m, d);
    for (int i = 0; i < 19; i++) out->qpos_next[i] = d->qpos[i];
    for (int i = 0; i < 18; i++) out->qvel_next[i] = d->qvel[i];
    free(d);
  } else {
    OData_f32* d = (OData_f32*)calloc(1, sizeof(OData_f32)); OParams_f32 p; params_from_f32(m, params, 1, &p);
    for (int i = 0; i < 19; i++) d->qpos[i] = (float)qpos[i];
    for (int i = 0; i < 18; i++) { d->qvel[i] = (float)qvel[i]; d->qacc_warmstart[i] = (float)warm[i]; }
    for (int i = 0; i < 12; i++) d->ctrl[i] = (float)ctrl[i];
    forward_f32(m, &p, boxes, box_friction, nbox, d);
    dump_f32(d, out);
    euler_f32(m, d);
    for (int i = 0; i < 19; i++) out->qpos_next[i] = d->qpos[i];
    for (int i = 0; i < 18; i++) out->qvel_next[i] = d->qvel[i];
    free(d);
  }
  return 0;
}

/* height scan alone: out[117*3] */
int pgtt_oracle_scan(const PgttConfig* cfg, const float* boxes, int nbox, const double* center, double yaw, int fp64, double* out) {
  if (fp64) {
    double hs[PGTT_NSCAN][3];
    scan_f64(cfg, boxes, nbox, center, yaw, hs);
    for (int i = 0; i < PGTT_NSCAN; i++) for (int k = 0; k < 3; k++) out[3*i + k] = hs[i][k];
  } else {
    float hs[PGTT_NSCAN][3], c[3] = {(float)center[0], (float)center[1], (float)center[2]};
    scan_f32(cfg, boxes, nbox, c, (float)yaw, hs);
    for (int i = 0; i < PGTT_NSCAN; i++) for (int k = 0; k < 3; k++) out[3*i + k] = hs[i][k];
  }
  return 0;
}
double pgtt_oracle_get_z(double phi, double h, double smin, int fp64) {
  return fp64 ? get_z_f64(phi, h, smin) : (double)get_z_f32((float)phi, (float)h, (float)smin);
}
double pgtt_oracle_quat_to_yaw(const double* q, int fp64) {
  if (fp64) return quat_to_yaw_f64(q);
  float qf[4] = {(float)q[0], (float)q[1], (float)q[2], (float)q[3]};
  return quat_to_yaw_f32(qf);
}

/* task layer after the physics, driven with caller-supplied "physics outputs" (fixtures from the reference's own
   step() run with stubbed MJX): info is passed through a 1-env PgttBuffers (host), physics outputs explicitly. */
typedef struct PgttOraclePostIn {
  double qpos[19], qvel[18], sensordata[49], site_imu_mat[9], site_foot_z[4] /* FR,FL,RR,RL */, actuator_force[12];
  double action[12], scan_z[117];
  int32_t contact[4];
} PgttOraclePostIn;

/* diagnostic side channel of the batch step (tests / tools): when set, step e writes the largest scaled gradient norm at the
   solver's exit over its substeps into g_diag_resid[e] */
static double* g_diag_resid = NULL;
void pgtt_oracle_set_diag(double* resid_N_or_null) { g_diag_resid = resid_N_or_null; }
static int32_t* g_diag_flags = NULL;      /* per env: OData.diag_flags of the step (physics_impl.h) */
void pgtt_oracle_set_diag_flags(int32_t* flags_N_or_null) { g_diag_flags = flags_N_or_null; }

/* ================================================================ batch drivers over the PgttBuffers SoA layout (HOST pointers) */
#define BATCH(SUF, RT)                                                                                                   \
  static void gather_##SUF(const PgttBuffers* B, long N, long e, OData_##SUF* d, OInfo_##SUF* in) {                      \
    const float* S = B->state; const int32_t* I = B->istate;                                                             \
    for (int i = 0; i < 19; i++) d->qpos[i] = (RT)S[(PGTT_S_QPOS + i)*N + e];                                            \
    for (int i = 0; i < 18; i++) { d->qvel[i] = (RT)S[(PGTT_S_QVEL + i)*N + e]; d->qacc_warmstart[i] = (RT)S[(PGTT_S_QWARM + i)*N + e]; } \
    for (int i = 0; i < 3; i++) in->command[i] = (RT)S[(PGTT_S_CMD + i)*N + e];                                          \
    for (int i = 0; i < 4; i++) { in->phase[i] = (RT)S[(PGTT_S_PHASE + i)*N + e]; in->air_time[i] = (RT)S[(PGTT_S_AIR_TIME + i)*N + e]; \
      in->swing_peak[i] = (RT)S[(PGTT_S_SWING_PEAK + i)*N + e]; in->H_max[i] = (RT)S[(PGTT_S_HMAX + i)*N + e];           \
      in->H_min[i] = (RT)S[(PGTT_S_HMIN + i)*N + e]; in->last_contact[i] = S[(PGTT_S_LAST_CONTACT + i)*N + e] != 0.0f; } \
    in->phase_dt = (RT)S[PGTT_S_PHASE_DT*N + e]; in->gait_freq = (RT)S[PGTT_S_GAIT_FREQ*N + e];                          \
    for (int i = 0; i < 12; i++) { in->last_act[i] = (RT)S[(PGTT_S_LAST_ACT + i)*N + e];                                 \
      in->last_last_act[i] = (RT)S[(PGTT_S_LAST_LAST_ACT + i)*N + e]; in->motor_targets[i] = (RT)S[(PGTT_S_MOTOR_TARGETS + i)*N + e]; } \
    for (int i = 0; i < 24; i++) { in->qerr_hist[i] = (RT)S[(PGTT_S_QERR_HIST + i)*N + e]; in->qvel_hist[i] = (RT)S[(PGTT_S_QVEL_HIST + i)*N + e]; } \
    for (int i = 0; i < PGTT_NSCAN; i++) in->scan_z[i] = (RT)B->scan_z[e*PGTT_NSCAN + i];                                   \
    in->step = I[PGTT_I_STEP*N + e]; in->steps_until_next_cmd = I[PGTT_I_STEPS_UNTIL_CMD*N + e];                         \
    in->rng_ctr = (uint32_t)I[PGTT_I_RNG_CTR*N + e]; in->ep_steps = I[PGTT_I_EP_STEPS*N + e];                            \
  }                                                                                                                      \
  static void scatter_##SUF(const PgttBuffers* B, long N, long e, const OData_##SUF* d, const OInfo_##SUF* in) {         \
    float* S = B->state; int32_t* I = B->istate;                                                                         \
    for (int i = 0; i < 19; i++) S[(PGTT_S_QPOS + i)*N + e] = (float)d->qpos[i];                                         \
    for (int i = 0; i < 18; i++) { S[(PGTT_S_QVEL + i)*N + e] = (float)d->qvel[i]; S[(PGTT_S_QWARM + i)*N + e] = (float)d->qacc_warmstart[i]; } \
    for (int i = 0; i < 3; i++) S[(PGTT_S_CMD + i)*N + e] = (float)in->command[i];                                       \
    for (int i = 0; i < 4; i++) { S[(PGTT_S_PHASE + i)*N + e] = (float)in->phase[i]; S[(PGTT_S_AIR_TIME + i)*N + e] = (float)in->air_time[i]; \
      S[(PGTT_S_SWING_PEAK + i)*N + e] = (float)in->swing_peak[i]; S[(PGTT_S_HMAX + i)*N + e] = (float)in->H_max[i];     \
      S[(PGTT_S_HMIN + i)*N + e] = (float)in->H_min[i]; S[(PGTT_S_LAST_CONTACT + i)*N + e] = (float)in->last_contact[i]; } \
    S[PGTT_S_PHASE_DT*N + e] = (float)in->phase_dt; S[PGTT_S_GAIT_FREQ*N + e] = (float)in->gait_freq;                    \
    for (int i = 0; i < 12; i++) { S[(PGTT_S_LAST_ACT + i)*N + e] = (float)in->last_act[i];                              \
      S[(PGTT_S_LAST_LAST_ACT + i)*N + e] = (float)in->last_last_act[i]; S[(PGTT_S_MOTOR_TARGETS + i)*N + e] = (float)in->motor_targets[i]; } \
    for (int i = 0; i < 24; i++) { S[(PGTT_S_QERR_HIST + i)*N + e] = (float)in->qerr_hist[i]; S[(PGTT_S_QVEL_HIST + i)*N + e] = (float)in->qvel_hist[i]; } \
    for (int i = 0; i < PGTT_NSCAN; i++) B->scan_z[e*PGTT_NSCAN + i] = (float)in->scan_z[i];                                \
    I[PGTT_I_STEP*N + e] = in->step; I[PGTT_I_STEPS_UNTIL_CMD*N + e] = in->steps_until_next_cmd;                         \
    I[PGTT_I_RNG_CTR*N + e] = (int32_t)in->rng_ctr; I[PGTT_I_EP_STEPS*N + e] = in->ep_steps;                             \
  }                                                                                                                      \
  static void write_frame_##SUF(const PgttBuffers* B, long N, long e, const OData_##SUF* d, const int* contact) {        \
    float* Fm = B->frame; const RT* s = d->sensordata; static const int lof[4] = {1, 0, 3, 2};                           \
    if (!Fm) return;                                                                                                     \
    for (int i = 0; i < 3; i++) { Fm[(PGTT_F_GYRO + i)*N + e] = (float)s[i]; Fm[(PGTT_F_ACCEL + i)*N + e] = (float)s[3 + i]; \
      Fm[(PGTT_F_GLOBAL_LINVEL + i)*N + e] = (float)s[13 + i]; Fm[(PGTT_F_GLOBAL_ANGVEL + i)*N + e] = (float)s[16 + i];   \
      Fm[(PGTT_F_LOCAL_LINVEL + i)*N + e] = (float)s[19 + i]; Fm[(PGTT_F_UPVECTOR + i)*N + e] = (float)s[22 + i];         \
      Fm[(PGTT_F_GRAVITY + i)*N + e] = (float)(-d->site_imu_mat[6 + i]); }                                               \
    for (int i = 0; i < 12; i++) { Fm[(PGTT_F_FEET_POS + i)*N + e] = (float)s[25 + i]; Fm[(PGTT_F_FEET_VEL + i)*N + e] = (float)s[37 + i]; \
      Fm[(PGTT_F_ACT_FORCE + i)*N + e] = (float)d->actuator_force[i]; }                                                  \
    for (int f = 0; f < 4; f++) { Fm[(PGTT_F_CONTACT + f)*N + e] = (float)contact[f];                                    \
      Fm[(PGTT_F_FOOT_SITE_Z + f)*N + e] = (float)d->site_foot[lof[f]][2]; }                                             \
  }                                                                                                                      \
  static void write_dbg_##SUF(const PgttBuffers* B, long e, const OData_##SUF* d) {                                      \
    if (B->dbg_contact) for (int c = 0; c < 8; c++) {                                                                    \
      B->dbg_contact[(e*8 + c)*2] = d->contact[c].foot; B->dbg_contact[(e*8 + c)*2 + 1] = d->contact[c].box; }           \
    if (B->dbg_dist) for (int c = 0; c < 8; c++) B->dbg_dist[e*8 + c] = (float)d->contact[c].dist;                       \
    if (B->dbg_niter) B->dbg_niter[e] = d->solver_niter_max > d->solver_niter ? d->solver_niter_max : d->solver_niter;   \
  }                                                                                                                      \
  static void env_ctx_##SUF(const PgttConfig* cfg, const PgttModel* m, const float* terrain, int T, int Bx, const PgttBuffers* B, \
                            long N, long e, uint64_t seed, int64_t off, OParams_##SUF* p, float* bf, OEnvCtx_##SUF* c) { \
    params_from_##SUF(m, B->params ? B->params + e : NULL, N, p);                                                        \
    int v = (B->variant && T > 0) ? B->variant[e] : 0;                                                                   \
    c->cfg = cfg; c->m = m; c->p = p; c->boxes = T > 0 ? terrain + (long)v*Bx*10 : NULL; c->nbox = T > 0 ? Bx : 0;       \
    c->box_friction = NULL;                                                                                              \
    if (B->box_friction && T > 0) { for (int b = 0; b < Bx; b++) bf[b] = B->box_friction[(long)b*N + e]; c->box_friction = bf; } \
    c->seed = seed; c->env_id = (uint32_t)(off + e);                                                                     \
  }                                                                                                                      \
  static void reset_all_##SUF(const PgttConfig* cfg, const PgttModel* m, const float* terrain, int T, int Bx, long N,    \
                              const PgttBuffers* B, uint64_t seed, int64_t off, const uint8_t* mask, int nthreads) {     \
    const int OD = cfg->method == PGTT_METHOD_BASELINE ? PGTT_OBS_BASELINE : PGTT_OBS, PD = OD + (PGTT_PRIV - PGTT_OBS); \
    (void)nthreads;                                                                                                      \
    _Pragma("omp parallel for num_threads(nthreads) schedule(static)")                                                   \
    for (long e = 0; e < N; e++) {                                                                                       \
      if (mask && !mask[e]) continue;                                                                                    \
      OData_##SUF* d = (OData_##SUF*)calloc(1, sizeof(OData_##SUF)); OInfo_##SUF in; OParams_##SUF p; OEnvCtx_##SUF c; float bf[PGTT_MAX_BOX]; \
      gather_##SUF(B, N, e, d, &in);                                                                                     \
      env_ctx_##SUF(cfg, m, terrain, T, Bx, B, N, e, seed, off, &p, bf, &c);                                             \
      RT obs[PGTT_OBS], priv[PGTT_PRIV];                                                                                 \
      task_reset_##SUF(&c, d, &in, obs, priv);                                                                           \
      scatter_##SUF(B, N, e, d, &in);                                                                                    \
      int contact[4]; static const int lofr[4] = {1, 0, 3, 2};                                                           \
      for (int f = 0; f < 4; f++) { contact[f] = 0;                                                                      \
        for (int cc = 0; cc < 8; cc++) if (d->contact[cc].foot == lofr[f] && d->contact[cc].box != -2 && d->contact[cc].dist < 0) contact[f] = 1; } \
      write_frame_##SUF(B, N, e, d, contact); write_dbg_##SUF(B, e, d);                                                  \
      for (int i = 0; i < OD; i++) B->obs_state[e*OD + i] = (float)obs[i];                                   \
      for (int i = 0; i < PD; i++) B->obs_priv[e*PD + i] = (float)priv[i];                                 \
      B->reward[e] = 0; B->done[e] = 0;                                                                                  \
      for (int k = 0; k < PGTT_NMETRIC; k++) B->metrics[(long)k*N + e] = 0;                                              \
      if (B->first_state) for (int i = 0; i < PGTT_S_CMD; i++) B->first_state[(long)i*N + e] = B->state[(long)i*N + e];  \
      if (B->first_obs) { for (int i = 0; i < OD; i++) B->first_obs[e*(OD + PD) + i] = (float)obs[i]; \
        for (int i = 0; i < PD; i++) B->first_obs[e*(OD + PD) + OD + i] = (float)priv[i]; }    \
      if (B->ep_metrics) for (int k = 0; k < PGTT_NMETRIC + 2; k++) B->ep_metrics[(long)k*N + e] = 0;                    \
      free(d);                                                                                                           \
    }                                                                                                                    \
  }                                                                                                                      \
  static void step_all_##SUF(const PgttConfig* cfg, const PgttModel* m, const float* terrain, int T, int Bx, long N,     \
                             const PgttBuffers* B, const float* action, uint64_t seed, int64_t off, int nthreads) {      \
    const int OD = cfg->method == PGTT_METHOD_BASELINE ? PGTT_OBS_BASELINE : PGTT_OBS, PD = OD + (PGTT_PRIV - PGTT_OBS); \
    (void)nthreads;                                                                                                      \
    _Pragma("omp parallel for num_threads(nthreads) schedule(static)")                                                   \
    for (long e = 0; e < N; e++) {                                                                                       \
      OData_##SUF* d = (OData_##SUF*)calloc(1, sizeof(OData_##SUF)); OInfo_##SUF in; OParams_##SUF p; OEnvCtx_##SUF c; float bf[PGTT_MAX_BOX]; \
      gather_##SUF(B, N, e, d, &in);                                                                                     \
      env_ctx_##SUF(cfg, m, terrain, T, Bx, B, N, e, seed, off, &p, bf, &c);                                             \
      RT obs[PGTT_OBS], priv[PGTT_PRIV], act[12], reward, done, metrics[PGTT_NMETRIC]; int contact[4];                   \
      for (int i = 0; i < 12; i++) act[i] = (RT)action[e*12 + i];                                                        \
      int prev_done = cfg->autoreset ? (B->done[e] != 0.0f) : 0;                                                         \
      if (prev_done) in.ep_steps = 0;                                                                                    \
      task_step_##SUF(&c, d, &in, act, obs, priv, &reward, &done, metrics, contact);                                     \
      write_frame_##SUF(B, N, e, d, contact); write_dbg_##SUF(B, e, d);                                                  \
      if (g_diag_resid) g_diag_resid[e] = (double)d->solver_resid_max;                                                   \
      if (g_diag_flags) g_diag_flags[e] = d->diag_flags;                                                                 \
      int idone = done != 0;                                                                                             \
      if (cfg->autoreset) {                                                                                              \
        in.ep_steps += 1;                                                                                                \
        if (in.ep_steps >= cfg->episode_length) idone = 1;                                                               \
        float keep = prev_done ? 0.0f : 1.0f;                                                                            \
        if (B->ep_metrics) {                                                                                             \
          for (int k = 0; k < PGTT_NMETRIC; k++) B->ep_metrics[(long)k*N + e] = (B->ep_metrics[(long)k*N + e] + (float)metrics[k])*keep; \
          B->ep_metrics[(long)PGTT_NMETRIC*N + e] = (B->ep_metrics[(long)PGTT_NMETRIC*N + e] + (float)reward)*keep;      \
          B->ep_metrics[(long)(PGTT_NMETRIC + 1)*N + e] = (B->ep_metrics[(long)(PGTT_NMETRIC + 1)*N + e] + 1.0f)*keep;   \
        }                                                                                                                \
      }                                                                                                                  \
      scatter_##SUF(B, N, e, d, &in);                                                                                    \
      for (int i = 0; i < OD; i++) B->obs_state[e*OD + i] = (float)obs[i];                                   \
      for (int i = 0; i < PD; i++) B->obs_priv[e*PD + i] = (float)priv[i];                                 \
      if (cfg->autoreset && idone) {                                                                                     \
        for (int i = 0; i < PGTT_S_CMD; i++) B->state[(long)i*N + e] = B->first_state[(long)i*N + e];                    \
        for (int i = 0; i < OD; i++) B->obs_state[e*OD + i] = B->first_obs[e*(OD + PD) + i];    \
        for (int i = 0; i < PD; i++) B->obs_priv[e*PD + i] = B->first_obs[e*(OD + PD) + OD + i]; \
      }                                                                                                                  \
      B->reward[e] = (float)reward; B->done[e] = (float)idone;                                                           \
      for (int k = 0; k < PGTT_NMETRIC; k++) B->metrics[(long)k*N + e] = (float)metrics[k];                              \
      if (B->interval_sums) {                                                                                            \
        for (int k = 0; k < PGTT_NMETRIC; k++) B->interval_sums[(long)k*N + e] += (float)metrics[k];                     \
        B->interval_sums[(long)PGTT_NMETRIC*N + e] += (float)reward;                                                     \
        B->interval_sums[(long)(PGTT_NMETRIC + 1)*N + e] += (float)idone;                                                \
      }                                                                                                                  \
      free(d);                                                                                                           \
    }                                                                                                                    \
  }
BATCH(f32, float)
BATCH(f64, double)

#define POST(SUF, RT)                                                                                                    \
  static void post_##SUF(const PgttConfig* cfg, const PgttModel* m, const PgttBuffers* B, const PgttOraclePostIn* in,    \
                         uint64_t seed, uint32_t env_id) {                                                               \
    OData_##SUF* d = (OData_##SUF*)calloc(1, sizeof(OData_##SUF)); OInfo_##SUF info; OParams_##SUF p; OEnvCtx_##SUF c;   \
    gather_##SUF(B, 1, 0, d, &info);                                                                                     \
    params_nominal_##SUF(m, &p);                                                                                         \
    c.cfg = cfg; c.m = m; c.p = &p; c.boxes = NULL; c.box_friction = NULL; c.nbox = 0; c.seed = seed; c.env_id = env_id; \
    static const int lof[4] = {1, 0, 3, 2};                                                                              \
    for (int i = 0; i < 19; i++) d->qpos[i] = (RT)in->qpos[i];                                                           \
    for (int i = 0; i < 18; i++) d->qvel[i] = (RT)in->qvel[i];                                                           \
    for (int i = 0; i < 49; i++) d->sensordata[i] = (RT)in->sensordata[i];                                               \
    for (int i = 0; i < 9; i++) d->site_imu_mat[i] = (RT)in->site_imu_mat[i];                                            \
    for (int f = 0; f < 4; f++) d->site_foot[lof[f]][2] = (RT)in->site_foot_z[f];                                        \
    RT act[12], mt[12], sz[PGTT_NSCAN], obs[PGTT_OBS], priv[PGTT_PRIV], reward, done, metrics[PGTT_NMETRIC];             \
    for (int i = 0; i < 12; i++) { d->actuator_force[i] = (RT)in->actuator_force[i]; act[i] = (RT)in->action[i];         \
      mt[i] = (RT)m->key_qpos[7 + i] + act[i]*(RT)cfg->action_scale; }                                                   \
    for (int i = 0; i < PGTT_NSCAN; i++) sz[i] = (RT)in->scan_z[i];                                                      \
    int contact[4] = {in->contact[0], in->contact[1], in->contact[2], in->contact[3]};                                   \
    task_post_##SUF(&c, d, &info, act, mt, contact, sz, obs, priv, &reward, &done, metrics);                             \
    /* keep the physics state untouched in the buffers: scatter only the info part */                                    \
    for (int i = 0; i < 19; i++) d->qpos[i] = (RT)B->state[PGTT_S_QPOS + i];                                             \
    for (int i = 0; i < 18; i++) { d->qvel[i] = (RT)B->state[PGTT_S_QVEL + i]; d->qacc_warmstart[i] = (RT)B->state[PGTT_S_QWARM + i]; } \
    scatter_##SUF(B, 1, 0, d, &info);                                                                                    \
    const int OD = cfg->method == PGTT_METHOD_BASELINE ? PGTT_OBS_BASELINE : PGTT_OBS, PD = OD + (PGTT_PRIV - PGTT_OBS); \
    for (int i = 0; i < OD; i++) B->obs_state[i] = (float)obs[i];                                                        \
    for (int i = 0; i < PD; i++) B->obs_priv[i] = (float)priv[i];                                                        \
    B->reward[0] = (float)reward; B->done[0] = (float)done;                                                              \
    for (int k = 0; k < PGTT_NMETRIC; k++) B->metrics[k] = (float)metrics[k];                                            \
    free(d);                                                                                                             \
  }
POST(f32, float)
POST(f64, double)
int pgtt_oracle_task_post(const PgttConfig* cfg, const PgttModel* m, const PgttBuffers* bufs, const PgttOraclePostIn* in, int fp64) {
  if (fp64) post_f64(cfg, m, bufs, in, 0, 0); else post_f32(cfg, m, bufs, in, 0, 0);
  return 0;
}
/* the same with the Philox key of a batch env (seed of the reset, GLOBAL env id): tests/test_gpu_fullsize.py feeds the DEVICE's physics outputs of every
   env-step of a rollout through the oracle's task layer, noise draws and command resampling included */
int pgtt_oracle_task_post_ex(const PgttConfig* cfg, const PgttModel* m, const PgttBuffers* bufs, const PgttOraclePostIn* in, int fp64,
                             uint64_t seed, uint32_t env_id) {
  if (fp64) post_f64(cfg, m, bufs, in, seed, env_id); else post_f32(cfg, m, bufs, in, seed, env_id);
  return 0;
}

int pgtt_oracle_reset(const PgttConfig* cfg, const PgttModel* m, const float* terrain, int T, int B, int N,
                      const PgttBuffers* bufs, uint64_t seed, int64_t env_id_offset, const uint8_t* mask, int fp64, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (fp64) reset_all_f64(cfg, m, terrain, T, B, N, bufs, seed, env_id_offset, mask, nthreads);
  else reset_all_f32(cfg, m, terrain, T, B, N, bufs, seed, env_id_offset, mask, nthreads);
  return 0;
}
int pgtt_oracle_step(const PgttConfig* cfg, const PgttModel* m, const float* terrain, int T, int B, int N,
                     const PgttBuffers* bufs, const float* action, uint64_t seed, int64_t env_id_offset, int fp64, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (fp64) step_all_f64(cfg, m, terrain, T, B, N, bufs, action, seed, env_id_offset, nthreads);
  else step_all_f32(cfg, m, terrain, T, B, N, bufs, action, seed, env_id_offset, nthreads);
  return 0;
}
int pgtt_oracle_sizeof_model(void) { return (int)sizeof(PgttModel); }
int pgtt_oracle_sizeof_config(void) { return (int)sizeof(PgttConfig); }
int pgtt_oracle_sizeof_buffers(void) { return (int)sizeof(PgttBuffers); }
