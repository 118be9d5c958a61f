/*
 * pgtt.h — C ABI of libpgtt.so: the MI355X-native vectorised Go2 "joystick_pgtt" environment step.
 *
 * This is the drop-in boundary for ONE hot path of NtagkasAlex/phase_guided_terrain_traversal:
 *   Joystick.step / Joystick.reset            (reference go2/joystick_pgtt.py:50-131, 141-231)
 *   mjx_env.step / mjx.forward (un-vendored)   (call sites go2/joystick_pgtt.py:72,78,146-148)
 *   Go2Env.compute_contact                     (go2/base.py:153-171)
 *   create_sensor_matrix / raycast_sensor      (go2/heightmap.py:10-67)
 *   domain_randomize (per-env model fields)    (go2/randomize.py:23-171) -> host side (randomize.py), handed over as
 *                                               PgttBuffers.params / variant / box_friction rows
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures (`stream` is a hipStream_t passed as void*).
 *   - every function returns 0 on success, a negative PGTT_E_* code otherwise; the message is
 *     available from pgtt_last_error(). Nothing throws across the boundary.
 *   - device buffers are CALLER-OWNED (torch-ROCm tensors); kernels are enqueued on the caller's
 *     stream and never synchronise. One handle per GPU; a handle is not thread-safe.
 *   - all floating point is fp32 (reference: jax_default_matmul_precision='highest',
 *     training/train.py:94); ints are int32.
 *   - persistent per-env state is SoA: buffer[row][env], row-major with leading dimension N.
 *   - index orders follow the reference: qpos[7:]/qvel[6:] are FL,FR,RL,RR (body-tree order,
 *     go2_mjx_feetonly.xml:103-187); ctrl/action/actuator_force and feet/contact/phase are
 *     FR,FL,RR,RL (actuator + go2_constants.py:55-74 order).
 */
#ifndef PGTT_H_
#define PGTT_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- sizes */
#define PGTT_NQ 19
#define PGTT_NV 18
#define PGTT_NU 12
#define PGTT_NBODY 13      /* moving bodies: 0 base, 1+3*leg+{0 hip,1 thigh,2 calf}; legs FL,FR,RL,RR */
#define PGTT_NLEG 4
#define PGTT_MAX_BOX 100   /* terrain_scene_mjx.xml:21 has 100 placeholder boxes */
#define PGTT_NCON 8        /* 4 plane-sphere + max_contact_points(4) sphere-box slots */
#define PGTT_NEFC 44       /* 12 joint limits + 8 contacts x 4 pyramid rows */
#define PGTT_SCAN_H 13     /* go2_constants.py:90-94 */
#define PGTT_SCAN_W 9
#define PGTT_NSCAN 117
#define PGTT_OBS 171       /* joystick_pgtt.py:336-349 */
#define PGTT_PRIV 215      /* joystick_pgtt.py:355-365 */
#define PGTT_OBS_BASELINE 162   /* go2/joystick.py:336-346: same rows without phase (8) and gait_freq (1) */
#define PGTT_PRIV_BASELINE 206  /* go2/joystick.py:352-362 */
/* PgttConfig.method: which of the reference's two task definitions the step computes (training/train.py:112-123, --method) */
#define PGTT_METHOD_PGTT 0      /* go2/joystick_pgtt.py + configs.default_config() */
#define PGTT_METHOD_BASELINE 1  /* go2/joystick.py + configs.baseline_config() */
#define PGTT_NREW 21       /* configs.py:31-59 */
#define PGTT_NMETRIC 22    /* 21 scaled reward terms + swing_peak (joystick_pgtt.py:122-125) */

/* ---------------------------------------------------------------- errors */
enum {
  PGTT_OK = 0,
  PGTT_E_ARG = -1,      /* null / out-of-range argument */
  PGTT_E_STATE = -2,    /* call order (e.g. step before bind_state) */
  PGTT_E_HIP = -3,      /* HIP runtime error (message has the hipError string) */
  PGTT_E_NODEVICE = -4  /* no usable GPU: there is NO CPU fallback in this library */
};

/* ---------------------------------------------------------------- reward term order (configs.py:31-59) */
enum {
  PGTT_R_TRACKING_LIN_VEL = 0, PGTT_R_TRACKING_ANG_VEL, PGTT_R_LIN_VEL_Z, PGTT_R_ANG_VEL_XY,
  PGTT_R_ORIENTATION, PGTT_R_DOF_POS_LIMITS, PGTT_R_POSE, PGTT_R_TERMINATION, PGTT_R_STAND_STILL,
  PGTT_R_TORQUES, PGTT_R_ACTION_RATE, PGTT_R_ENERGY, PGTT_R_FEET_CLEARANCE, PGTT_R_FEET_HEIGHT,
  PGTT_R_FEET_SLIP, PGTT_R_FEET_AIR_TIME, PGTT_R_FEET_PHASE, PGTT_R_FEET_SWING, PGTT_R_BODY_HEIGHT,
  PGTT_R_CONTACT, PGTT_R_CENTER
};

/* ---------------------------------------------------------------- model constants
 * Produced by phase_guided_terrain_traversal_amd/mjcf.py from the MJCF subset the reference trains on
 * (go2/xmls/go2_mjx_feetonly.xml + scene files) with go2/base.py:57-61 overrides applied.
 * Derived fields (invweight0, meaninertia) follow MuJoCo's compile-time set0 at qpos0 and are NOT
 * refreshed by domain randomisation, exactly like the reference. */
typedef struct PgttModel {
  float body_pos[PGTT_NBODY][3];      /* in parent frame */
  float body_quat[PGTT_NBODY][4];
  float body_ipos[PGTT_NBODY][3];
  float body_iquat[PGTT_NBODY][4];
  float body_mass[PGTT_NBODY];
  float body_inertia[PGTT_NBODY][3];  /* diagonal, inertial frame */
  float body_invweight0[PGTT_NBODY][2];
  float jnt_axis[12][3];              /* hinge axis in body frame, joint j belongs to body 1+j */
  float jnt_range[12][2];
  float jnt_solref[2];
  float jnt_solimp[5];
  float qpos0[PGTT_NQ];
  float dof_armature[PGTT_NV];
  float dof_damping[PGTT_NV];
  float dof_invweight0[PGTT_NV];
  int32_t act_dof[PGTT_NU];           /* actuator a drives dof act_dof[a] (FR,FL,RR,RL -> FL,FR,RL,RR) */
  float act_gain[PGTT_NU];            /* gainprm[:,0] */
  float act_bias[PGTT_NU][3];         /* biasprm[:,0:3] */
  float act_ctrlrange[PGTT_NU][2];
  float act_forcerange[PGTT_NU][2];
  float foot_geom_pos[PGTT_NLEG][3];  /* leg order FL,FR,RL,RR; in calf frame */
  float foot_radius[PGTT_NLEG];
  float foot_site_pos[PGTT_NLEG][3];
  float imu_pos[3];                   /* site on the base body; site quat is identity */
  /* geom-level contact parameters (mixed at run time like mjx collision_driver) */
  float foot_friction[3], foot_solref[2], foot_solimp[5], foot_margin, foot_gap, foot_solmix;
  float floor_friction[3], floor_solref[2], floor_solimp[5], floor_margin, floor_gap, floor_solmix;
  float box_friction[3], box_solref[2], box_solimp[5], box_margin, box_gap, box_solmix;
  /* foot-box pairs: max(foot_margin, box_margin) - max(foot_gap, box_gap) must be <= 0 (pgtt_create refuses more: only penetrating box pairs become
   * constraint rows; the reference has -0.001 / 0).  The plane contact takes any margin. */
  float box_rbound;                   /* stale compiled rbound of the 1x1x1 placeholder = sqrt(3) */
  int32_t foot_condim, floor_condim, box_condim;
  /* options */
  float timestep;
  float gravity[3];
  float impratio;
  float tolerance;
  float ls_tolerance;
  float meaninertia;
  int32_t iterations;
  int32_t ls_iterations;
  int32_t max_geom_pairs;
  int32_t max_contact_points;
  float key_qpos[PGTT_NQ];            /* keyframe "home" */
} PgttModel;

/* ---------------------------------------------------------------- task configuration
 * Mirror of go2/configs.py:6-79 default_config() (+ training/train.py:127-129 overrides). */
typedef struct PgttConfig {
  float ctrl_dt;                 /* 0.02 */
  float sim_dt;                  /* 0.005 */
  int32_t n_substeps;            /* round(ctrl_dt/sim_dt) = 4 */
  int32_t episode_length;        /* 1000 */
  float action_scale;            /* 0.5 */
  int32_t history_len;           /* 2 */
  int32_t history_update_steps;  /* 5 */
  float soft_joint_pos_limit_factor; /* 0.95 */
  float noise_level;             /* 1.0 */
  float noise_joint_pos, noise_joint_vel, noise_gyro, noise_gravity, noise_linvel, noise_heightscan;
  float reward_scale[PGTT_NREW]; /* order PGTT_R_* */
  float tracking_sigma, swing_height, base_feet_distance, phase_sigma;
  float cmd_u_max[3], cmd_u_min[3], cmd_b[3];
  float gait_freq[2];
  float scan_dist_x, scan_dist_y; /* 0.1, 0.1 */
  float scan_z_offset;            /* 0.6 (heightmap.py:38) */
  int32_t autoreset;              /* 1: fuse Episode(1000)+AutoReset-to-first-state wrapper semantics into step */
  int32_t method;                 /* PGTT_METHOD_*: observation layout, H_max definition, clearance target, air-time threshold */
  /* execution options (no counterpart in the reference; they choose between kernels that compute the same step) */
  int32_t lane_layout;            /* PGTT_LAYOUT_*: lanes per env of physics_kernel.  AUTO goes by the per-GPU batch; results are bit-identical
                                   * between batch sizes / shards only WITHIN one layout (between layouts: fp32 rounding), so a job that must
                                   * reproduce another one's bits pins the layout here.  In the OCT layout an env's roundings also depend on the
                                   * other seven envs of its wave (the plane contact joins the sub-lane split when any of them holds two box
                                   * contacts): there the shards must start at multiples of 8 envs as well.  QUAD does not care; HEX only when a foot of
                                   * the wave holds four box contacts (multiples of 4 envs then). */
  int32_t observe_form;           /* PGTT_OBSERVE_*: scan + obs + rewards as one kernel or as observe + task kernels */
  int32_t test_hooks;             /* non-zero: pgtt_set_test_overrides is allowed on this handle (fixture replay); 0 in production */
} PgttConfig;
enum { PGTT_LAYOUT_AUTO = 0, PGTT_LAYOUT_QUAD = 1, PGTT_LAYOUT_OCT = 2, PGTT_LAYOUT_HEX = 4 };   /* 4, 8, 16 lanes per env */
enum { PGTT_OBSERVE_FUSED = 0, PGTT_OBSERVE_SPLIT = 1 };

/* ---------------------------------------------------------------- persistent per-env state rows (float SoA) */
enum {
  PGTT_S_QPOS = 0,                 /* 19 */
  PGTT_S_QVEL = 19,                /* 18 */
  PGTT_S_QWARM = 37,               /* 18 qacc_warmstart */
  PGTT_S_CMD = 55,                 /* 3 */
  PGTT_S_PHASE = 58,               /* 4 (FR,FL,RR,RL) */
  PGTT_S_PHASE_DT = 62,
  PGTT_S_GAIT_FREQ = 63,
  PGTT_S_LAST_ACT = 64,            /* 12 */
  PGTT_S_LAST_LAST_ACT = 76,       /* 12 */
  PGTT_S_AIR_TIME = 88,            /* 4 */
  PGTT_S_SWING_PEAK = 92,          /* 4 */
  PGTT_S_HMAX = 96,                /* 4 */
  PGTT_S_HMIN = 100,               /* 4 */
  PGTT_S_MOTOR_TARGETS = 104,      /* 12 */
  PGTT_S_QERR_HIST = 116,          /* 24 */
  PGTT_S_QVEL_HIST = 140,          /* 24 */
  PGTT_S_LAST_CONTACT = 164,       /* 4 (0/1) */
  PGTT_NSTATE = 168
};
/* int SoA rows */
enum {
  PGTT_I_STEP = 0,                 /* info["step"] */
  PGTT_I_STEPS_UNTIL_CMD = 1,
  PGTT_I_RNG_CTR = 2,              /* counter of the per-env Philox stream */
  PGTT_I_EP_STEPS = 3,             /* EpisodeWrapper steps */
  PGTT_NISTATE = 4
};
/* sensor frame written by the physics kernel, read by the observe kernel (float SoA) */
enum {
  PGTT_F_GYRO = 0, PGTT_F_ACCEL = 3, PGTT_F_GLOBAL_LINVEL = 6, PGTT_F_GLOBAL_ANGVEL = 9,
  PGTT_F_LOCAL_LINVEL = 12, PGTT_F_UPVECTOR = 15, PGTT_F_GRAVITY = 18,
  PGTT_F_FEET_POS = 21,            /* 12: FR,FL,RR,RL x xyz in imu frame */
  PGTT_F_FEET_VEL = 33,            /* 12: world-frame site linvel */
  PGTT_F_ACT_FORCE = 45,           /* 12 */
  PGTT_F_CONTACT = 57,             /* 4 (0/1) FR,FL,RR,RL */
  PGTT_F_FOOT_SITE_Z = 61,         /* 4 world z of foot sites */
  PGTT_NFRAME = 65
};
/* per-env domain-randomised model rows (float SoA); NULL params => nominal PgttModel for every env */
enum {
  PGTT_P_BODY_MASS = 0,            /* 13 */
  PGTT_P_BASE_IPOS = 13,           /* 3 */
  PGTT_P_QPOS0 = 16,               /* 12 hinge zero offsets (qpos0[7:]) */
  PGTT_P_ARMATURE = 28,            /* 12 */
  PGTT_P_DAMPING = 40,             /* 12 */
  PGTT_P_GAIN = 52,                /* 12 gainprm[:,0] (actuator order) */
  PGTT_P_BIAS1 = 64,               /* 12 biasprm[:,1] */
  PGTT_P_FLOOR_FRICTION = 76,      /* 1 */
  PGTT_NPARAM = 77
};

/* Philox4x32-10 stream ids: uniform(seed, env, epoch, stream, i) = top 24 bits of word (i%4) of
 * philox4x32_10(key = (seed_lo, seed_hi), counter = (global env id, epoch, stream, i/4)) * 2^-24.
 * `epoch` is istate[PGTT_I_RNG_CTR], incremented once per reset and once per step. */
enum {
  PGTT_RS_GYRO = 0, PGTT_RS_GRAVITY = 1, PGTT_RS_QPOS = 2, PGTT_RS_QVEL = 3, PGTT_RS_SCAN = 4,
  PGTT_RS_CMD_Y = 5, PGTT_RS_CMD_W = 6, PGTT_RS_CMD_Z = 7, PGTT_RS_TIMER = 8,
  PGTT_RS_RESET_XY = 16, PGTT_RS_RESET_YAW = 17, PGTT_RS_RESET_VEL = 18, PGTT_RS_RESET_TIMER = 19,
  PGTT_RS_RESET_CMD = 20, PGTT_RS_RESET_FREQ = 21
};

/* device pointers, all caller-owned, all sized for N = num_envs given to pgtt_create */
typedef struct PgttBuffers {
  float*   state;        /* [PGTT_NSTATE][N] */
  int32_t* istate;       /* [PGTT_NISTATE][N] */
  float*   frame;        /* [PGTT_NFRAME][N] */
  float*   scan_z;       /* [N][PGTT_NSCAN] hit heights (info["heightscan"][...,2]), row-major 13x9 */
  float*   obs_state;    /* [N][state_dim]  row-major, as the trainer consumes it; dims by pgtt_obs_dims(): 171 / 162 */
  float*   obs_priv;     /* [N][priv_dim]   215 / 206 */
  float*   reward;       /* [N] */
  float*   done;         /* [N] 0/1 */
  float*   metrics;      /* [PGTT_NMETRIC][N] */
  float*   first_state;  /* [PGTT_S_CMD][N] qpos,qvel,qwarm captured at reset (AutoReset) */
  float*   first_obs;    /* [N][state_dim + priv_dim] */
  float*   ep_metrics;   /* [PGTT_NMETRIC + 2][N] running episode sums: metrics, sum_reward, length */
  /* optional */
  const float*   params;        /* [PGTT_NPARAM][N] or NULL */
  const int32_t* variant;       /* [N] terrain variant per env in [0, T), or NULL (=0).  pgtt_reset refuses labels outside the range (PGTT_E_ARG);
                                 * the step kernels clamp them, so a label edited afterwards can never index past the terrain tables */
  const float*   box_friction;  /* [PGTT_MAX_BOX][N] sliding friction per env per box, or NULL */
  int32_t* dbg_contact;  /* [N][PGTT_NCON][2] (foot 0..3 FL,FR,RL,RR ; geom: -1 plane, box idx, -2 none) or NULL */
  float*   dbg_dist;     /* [N][PGTT_NCON] or NULL */
  int32_t* dbg_niter;    /* [N] low 16 bits: max Newton iterations over the substeps of the last physics call (== iterations => truncated);
                          * bit PGTT_DBG_PEN_OVERFLOW (informational): some foot met more than 4 simultaneously penetrating boxes in a substep
                          * and the wave took the exact many-box pass of the collision stage (same contact set as MJX's max_geom_pairs /
                          * max_contact_points selection, only slower).  Or NULL */
  /* [PGTT_NMETRIC + 2][N] or NULL: per-env running sums of the step outputs (22 metrics, reward, done) since the caller last
   * cleared the block.  The reference's trainer averages the Episode-wrapper metrics per log interval (training/train.py:198-229);
   * with the sums kept by the step itself a logging interval costs ONE reduction over the envs instead of one per step. */
  float*   interval_sums;
} PgttBuffers;

#define PGTT_DBG_PEN_OVERFLOW 0x10000

typedef struct pgtt_env* pgtt_handle;

/* create a handle for `num_envs` environments on HIP device `device`. */
int pgtt_create(const PgttConfig* cfg, const PgttModel* model, int device, int num_envs, pgtt_handle* out);
int pgtt_destroy(pgtt_handle h);

/* terrain: T variants x B(<=100) boxes x [pos xyz, quat wxyz, half-size xyz] (terrains/level*.npy layout,
 * terrain/generator.py:368-391). HOST pointer, copied once into a resident device table. T = 0 => plane only. */
int pgtt_set_terrain(pgtt_handle h, const float* boxes_TxBx10, int T, int B);

int pgtt_bind(pgtt_handle h, const PgttBuffers* bufs);

/* Joystick.reset for the envs whose mask byte is non-zero (mask NULL => all). `seed` keys the Philox
 * streams; draws are a function of (seed, global env id, counter) only, so results do not depend on
 * how envs are sharded over GPUs. `env_id_offset` is the global id of local env 0.
 * With a terrain and per-env variant labels bound, the labels are range-checked first (one launch + a 4-byte read-back: the ONE place where the
 * library waits for `stream`): PGTT_E_ARG, nothing written, when one is outside [0, T).  Checked by every whole-batch reset (mask NULL) and by the
 * first reset of any kind after pgtt_bind / pgtt_set_terrain; a masked reset after that is asynchronous (labels edited in place since are clamped by
 * the kernels, never used as they are).  Skipped while a stream capture is under way (a capture cannot wait). */
int pgtt_reset(pgtt_handle h, uint64_t seed, int64_t env_id_offset, const uint8_t* mask_dev_or_null, void* stream);

/* Joystick.step for all envs. action is [N][12] row-major (FR,FL,RR,RL), device pointer. */
int pgtt_step(pgtt_handle h, const float* action_Nx12, void* stream);

/* sub-steps of pgtt_step exposed for testing / profiling */
int pgtt_physics(pgtt_handle h, const float* action_Nx12, void* stream);  /* 4 x mjx.step + sensors + contact flags */
int pgtt_observe(pgtt_handle h, const float* action_Nx12, void* stream);  /* scan + obs + rewards + bookkeeping */
int pgtt_scan(pgtt_handle h, float yaw_override_or_nan, void* stream);     /* K11 alone -> scan_z */

/* Interval reduction of the running sums the step kernels keep (PgttBuffers.interval_sums, [PGTT_NMETRIC + 2][N]: metrics, reward, done)
 * into out_dev[PGTT_NMETRIC + 3]: entry k < PGTT_NMETRIC + 2 receives the sum over the envs of row k, the last entry `env_steps` (the
 * caller's count of the env-steps the block holds, so that the PGTT_NMETRIC + 3 floats are ready for ONE all-reduce over the ranks);
 * accumulate != 0 adds to out_dev instead of overwriting it.  The rows are cleared for the next interval.  One launch: the rank-local
 * half of the logging reduction (the reference averages its metrics over the batch inside the jitted epoch,
 * training/train.py:142-161).  PGTT_E_STATE when no interval_sums buffer is bound. */
int pgtt_interval_reduce(pgtt_handle h, float* out_dev, float env_steps, int accumulate, void* stream);

/* TEST HOOKS (refused with PGTT_E_STATE unless the handle was created with PgttConfig.test_hooks != 0; tests/test_gpu_golden.py): the reference-generated fixtures of Joystick.step / Joystick.reset
 * (go2/joystick_pgtt.py:50-131,141-231 executed with jax.random stubbed and fake physics outputs) can only be replayed when every
 * uniform draw returns a fixed value (rng_value; NaN = the Philox streams) and when the step takes the 117 scan heights from
 * buf.scan_z instead of casting rays (scan_preset != 0). */
int pgtt_set_test_overrides(pgtt_handle h, float rng_value_or_nan, int scan_preset);

/* enable = 0 off, 1 time every step, n > 1 time every n-th step, starting with step n / 2 (an event record costs a few us of GPU idle).
 * time of the most recent physics / observe kernels, measured with HIP events on `stream`
 * (valid after the stream is synchronised; used by bench.py for the roofline figure). */
int pgtt_enable_timing(pgtt_handle h, int enable);
int pgtt_last_kernel_ms(pgtt_handle h, float* physics_ms, float* observe_ms);
/* mean kernel times over ALL steps since pgtt_enable_timing(h, 1): events are kept in a ring and read back when their
 * slot is re-used, so timing a long run does not stall the stream; synchronises on the steps still in flight */
int pgtt_kernel_ms_mean(pgtt_handle h, float* physics_ms, float* observe_ms, int* steps);

/* observation widths for cfg->method (env.observation_size of the reference, go2/joystick*.py) */
int pgtt_obs_dims(const PgttConfig* cfg, int* state_dim, int* priv_dim);
int pgtt_sizeof_model(void);
int pgtt_sizeof_config(void);
int pgtt_sizeof_buffers(void);
const char* pgtt_version(void);
/* "src=<SHA-256 of the kernel sources this library was built from>;flavor=<product|fastdiv|flip>": lets a measurement record say which build it
 * belongs to (bench.py quotes rocprofv3 counters only for the build they were collected on) */
const char* pgtt_build_info(void);
const char* pgtt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PGTT_H_ */
