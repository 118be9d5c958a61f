/*
 * pgtt_train.h - trainer-side helper kernels of libpgtt.so.  NOT part of the environment boundary (include/pgtt.h): they replace
 * no entry point of the reference (its PPO is Brax's, configured at training/train.py:135-161) and exist only because the
 * repo's own torch PPO loop was launch-bound on ~100 elementwise kernels per minibatch.
 * Same conventions as pgtt.h: plain C, device pointers (float32), enqueued on the caller's stream, no synchronisation,
 * 0 / negative PGTT_E_* return codes, message through the env header's error channel.
 */
#ifndef PGTT_TRAIN_H_
#define PGTT_TRAIN_H_
#include "pgtt.h"
#ifdef __cplusplus
extern "C" {
#endif
/* The policy part of the PPO minibatch loss, -mean(min(r a, clip(r) a)) - entropy_cost * mean(entropy), of a tanh-normal
 * policy head (loc | raw scale, scale = softplus(raw) + 1e-3; the Brax loss configured at training/train.py:135-161) and its
 * gradient with respect to the network output, in one launch + a single-wave finish instead of ~100 elementwise launches.
 * All pointers are device pointers (float32); partial holds 2 * ceil(B / 64) floats of scratch; loss_3 = {total, policy
 * term, mean entropy}; A must be 12.  Enqueued on `stream`, no synchronisation. */
int pgtt_ppo_policy_loss(const float* out_Bx2A, const float* u_BxA, const float* logp_old_B, const float* adv_B,
                         const float* eps_BxA, int B, int A, float clip_eps, float entropy_cost,
                         float* partial_2xceilB64, float* loss_3, float* grad_Bx2A, void* stream);
/* Weight and bias gradient of a Linear layer over a long batch, dW[n][m] = sum_k dY[k][n] X[k][m],
 * db[n] = sum_k dY[k][n] (X [K][M], dY [K][N], dW in torch's [N][M] layout), K split over S workgroups per 64x64 tile on
 * fp32 MFMA, summed in a fixed order.  partial holds S * (N * M + N) floats of scratch.  Device pointers, caller's stream. */
int pgtt_ppo_linear_backward(const float* x_KxM, const float* dy_KxN, int K, int M, int N, int S,
                             float* partial_Sx_NM_plus_N, float* dw_NxM, float* db_N, void* stream);

/* ---------------------------------------------------------------- the acting step of a roll-out around the env step (round 4)
 * What the reference's trainer does between two env steps (Brax's `generate_unroll` with the networks of training/train.py:135-161;
 * the deployed form of the same network is deploy/policy_net.py:36-71): normalise the "state" observation, policy MLP
 * obs -> 512 -> 256 -> 128 -> 24 with SiLU, tanh-normal head (loc | raw, scale = softplus(raw) + 1e-3), draw u = loc + scale * eps,
 * action = tanh(u), log-probability of u; and after the env step: reward / done / truncation into the roll-out storage, the finished
 * episodes' sums into the logging accumulators.  Two launches (pgtt_policy_act, pgtt_rollout_record) instead of ~60.
 *
 * Weights are PACKED by the caller (phase_guided_terrain_traversal_amd/ppo.py::pack_linear) for the fp32 MFMA tiles: a layer [out][in] is
 * zero-padded to multiples of 16 and stored as [out / 16][in / 16][g = 0..3][i = 0..15][s = 0..3] = W[16 tile + i][16 kb + 4 g + s]
 * (pgtt_policy_packed_floats(in, out) floats), its bias zero-padded to a multiple of 16.  All pointers are device pointers (float32
 * unless said otherwise), kernels are enqueued on `stream`, nothing synchronises; 0 / negative PGTT_E_* codes as in pgtt.h. */
typedef struct PgttPolicyActArgs {
  const float* obs;            /* [N][obs_dim] the env's observation rows (PgttBuffers.obs_state) */
  const float* priv;           /* [N][priv_dim] or NULL: only copied into store_priv */
  const float* mean;           /* [obs_dim] running statistics of the observation */
  const float* std;            /* [obs_dim] */
  const float* w[4];           /* packed weights of the four layers (hidden sizes 512, 256, 128; head 24 = 2 x 12) */
  const float* b[4];           /* padded biases */
  const float* eps;            /* [N][12] standard-normal draws, or NULL: drawn in the kernel (Philox4x32-10 keyed by seed, global env id, draw counter) */
  float* act;                  /* [N][12] tanh(u): the action for pgtt_step */
  float* head;                 /* [N][24] or NULL: the network's raw output (loc | raw scale) */
  float* store_obs;            /* [T][N][obs_dim] or NULL: row counters[0] receives obs */
  float* store_priv;           /* [T][N][priv_dim] or NULL */
  float* store_u;              /* [T][N][12] or NULL: the pre-tanh sample */
  float* store_logp;           /* [T][N] or NULL */
  const int64_t* counters;     /* device int64[2] {storage row t, draw counter} or NULL (row 0, counter 0); advanced by pgtt_rollout_record */
  uint64_t seed;
  int64_t env_id_offset;       /* global id of env 0 of this shard (draws are keyed by global ids, like the env's own) */
  int32_t num_envs, obs_dim, priv_dim;
  int32_t deterministic;       /* non-zero: u = loc (evaluation, deploy/policy_net.py:71-80) */
  int32_t store_rows;          /* T of the store_* blocks: a call whose row counters[0] is outside [0, T) stores nothing (the action is still produced) */
} PgttPolicyActArgs;
int pgtt_policy_act(const PgttPolicyActArgs* args, void* stream);
int pgtt_policy_packed_floats(int in_dim, int out_dim);

typedef struct PgttRolloutRecordArgs {
  const float* reward;         /* [N] PgttBuffers.reward */
  const float* done;           /* [N] */
  const int32_t* ep_steps;     /* [N] PgttBuffers.istate row PGTT_I_EP_STEPS (after the step; AutoReset leaves the finished episode's count) */
  const float* up_z;           /* [N] PgttBuffers.frame row PGTT_F_UPVECTOR + 2: truncation = episode_length reached and not fallen */
  const float* ep_metrics;     /* [PGTT_NMETRIC + 2][N] PgttBuffers.ep_metrics */
  float* store_rew;            /* [T][N] row counters[0] */
  float* store_done;           /* [T][N] */
  float* store_trunc;          /* [T][N] */
  int64_t* counters;           /* device int64[2]: both advanced by one per call */
  float* episode_sums;         /* [PGTT_NMETRIC + 3] += over the envs whose episode ended: 22 metric sums, return, length, count */
  float reward_scaling;
  int32_t num_envs, episode_length;
  int32_t store_rows;          /* T of the store_* blocks: a call whose row counters[0] is outside [0, T) stores no row (sums and counters still advance) */
} PgttRolloutRecordArgs;
int pgtt_rollout_record(const PgttRolloutRecordArgs* args, void* stream);
int pgtt_sizeof_policy_act_args(void);
int pgtt_sizeof_rollout_record_args(void);
#ifdef __cplusplus
}
#endif
#endif /* PGTT_TRAIN_H_ */
