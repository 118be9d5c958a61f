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
#ifdef __cplusplus
}
#endif
#endif /* PGTT_TRAIN_H_ */
