// pgtt_ppo.hip — trainer-side helper of libpgtt.so (gfx950): the policy part of the PPO minibatch loss and its gradient with
// respect to the policy network's output in ONE launch (+ a 64-lane finalise), instead of the ~100 elementwise / reduction
// launches (4-5 us each, launch-latency bound) that the same arithmetic costs as PyTorch ops forward and backward.
//
// Semantics = phase_guided_terrain_traversal_amd/ppo.py (_Learner._loss, the Brax PPO loss of training/train.py:135-161):
//   loc, raw = out[:, :A], out[:, A:];  scale = softplus(raw) + 1e-3
//   logp   = sum_j -0.5 z^2 - log(scale) - 0.5 log(2 pi),  z = (u - loc) / scale,   minus the tanh correction of u
//   ratio  = exp(logp - logp_old);  surr = min(ratio a, clip(ratio, 1 - e, 1 + e) a)
//   ent    = sum_j 0.5 + 0.5 log(2 pi) + log(scale) + 2 (log 2 - s - softplus(-2 s)),  s = loc + scale * eps
//   loss   = -mean(surr) - c * mean(ent)
// One thread per sample; the loss is summed per block and finished by a second, single-wave launch in a fixed order
// (deterministic).  No torch types: raw device pointers, the caller's stream.
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/pgtt.h"

namespace {

__device__ inline float softplus_t(float x) { return x > 20.f ? x : log1pf(expf(x)); }      // torch F.softplus (threshold 20)

template <int A>
__global__ __launch_bounds__(64) void ppo_policy_loss_kernel(const float* __restrict__ out, const float* __restrict__ u,
                                                              const float* __restrict__ logp_old, const float* __restrict__ adv,
                                                              const float* __restrict__ eps, int B, float clip, float cost,
                                                              float* __restrict__ partial, float* __restrict__ grad) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const bool on = i < B;
  const int ii = on ? i : B - 1;
  const float kHalfLog2Pi = 0.91893853320467274f, kLog2 = 0.69314718055994531f;
  float loc[A], sc[A], z[A], sg[A];
  float logp = 0.f, ent = 0.f;
#pragma unroll
  for (int j = 0; j < A; j++) {
    loc[j] = out[(long)ii * 2 * A + j];
    const float raw = out[(long)ii * 2 * A + A + j];
    sc[j] = softplus_t(raw) + 1e-3f;
    sg[j] = 1.0f / (1.0f + expf(-raw));                       // d softplus / d raw
    const float uj = u[(long)ii * A + j];
    z[j] = (uj - loc[j]) / sc[j];
    const float ls = logf(sc[j]);
    logp += -0.5f * z[j] * z[j] - ls - kHalfLog2Pi;
    logp -= 2.0f * (kLog2 - uj - softplus_t(-2.0f * uj));
    ent += 0.5f + kHalfLog2Pi + ls;
  }
  const float a = adv[ii];
  const float ratio = expf(logp - logp_old[ii]);
  const float s1 = ratio * a, s2 = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip) * a;
  const float surr = fminf(s1, s2);
  const float dlogp = s1 <= s2 ? a * ratio : 0.f;              // the min / clamp pair passes the gradient exactly when the unclipped term is taken
  const float invB = 1.0f / (float)B;
#pragma unroll
  for (int j = 0; j < A; j++) {
    const float e = eps[(long)ii * A + j];
    const float s = loc[j] + sc[j] * e;
    ent += 2.0f * (kLog2 - s - softplus_t(-2.0f * s));
    const float th = -2.0f * tanhf(s);                         // d ent / d s
    const float g_loc = -invB * dlogp * (z[j] / sc[j]) - cost * invB * th;
    const float g_sc = -invB * dlogp * ((z[j] * z[j] - 1.0f) / sc[j]) - cost * invB * (1.0f / sc[j] + th * e);
    if (on) { grad[(long)i * 2 * A + j] = g_loc; grad[(long)i * 2 * A + A + j] = g_sc * sg[j]; }
  }
  // block sums of surr and ent (64 lanes, xor butterfly through the LDS crossbar: once per launch, not hot)
  float v0 = on ? surr : 0.f, v1 = on ? ent : 0.f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { v0 += __shfl_xor(v0, o); v1 += __shfl_xor(v1, o); }
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = v0; partial[2 * blockIdx.x + 1] = v1; }
}

__global__ __launch_bounds__(64) void ppo_policy_loss_finish(const float* __restrict__ partial, int nblocks, int B, float cost, float* __restrict__ loss) {
  float v0 = 0.f, v1 = 0.f;
  for (int b = threadIdx.x; b < nblocks; b += 64) { v0 += partial[2 * b]; v1 += partial[2 * b + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { v0 += __shfl_xor(v0, o); v1 += __shfl_xor(v1, o); }
  if (threadIdx.x == 0) { loss[0] = -v0 / (float)B - cost * v1 / (float)B; loss[1] = -v0 / (float)B; loss[2] = v1 / (float)B; }
}

}  // namespace

extern "C" int pgtt_ppo_policy_loss(const float* out_Bx2A, const float* u_BxA, const float* logp_old_B, const float* adv_B,
                                    const float* eps_BxA, int B, int A, float clip_eps, float entropy_cost,
                                    float* partial_2xceilB64, float* loss_3, float* grad_Bx2A, void* stream) {
  if (!out_Bx2A || !u_BxA || !logp_old_B || !adv_B || !eps_BxA || !partial_2xceilB64 || !loss_3 || !grad_Bx2A || B <= 0) return PGTT_E_ARG;
  if (A != 12) return PGTT_E_ARG;                                // the Go2 has 12 actuators; one instantiation
  const int nb = (B + 63) / 64;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((ppo_policy_loss_kernel<12>), dim3(nb), dim3(64), 0, st, out_Bx2A, u_BxA, logp_old_B, adv_B, eps_BxA, B, clip_eps, entropy_cost,
                     partial_2xceilB64, grad_Bx2A);
  hipLaunchKernelGGL(ppo_policy_loss_finish, dim3(1), dim3(64), 0, st, partial_2xceilB64, nb, B, entropy_cost, loss_3);
  return hipGetLastError() == hipSuccess ? PGTT_OK : PGTT_E_HIP;
}
