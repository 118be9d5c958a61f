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
#include "../../include/pgtt_train.h"

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

// ------------------------------------------------------------------ weight / bias gradient of a Linear layer over a long batch
// dW[n][m] = sum_k dY[k][n] X[k][m],  db[n] = sum_k dY[k][n]   (X: [K][M] activations, dY: [K][N], torch weight layout [N][M])
// K is the minibatch (5120 rows), M and N are 1..512: the library GEMM picked for this shape walks K in a handful of
// workgroups (34 us whatever M x N) and the bias gradient is a second 15 us reduction.  Here K is split over S workgroups per
// 64x64 output tile (one wave each, 2x2 v_mfma_f32_32x32x2_f32 blocks, exact fp32), the column sums of dY ride along, and a
// second launch adds the S partial planes in a fixed order (deterministic; no atomics).
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(64) void linear_bwd_partial_kernel(const float* __restrict__ X, const float* __restrict__ dY, int K, int M, int N,
                                                                int kc, float* __restrict__ Pw, float* __restrict__ Pb) {
  const int lane = threadIdx.x, half = lane >> 5, col = lane & 31;
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64, s = blockIdx.z;
  const int kbeg = s * kc, kend = min(K, kbeg + kc);
  // operand columns of this lane (clamped; out-of-range columns are multiplied by 0)
  const int na = n0 + col, nb = n0 + 32 + col, ma = m0 + col, mb = m0 + 32 + col;
  const float fna = na < N ? 1.f : 0.f, fnb = nb < N ? 1.f : 0.f, fma_ = ma < M ? 1.f : 0.f, fmb = mb < M ? 1.f : 0.f;
  const int cna = na < N ? na : N - 1, cnb = nb < N ? nb : N - 1, cma = ma < M ? ma : M - 1, cmb = mb < M ? mb : M - 1;
  f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
  float sa = 0.f, sb = 0.f;
  // wave-uniform trip count (an MFMA needs both half-waves); eight k-pairs per trip, all 32 operand loads issued before the
  // 32 MFMAs (scheduling fence) so that one memory latency is paid per trip, not per pair
  constexpr int T = 8;
  for (int k2 = kbeg; k2 < kend; k2 += 2 * T) {
    float a0[T], a1[T], b0[T], b1[T];
#pragma unroll
    for (int t = 0; t < T; t++) {
      const int k = k2 + 2 * t + half;
      const int kk = k < kend ? k : kend - 1;
      a0[t] = dY[(long)kk * N + cna]; a1[t] = dY[(long)kk * N + cnb];
      b0[t] = X[(long)kk * M + cma]; b1[t] = X[(long)kk * M + cmb];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < T; t++) {
      const float kv = k2 + 2 * t + half < kend ? 1.f : 0.f;           // tail: zeros
      const float x0 = a0[t] * (fna * kv), x1 = a1[t] * (fnb * kv), y0 = b0[t] * fma_, y1 = b1[t] * fmb;
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, acc11, 0, 0, 0);
      sa += x0; sb += x1;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float* pw = Pw + (long)s * N * M;
  auto put = [&](const f32x16& acc, int nbase, int mbase) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int n = nbase + (r & 3) + 8 * (r >> 2) + 4 * half, m = mbase + col;       // C/D map of the 32x32 forms: row, column
      if (n < N && m < M) pw[(long)n * M + m] = acc[r];
    }
  };
  put(acc00, n0, m0); put(acc01, n0, m0 + 32); put(acc10, n0 + 32, m0); put(acc11, n0 + 32, m0 + 32);
  if (blockIdx.x == 0) {                       // column sums of dY: the two half-waves hold the even / odd rows
    sa += __shfl_xor(sa, 32); sb += __shfl_xor(sb, 32);
    if (half == 0) { if (na < N) Pb[(long)s * N + na] = sa; if (nb < N) Pb[(long)s * N + nb] = sb; }
  }
}

__global__ __launch_bounds__(256) void linear_bwd_reduce_kernel(const float* __restrict__ Pw, const float* __restrict__ Pb, int S, int NM, int N,
                                                                float* __restrict__ dW, float* __restrict__ db) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < NM) {
    float v = 0.f;
    int s = 0;
    for (; s + 8 <= S; s += 8) {                         // eight planes in flight, added in plane order
      float p[8];
#pragma unroll
      for (int t = 0; t < 8; t++) p[t] = Pw[(long)(s + t) * NM + i];
#pragma unroll
      for (int t = 0; t < 8; t++) v += p[t];
    }
    for (; s < S; s++) v += Pw[(long)s * NM + i];
    dW[i] = v;
  }
  if (i < N) { float v = 0.f; for (int s = 0; s < S; s++) v += Pb[(long)s * N + i]; db[i] = v; }
}

}  // namespace

extern "C" int pgtt_ppo_linear_backward(const float* x_KxM, const float* dy_KxN, int K, int M, int N, int S,
                                        float* partial_Sx_NM_plus_N, float* dw_NxM, float* db_N, void* stream) {
  if (!x_KxM || !dy_KxN || !partial_Sx_NM_plus_N || !dw_NxM || !db_N || K <= 0 || M <= 0 || N <= 0 || S <= 0) return PGTT_E_ARG;
  int kc = (K + S - 1) / S; kc += kc & 1;                       // even chunk: both half-waves start on their own parity
  const int Sused = (K + kc - 1) / kc;
  float* Pw = partial_Sx_NM_plus_N; float* Pb = partial_Sx_NM_plus_N + (long)S * N * M;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(linear_bwd_partial_kernel, dim3((M + 63) / 64, (N + 63) / 64, Sused), dim3(64), 0, st, x_KxM, dy_KxN, K, M, N, kc, Pw, Pb);
  const int NM = N * M;
  hipLaunchKernelGGL(linear_bwd_reduce_kernel, dim3((NM + 255) / 256), dim3(256), 0, st, Pw, Pb, Sused, NM, N, dw_NxM, db_N);
  return hipGetLastError() == hipSuccess ? PGTT_OK : PGTT_E_HIP;
}
