// pgtt_policy.hip — trainer-side helpers of libpgtt.so (gfx950): the ACTING step of a roll-out around pgtt_step as two launches.
//
//   policy_act_kernel      normalise the observation, the 171-512-256-128-24 SiLU MLP of the reference's policy network
//                          (deploy/policy_net.py:36-71; Brax `make_policy_network`, training/train.py:135-161), the tanh-normal head
//                          (scale = softplus(raw) + 1e-3), one sample per env, its log-probability and tanh - and the roll-out storage rows
//                          of the step (observation, privileged observation, pre-tanh action, log-probability).  One launch instead of the
//                          ~25 library / elementwise launches the same arithmetic costs as PyTorch ops.
//   rollout_record_kernel  what the trainer keeps of a step AFTER pgtt_step: reward, done, truncation flag into the storage rows, the
//                          finished episodes' sums (return, length, 22 metric sums) into running accumulators, the step counter (one workgroup).
//
// The MLP runs on fp32 MFMA (v_mfma_f32_16x16x4_f32: exact fp32 products and sums, the policy's outputs agree with the fp32 reference to
// rounding).  A workgroup of eight waves (two per SIMD) owns 16 envs: the activations of a layer sit in LDS as X[env][k] (row stride K + 4 floats: the
// sixteen 128-bit reads of a quarter-wave fall into sixteen different bank quadruples), every wave computes an eighth of the layer's output
// neurons for all 16 envs, D[neuron][env] += W[neuron][k] X[env][k] with the weights as the A operand.  The k index of an MFMA is
// permuted so that a lane's four A values of four consecutive MFMAs are ONE 16-byte load: lane (i = l & 15, g = l >> 4) holds
// W[n0 + i][16 kb + 4 g + s] in step s, and the host packs the weights tile-major in exactly that order (pgtt_train.h), so that a (tile,
// k-block) is one fully coalesced 1 KB read; the B operand is X[i][16 kb + 4 g + s], one ds_read_b128.  With one wave per SIMD nothing but
// the software pipeline hides the L2 latency of the weight stream: DEPTH k-blocks of A operands are in flight.
// Weights total 1 MB and are the same for every workgroup: they stream from the XCD's L2 (8 MB of HBM reads per launch for 4096 envs), and THAT is
// the bound of this design at 4096 envs: 256 workgroups x 1 MB = 256 MB of L2 -> CU reads per launch; at 30 us that is 8.5 TB/s out of the eight
// L2s, with the MFMA pipe 45 % busy (rocprofv3 --pmc, round 4: SQ_VALU_MFMA_BUSY_CYCLES = 13.4 us of work per SIMD).  More envs per workgroup
// would halve the stream and leave half of the CUs without a workgroup.
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/pgtt_train.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kEnvs = 16;                         // envs per workgroup = the N of the 16x16x4 MFMA
constexpr int kH1 = 512, kH2 = 256, kH3 = 128;    // hidden layers of the reference's policy network
constexpr int kOut = 24, kOutPad = 32;            // loc | raw scale of 12 actuators
constexpr int kA = 12;
constexpr int kMaxObsPad = 224;                   // >= ceil16 of the widest observation (171 policy, 215 privileged)

__device__ inline float softplus_t(float x) { return x > 20.f ? x : log1pf(expf(x)); }      // torch F.softplus (threshold 20)
__device__ inline float silu(float x) { return x / (1.0f + expf(-x)); }

__device__ inline void philox4x32_10(unsigned k0, unsigned k1, unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const unsigned h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const unsigned h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    const unsigned n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// one layer for one wave: NT output tiles of 16 neurons starting at tile0, KB k-blocks of 16; Wp = packed weights [tile][kb][lane] float4.
// The k loop is unrolled in full (straight-line code: every "is there a next block" test is a compile-time one), so the compiler's wait counts
// are exact - with a rolled loop and conditional prefetches it drained the load queue (s_waitcnt vmcnt(0)) once per trip, and the A operands of
// DEPTH k-blocks that were meant to be in flight arrived one L2 round trip late every time (34.6 us per launch, MFMA pipe 39 % busy).
template <int NT, int KB, int DEPTH>
__device__ inline void layer_tiles(const float4* __restrict__ Wp, const float* __restrict__ X, int KS, int tile0, f32x4 (&acc)[NT]) {
  const int lane = threadIdx.x & 63;
  const float* xrow = X + (lane & 15) * KS + 4 * (lane >> 4);
  const float4* wl = Wp + (long)tile0 * KB * 64 + lane;
  float4 a[DEPTH][NT];
#pragma unroll
  for (int d = 0; d < DEPTH; d++)
#pragma unroll
    for (int t = 0; t < NT; t++) if (d < KB) a[d][t] = wl[(t * KB + d) * 64];
#pragma unroll
  for (int t = 0; t < NT; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 bq[2];
  bq[0] = *reinterpret_cast<const float4*>(xrow);
#pragma unroll
  for (int kb = 0; kb < KB; kb++) {
    const float4 b = bq[kb & 1];
    if (kb + 1 < KB) bq[(kb + 1) & 1] = *reinterpret_cast<const float4*>(xrow + 16 * (kb + 1));      // next B operand: its LDS round trip under this block's MFMAs
    float4 cur[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) cur[t] = a[kb % DEPTH][t];
    if (kb + DEPTH < KB) {
#pragma unroll
      for (int t = 0; t < NT; t++) a[kb % DEPTH][t] = wl[(t * KB + kb + DEPTH) * 64];
    }
    // the prefetches are issued HERE, DEPTH blocks ahead of their use: left to itself the scheduler sinks each load next to its first use
    // (fewer live registers, higher nominal occupancy) and one L2 round trip per k-block is exposed again
    __builtin_amdgcn_sched_barrier(0);
    // consecutive MFMAs go to different accumulators (dependent-accumulator latency 40 cycles against 32 of issue)
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[t].x, b.x, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[t].y, b.y, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[t].z, b.z, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[t].w, b.w, acc[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// bias + activation, D[neuron = 16 tile + 4 g + r][env = i] -> Xout[env][neuron] (one 16-byte LDS store per tile)
template <int NT, bool ACT>
__device__ inline void store_tiles(const f32x4 (&acc)[NT], const float* __restrict__ bias, int tile0, float* __restrict__ Xout, int KS) {
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int n = 16 * (tile0 + t) + 4 * g;
    const float4 bv = *reinterpret_cast<const float4*>(bias + n);
    float4 v = make_float4(acc[t][0] + bv.x, acc[t][1] + bv.y, acc[t][2] + bv.z, acc[t][3] + bv.w);
    if (ACT) { v.x = silu(v.x); v.y = silu(v.y); v.z = silu(v.z); v.w = silu(v.w); }
    *reinterpret_cast<float4*>(Xout + i * KS + n) = v;
  }
}

constexpr int kWaves = 8;                        // two waves per SIMD: one wave's operand round trips under the other's MFMAs
constexpr int kThreads = 64 * kWaves;

__global__ __launch_bounds__(kThreads) void policy_act_kernel(PgttPolicyActArgs a) {
  // activations: two buffers, alternately input and output of a layer (row strides K + 4)
  __shared__ __attribute__((aligned(16))) float bufA[kEnvs * (kH1 + 4)];      // layer-1 output (512), layer-3 output (128)
  __shared__ __attribute__((aligned(16))) float bufB[kEnvs * (kH2 + 4)];      // normalised observation (<= 224), layer-2 output (256), head (32)
  __shared__ float sh_lp[kEnvs * kA];
  const int tid = threadIdx.x, wave = tid >> 6;
  const int N = a.num_envs, od = a.obs_dim, kp0 = (od + 15) & ~15, ks0 = kp0 + 4, kb0n = kp0 >> 4;
  const long e0 = (long)blockIdx.x * kEnvs;
  const long t_row = a.counters ? a.counters[0] : 0;
  const bool row_ok = t_row >= 0 && t_row < a.store_rows;      // a caller that ran past its storage (no rewind) gets actions, not an out-of-bounds store
  const float4* W0 = reinterpret_cast<const float4*>(a.w[0]);
  // ---- stage the 16 observations (contiguous rows), normalised; the storage copies ride along.  Every load of the prologue is issued before the
  //      first store (unrolled, bounds by predicates): as load -> store loops the seven trips of the copies were seven dependent HBM round trips
  // (env, k) by shifts: 32 lanes walk the k of one env, the 16 half-waves take the 16 envs
  {
    constexpr int kObsTrips = kMaxObsPad / 32;                     // 7
    constexpr int kPrivTrips = (kEnvs * kMaxObsPad + kThreads - 1) / kThreads;      // 7: priv_dim <= kMaxObsPad
    const int env = tid >> 5, k0 = tid & 31;
    const long e = e0 + env;
    const bool copy_priv = a.store_priv && a.priv && row_ok;
    const int pd = a.priv_dim;
    const long pbase = e0 * pd, plim = (long)N * pd;
    float o[kObsTrips], mu[kObsTrips], sd[kObsTrips], pv[kPrivTrips];
#pragma unroll
    for (int j = 0; j < kObsTrips; j++) {
      const int k = k0 + 32 * j;
      const bool in = k < od && e < N;
      o[j] = in ? a.obs[e * od + k] : 0.f; mu[j] = in ? a.mean[k] : 0.f; sd[j] = in ? a.std[k] : 1.f;
    }
#pragma unroll
    for (int j = 0; j < kPrivTrips; j++) {
      const int idx = tid + kThreads * j;
      pv[j] = (copy_priv && idx < kEnvs * pd && pbase + idx < plim) ? a.priv[pbase + idx] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < kObsTrips; j++) {
      const int k = k0 + 32 * j;
      if (k < kp0) bufB[env * ks0 + k] = (k < od && e < N) ? (o[j] - mu[j]) / sd[j] : 0.f;
      if (a.store_obs && row_ok && k < od && e < N) a.store_obs[(t_row * N + e) * od + k] = o[j];
    }
#pragma unroll
    for (int j = 0; j < kPrivTrips; j++) {
      const int idx = tid + kThreads * j;
      if (copy_priv && idx < kEnvs * pd && pbase + idx < plim) a.store_priv[t_row * N * pd + pbase + idx] = pv[j];
    }
  }
  __syncthreads();
  // ---- layer 1: od -> 512 (4 tiles per wave); the k-block count depends on the observation width: 11 for 171 / 162 observations (-> 176), 14 for
  //      215 (-> 224); any other width takes the untuned loop
  {
    f32x4 acc[4];
    if (kb0n == 11) layer_tiles<4, 11, 4>(W0, bufB, ks0, 4 * wave, acc);
    else if (kb0n == 14) layer_tiles<4, 14, 4>(W0, bufB, ks0, 4 * wave, acc);
    else {          // any other width: one k-block at a time (kept for completeness, not tuned)
      const int lane = tid & 63;
      const float* xrow = bufB + (lane & 15) * ks0 + 4 * (lane >> 4);
#pragma unroll
      for (int t = 0; t < 4; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int kb = 0; kb < kb0n; kb++) {
        const float4 b = *reinterpret_cast<const float4*>(xrow + 16 * kb);
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const float4 w = W0[((long)(4 * wave + t) * kb0n + kb) * 64 + lane];
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, b.x, acc[t], 0, 0, 0); acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, b.y, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, b.z, acc[t], 0, 0, 0); acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, b.w, acc[t], 0, 0, 0);
        }
      }
    }
    store_tiles<4, true>(acc, a.b[0], 4 * wave, bufA, kH1 + 4);
  }
  __syncthreads();
  // ---- layer 2: 512 -> 256 (2 tiles per wave, 32 k-blocks)
  {
    f32x4 acc[2];
    layer_tiles<2, kH1 / 16, 8>(reinterpret_cast<const float4*>(a.w[1]), bufA, kH1 + 4, 2 * wave, acc);
    store_tiles<2, true>(acc, a.b[1], 2 * wave, bufB, kH2 + 4);
  }
  __syncthreads();
  // ---- layer 3: 256 -> 128 (1 tile per wave, 16 k-blocks)
  {
    f32x4 acc[1];
    layer_tiles<1, kH2 / 16, 8>(reinterpret_cast<const float4*>(a.w[2]), bufB, kH2 + 4, wave, acc);
    store_tiles<1, true>(acc, a.b[2], wave, bufA, kH3 + 4);
  }
  __syncthreads();
  // ---- head: 128 -> 24 (padded to 32: two tiles, waves 0 and 1), no activation
  if (wave < 2) {
    f32x4 acc[1];
    layer_tiles<1, kH3 / 16, 8>(reinterpret_cast<const float4*>(a.w[3]), bufA, kH3 + 4, wave, acc);
    store_tiles<1, false>(acc, a.b[3], wave, bufB, kOutPad + 4);
  }
  __syncthreads();
  // ---- tanh-normal head: one thread per (env, actuator)
  if (tid < kEnvs * kA) {
    const int env = tid / kA, j = tid - env * kA;
    const long e = e0 + env;
    const float loc = bufB[env * (kOutPad + 4) + j], raw = bufB[env * (kOutPad + 4) + kA + j];
    const float sc = softplus_t(raw) + 1e-3f;
    float eps;
    if (a.eps) eps = e < N ? a.eps[e * kA + j] : 0.f;
    else {
      // one Philox block per (env, step, actuator pair-of-pairs): Box-Muller on two of its four words
      const unsigned long long draw = a.counters ? (unsigned long long)a.counters[1] : 0ull;
      unsigned c0 = (unsigned)(a.env_id_offset + e), c1 = (unsigned)draw, c2 = (unsigned)(draw >> 32) ^ 0x50475454u, c3 = (unsigned)(j >> 1);
      philox4x32_10((unsigned)a.seed, (unsigned)(a.seed >> 32), c0, c1, c2, c3);
      const float u1 = ((float)(c0 >> 8) + 0.5f) * (1.0f / 16777216.0f), u2 = (float)(c1 >> 8) * (1.0f / 16777216.0f);
      const float r = sqrtf(-2.0f * logf(u1));
      float sn, cs; sincosf(6.28318530717958648f * u2, &sn, &cs);
      eps = (j & 1) ? r * sn : r * cs;
    }
    const float u = a.deterministic ? loc : loc + sc * eps;
    const float z = (u - loc) / sc;
    const float kHalfLog2Pi = 0.91893853320467274f, kLog2 = 0.69314718055994531f;
    sh_lp[tid] = -0.5f * z * z - logf(sc) - kHalfLog2Pi - 2.0f * (kLog2 - u - softplus_t(-2.0f * u));
    if (e < N) {
      const float th = tanhf(u);
      a.act[e * kA + j] = th;
      if (a.store_u && row_ok) a.store_u[(t_row * N + e) * kA + j] = u;
      if (a.head) { a.head[e * kOut + j] = loc; a.head[e * kOut + kA + j] = raw; }
    }
  }
  __syncthreads();
  if (tid < kEnvs && e0 + tid < N && a.store_logp && row_ok) {
    float lp = 0.f;
#pragma unroll
    for (int j = 0; j < kA; j++) lp += sh_lp[tid * kA + j];
    a.store_logp[t_row * N + e0 + tid] = lp;
  }
}

// ------------------------------------------------------------------ what the trainer keeps of a step after pgtt_step
constexpr int kRecSums = PGTT_NMETRIC + 3;        // finished episodes: 22 metric sums, return, length, count
constexpr int kRecThreads = 1024;

// ONE workgroup walks all envs (a step's bookkeeping is 12 bytes in and 12 bytes out per env: at 4096 envs four coalesced trips of sixteen waves):
// no partial sums across workgroups, no atomics, no "last block" hand-over - the step counters are advanced by the same workgroup that read them.
// The finished-episode sums are formed only on a step in which some env's episode ended (a few envs per step), in a fixed order: deterministic.
__global__ __launch_bounds__(kRecThreads) void rollout_record_kernel(PgttRolloutRecordArgs a) {
  __shared__ float sh[kRecSums][kRecThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = a.num_envs;
  const long t_row = a.counters[0];
  const long draws = a.counters[1];
  const bool row_ok = t_row >= 0 && t_row < a.store_rows;
  float sums[kRecSums];
#pragma unroll
  for (int k = 0; k < kRecSums; k++) sums[k] = 0.f;
  bool any = false;
  for (int e = tid; e < N; e += kRecThreads) {
    const float done = a.done[e], rew = a.reward[e];
    const bool fallen = a.up_z[e] < 0.f;
    const bool trunc = (a.ep_steps[e] >= a.episode_length) && !fallen;
    if (row_ok) {
      a.store_rew[t_row * N + e] = rew * a.reward_scaling;
      a.store_done[t_row * N + e] = done;
      a.store_trunc[t_row * N + e] = trunc ? 1.f : 0.f;
    }
    if (done != 0.f) {
      any = true;
#pragma unroll
      for (int k = 0; k < PGTT_NMETRIC + 2; k++) sums[k] += a.ep_metrics[(long)k * N + e] * done;
      sums[PGTT_NMETRIC + 2] += done;
    }
  }
  if (__syncthreads_or(any ? 1 : 0)) {
#pragma unroll
    for (int k = 0; k < kRecSums; k++) {
      float v = sums[k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) sh[k][wave] = v;
    }
    __syncthreads();
    if (tid < kRecSums) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < kRecThreads / 64; w++) v += sh[tid][w];
      a.episode_sums[tid] += v;
    }
  }
  if (tid == 0) { a.counters[0] = t_row + 1; a.counters[1] = draws + 1; }
}

}  // namespace

extern "C" int pgtt_policy_act(const PgttPolicyActArgs* args, void* stream) {
  if (!args || !args->obs || !args->mean || !args->std || !args->act || args->num_envs <= 0) return PGTT_E_ARG;
  for (int l = 0; l < 4; l++) if (!args->w[l] || !args->b[l]) return PGTT_E_ARG;
  if (args->obs_dim <= 0 || ((args->obs_dim + 15) & ~15) > kMaxObsPad) return PGTT_E_ARG;
  if (args->store_priv && (!args->priv || args->priv_dim <= 0 || args->priv_dim > kMaxObsPad)) return PGTT_E_ARG;
  if ((args->store_obs || args->store_priv || args->store_u || args->store_logp) && args->store_rows <= 0) return PGTT_E_ARG;
  hipLaunchKernelGGL(policy_act_kernel, dim3((args->num_envs + kEnvs - 1) / kEnvs), dim3(kThreads), 0, (hipStream_t)stream, *args);
  return hipGetLastError() == hipSuccess ? PGTT_OK : PGTT_E_HIP;
}

extern "C" int pgtt_rollout_record(const PgttRolloutRecordArgs* args, void* stream) {
  if (!args || !args->reward || !args->done || !args->ep_steps || !args->up_z || !args->ep_metrics || !args->store_rew || !args->store_done ||
      !args->store_trunc || !args->counters || !args->episode_sums || args->num_envs <= 0 || args->store_rows <= 0) return PGTT_E_ARG;
  hipLaunchKernelGGL(rollout_record_kernel, dim3(1), dim3(kRecThreads), 0, (hipStream_t)stream, *args);
  return hipGetLastError() == hipSuccess ? PGTT_OK : PGTT_E_HIP;
}

extern "C" int pgtt_policy_packed_floats(int in_dim, int out_dim) { return ((out_dim + 15) & ~15) * ((in_dim + 15) & ~15); }
extern "C" int pgtt_sizeof_policy_act_args(void) { return (int)sizeof(PgttPolicyActArgs); }
extern "C" int pgtt_sizeof_rollout_record_args(void) { return (int)sizeof(PgttRolloutRecordArgs); }
