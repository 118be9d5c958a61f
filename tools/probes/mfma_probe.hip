// Experiment for DESIGN.md 6 ("MFMA on the small dense solver blocks"): the Schur update of the arrowhead factorisation in the
// quad layout (lane = 4 * env + leg; every leg holds W = L_ll^-1 A_lb, 3 x 6; all four lanes need  S = sum_leg W_leg^T W_leg,
// the 21 entries of a symmetric 6 x 6) done two ways over the same data:
//   (A) vector ALU + DPP, as pgtt_physics_quad.hip.h::qarrow_factor does it: 63 FMAs per lane + 21 quad butterflies;
//   (B) v_mfma_f32_4x4x1_16B_f32, 16 independent 4x4 blocks = the 16 envs of the wave.  The instruction has NO reduction over
//       lanes (K = 1: D_b[i][j] += A_b[i] * B_b[j] with A_b[i] taken from lane i of block b), so the sum over legs and rows becomes
//       12 accumulating issues per 4 x 4 tile, and the operands have to be TRANSPOSED first: lane i of the env must hold column i
//       of (leg, row) for every leg - 4 quad broadcasts + 3 selects per 4-vector - and the result comes back as one column per lane
//       and has to be redistributed (16 broadcasts + selects) because every lane of the env needs all 21 entries.
// Prints cycles per env for both and checks they agree.    hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_probe.hip -o alt_build/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int CTRL> __device__ __forceinline__ float dpp(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_sum(float x) {
#pragma clang fp contract(off)
  x = x + dpp<0xB1>(x); x = x + dpp<0x4E>(x); return x;
}
__device__ __forceinline__ constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }

template <int REP> __global__ __launch_bounds__(64) void schur_valu(const float* __restrict__ w_in, float* __restrict__ out, long long* cyc) {
  const int lane = threadIdx.x, g = blockIdx.x * 64 + lane;
  float w[18];
#pragma unroll
  for (int i = 0; i < 18; i++) w[i] = w_in[(long)i * gridDim.x * 64 + g];
  float acc[21];
#pragma unroll
  for (int t = 0; t < 21; t++) acc[t] = 0.f;
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) acc[tri(i, j)] += quad_sum(w[i] * w[j] + w[6 + i] * w[6 + j] + w[12 + i] * w[12 + j]);
#pragma unroll
    for (int i = 0; i < 18; i++) asm volatile("" : "+v"(w[i]));       // keep the loop body from being hoisted
  }
  const long long t1 = __builtin_readcyclecounter();
#pragma unroll
  for (int t = 0; t < 21; t++) out[(long)t * gridDim.x * 64 + g] = acc[t];
  if (g == 0) cyc[0] = t1 - t0;
}

template <int REP> __global__ __launch_bounds__(64) void schur_mfma(const float* __restrict__ w_in, float* __restrict__ out, long long* cyc) {
  const int lane = threadIdx.x, g = blockIdx.x * 64 + lane, li = lane & 3;
  float w[18];
#pragma unroll
  for (int i = 0; i < 18; i++) w[i] = w_in[(long)i * gridDim.x * 64 + g];
  float acc[21];
#pragma unroll
  for (int t = 0; t < 21; t++) acc[t] = 0.f;
  const bool b0 = (li & 1) != 0, b1 = (li & 2) != 0;
  auto pick = [&](float x0, float x1, float x2, float x3) { const float lo = b0 ? x1 : x0, hi = b0 ? x3 : x2; return b1 ? hi : lo; };
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; r++) {
    // tiles of the 8 x 8 padded product: T00 (rows 0-3, cols 0-3), T10 (rows 4-7, cols 0-3), T11 (rows 4-7, cols 4-7)
    f4 T00 = {0, 0, 0, 0}, T10 = {0, 0, 0, 0}, T11 = {0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < 3; m++) {
      // own lane's picks of row m: element li of the low half, element 4 + li of the high half (columns 6, 7 are padding)
      const float lo_own = pick(w[6 * m + 0], w[6 * m + 1], w[6 * m + 2], w[6 * m + 3]);
      const float hi_own = pick(w[6 * m + 4], w[6 * m + 5], 0.f, 0.f);
#pragma unroll
      for (int leg = 0; leg < 4; leg++) {
        // lane i of the env needs W_leg[m][i]: every lane picks ITS index from its own registers, then the value held by
        // lane `leg`... is the wrong one - the pick index must be the RECEIVING lane's.  So: broadcast the four candidates
        // of lane `leg` and select by the own index.
        const float c0 = leg == 0 ? dpp<0x00>(w[6 * m + 0]) : (leg == 1 ? dpp<0x55>(w[6 * m + 0]) : (leg == 2 ? dpp<0xAA>(w[6 * m + 0]) : dpp<0xFF>(w[6 * m + 0])));
        const float c1 = leg == 0 ? dpp<0x00>(w[6 * m + 1]) : (leg == 1 ? dpp<0x55>(w[6 * m + 1]) : (leg == 2 ? dpp<0xAA>(w[6 * m + 1]) : dpp<0xFF>(w[6 * m + 1])));
        const float c2 = leg == 0 ? dpp<0x00>(w[6 * m + 2]) : (leg == 1 ? dpp<0x55>(w[6 * m + 2]) : (leg == 2 ? dpp<0xAA>(w[6 * m + 2]) : dpp<0xFF>(w[6 * m + 2])));
        const float c3 = leg == 0 ? dpp<0x00>(w[6 * m + 3]) : (leg == 1 ? dpp<0x55>(w[6 * m + 3]) : (leg == 2 ? dpp<0xAA>(w[6 * m + 3]) : dpp<0xFF>(w[6 * m + 3])));
        const float c4 = leg == 0 ? dpp<0x00>(w[6 * m + 4]) : (leg == 1 ? dpp<0x55>(w[6 * m + 4]) : (leg == 2 ? dpp<0xAA>(w[6 * m + 4]) : dpp<0xFF>(w[6 * m + 4])));
        const float c5 = leg == 0 ? dpp<0x00>(w[6 * m + 5]) : (leg == 1 ? dpp<0x55>(w[6 * m + 5]) : (leg == 2 ? dpp<0xAA>(w[6 * m + 5]) : dpp<0xFF>(w[6 * m + 5])));
        const float lo = pick(c0, c1, c2, c3), hi = pick(c4, c5, 0.f, 0.f);
        (void)lo_own; (void)hi_own;
        T00 = __builtin_amdgcn_mfma_f32_4x4x1f32(lo, lo, T00, 0, 0, 0);
        T10 = __builtin_amdgcn_mfma_f32_4x4x1f32(hi, lo, T10, 0, 0, 0);
        T11 = __builtin_amdgcn_mfma_f32_4x4x1f32(hi, hi, T11, 0, 0, 0);
      }
    }
    // lane j holds column j of every tile: T[i] = D[i][j].  Every lane needs all 21 entries S[i][j], j <= i: broadcast columns.
    float S[36];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      S[i * 6 + 0] = dpp<0x00>(T00[i]); S[i * 6 + 1] = dpp<0x55>(T00[i]); S[i * 6 + 2] = dpp<0xAA>(T00[i]); S[i * 6 + 3] = dpp<0xFF>(T00[i]);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      S[(4 + i) * 6 + 0] = dpp<0x00>(T10[i]); S[(4 + i) * 6 + 1] = dpp<0x55>(T10[i]); S[(4 + i) * 6 + 2] = dpp<0xAA>(T10[i]); S[(4 + i) * 6 + 3] = dpp<0xFF>(T10[i]);
      S[(4 + i) * 6 + 4] = dpp<0x00>(T11[i]); S[(4 + i) * 6 + 5] = dpp<0x55>(T11[i]);
    }
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) acc[tri(i, j)] += S[i * 6 + j];
#pragma unroll
    for (int i = 0; i < 18; i++) asm volatile("" : "+v"(w[i]));
  }
  const long long t1 = __builtin_readcyclecounter();
#pragma unroll
  for (int t = 0; t < 21; t++) out[(long)t * gridDim.x * 64 + g] = acc[t];
  if (g == 0) cyc[0] = t1 - t0;
}

int main() {
  const int blocks = 1024, n = blocks * 64, REP = 256;
  std::vector<float> h(18 * n);
  unsigned s = 7u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
  float *dw, *o1, *o2; long long *c1, *c2;
  hipMalloc(&dw, h.size() * 4); hipMalloc(&o1, 21 * n * 4); hipMalloc(&o2, 21 * n * 4); hipMalloc(&c1, 8); hipMalloc(&c2, 8);
  hipMemcpy(dw, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int it = 0; it < 2; it++) {
    hipLaunchKernelGGL(schur_valu<REP>, dim3(blocks), dim3(64), 0, 0, dw, o1, c1);
    hipLaunchKernelGGL(schur_mfma<REP>, dim3(blocks), dim3(64), 0, 0, dw, o2, c2);
  }
  hipDeviceSynchronize();
  std::vector<float> a(21 * n), b(21 * n); long long ca, cb;
  hipMemcpy(a.data(), o1, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, b.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(&ca, c1, 8, hipMemcpyDeviceToHost); hipMemcpy(&cb, c2, 8, hipMemcpyDeviceToHost);
  double err = 0, mag = 0;
  for (size_t i = 0; i < a.size(); i++) { err = fmax(err, fabs((double)a[i] - b[i])); mag = fmax(mag, fabs((double)a[i])); }
  printf("Schur update S = sum_leg W^T W (21 entries replicated in the 4 lanes of an env), quad layout, one wave per SIMD:\n");
  printf("  VALU + DPP  : %8.1f shader cycles per update\n", (double)ca / REP);
  printf("  MFMA 4x4x1  : %8.1f shader cycles per update   (36 MFMA issues + operand transposes + column broadcasts)\n", (double)cb / REP);
  printf("  max |diff| %.3g (values up to %.3g)\n", err, mag);
  return 0;
}
