// Probe: can the observe launch of a step run in the TAIL of the physics launch?  (docs/HISTORY.md 5.6)
// physics_kernel holds one wave per SIMD with all 512 registers, its waves finish between ~0.92 and 1.0 of the launch, and the observe
// launch (one wave per env, 128 registers, four per SIMD) may only start when the last physics wave has left.  Here: kernel P = 1024 waves,
// one per SIMD (amdgpu_waves_per_eu(1,1)), each busy for 100 + (0..20) us, then it publishes its 4 "envs" (data rows, release fence, queue
// entries); kernel O = 4096 waves of 128 registers, wave i takes queue entry i (bounded spin), reads the env's data and works for a latency
// chain of ~LAT dependent loads.  Modes: sequential on one stream, concurrent on two streams.  Reports launch-to-end times, how many O waves
// started before P ended, and whether O saw P's data (plain loads / agent-scope atomic loads).
//   hipcc --offload-arch=gfx950 -O3 -o overlap_probe overlap_probe.hip && ./overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int NB = 1024, EPW = 4, NE = NB * EPW, ROW = 128;

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void kP(float* data, unsigned* q, unsigned* cnt, unsigned epoch, long long* tl, int base_us, int spread_us) {
  const long long t0 = wall_clock64();
  asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a255, v255" ::: "v255", "a255");      // 256 + 256 registers, like physics_kernel
  const int b = blockIdx.x;
  const unsigned h = (b * 2654435761u) >> 16;
  const long long dur = 100ll * (base_us + (int)(h % (unsigned)(spread_us + 1)));
  float acc = (float)threadIdx.x;
  while (wall_clock64() - t0 < dur) { for (int i = 0; i < 64; i++) acc = acc * 1.0000001f + 0.5f; }
  for (int k = 0; k < EPW; k++) {
    const int e = b * EPW + k;
    data[(long)e * ROW + threadIdx.x] = (float)(epoch * 8192u + e) + (acc == 12345.f ? 1.f : 0.f);
    data[(long)e * ROW + 64 + threadIdx.x] = (float)(epoch * 8192u + e) + 0.5f;
  }
  __threadfence();      // agent-scope release: the rows above are visible to every XCD before the queue entries are
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(cnt, (unsigned)EPW);
    for (int k = 0; k < EPW; k++) __hip_atomic_store(&q[t + k], (epoch << 16) | (unsigned)(b * EPW + k), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    tl[2 * b] = t0; tl[2 * b + 1] = wall_clock64();
  }
}

template <bool ATOMIC_LOADS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void kO(const float* data, const unsigned* q, unsigned epoch, const int* chain, int lat, float* out, long long* tl, unsigned* bad,
                                                                                   int queued, long long spin_ticks) {
  const long long t0 = wall_clock64();
  asm volatile("v_mov_b32 v127, 0" ::: "v127");      // 128 registers, like observe_kernel
  const int b = blockIdx.x;
  int e = b;
  bool timeout = false;
  if (queued) {
    unsigned v = 0;
    for (;;) {
      v = __hip_atomic_load(&q[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((v >> 16) == epoch) break;
      if (wall_clock64() - t0 > spin_ticks) { timeout = true; break; }
      __builtin_amdgcn_s_sleep(8);
    }
    e = timeout ? b : (int)(v & 0xffffu);
    // plain loads need the agent-scope acquire (invalidates this XCD's L2); the atomic (sc1) loads of the other variant do without
    if (!ATOMIC_LOADS) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  const long long t1 = wall_clock64();
  float x0, x1;
  if (ATOMIC_LOADS) {
    x0 = __hip_atomic_load(&data[(long)e * ROW + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    x1 = __hip_atomic_load(&data[(long)e * ROW + 64 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    x0 = data[(long)e * ROW + threadIdx.x]; x1 = data[(long)e * ROW + 64 + threadIdx.x];
  }
  const float w0 = (float)(epoch * 8192u + e), w1 = w0 + 0.5f;
  if (!timeout && (x0 != w0 || x1 != w1)) atomicAdd(bad, 1u);
  if (timeout) atomicAdd(bad + 1, 1u);
  int p = (e * 64 + threadIdx.x) & 4095;
  for (int i = 0; i < lat; i++) p = chain[p];      // dependent loads: the latency chain of an observe wave
  out[(long)e * 64 + threadIdx.x] = x0 + x1 + (float)p;
  if (threadIdx.x == 0) { tl[2 * b] = t0; tl[2 * b + 1] = wall_clock64(); tl[2 * NE + b] = t1; }
}

int main(int argc, char** argv) {
  const int lat = argc > 1 ? atoi(argv[1]) : 40, base_us = argc > 2 ? atoi(argv[2]) : 150, spread_us = argc > 3 ? atoi(argv[3]) : 16;
  float *data, *out; unsigned *q, *cnt, *bad; int* chain; long long *tlP, *tlO;
  CK(hipMalloc(&data, sizeof(float) * NE * ROW)); CK(hipMalloc(&out, sizeof(float) * NE * 64)); CK(hipMalloc(&q, 4 * NE)); CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&bad, 8));
  CK(hipMalloc(&chain, 4 * 4096)); CK(hipMalloc(&tlP, 16 * NB)); CK(hipMalloc(&tlO, 24 * NE));
  std::vector<int> hc(4096); for (int i = 0; i < 4096; i++) hc[i] = (i * 1237 + 611) & 4095;
  CK(hipMemcpy(chain, hc.data(), 4 * 4096, hipMemcpyHostToDevice));
  CK(hipMemset(q, 0, 4 * NE)); CK(hipMemset(bad, 0, 8));
  hipStream_t sA, sB; CK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
  hipEvent_t e0, e1, eO0, eO1, eJoin, eFork; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&eO0)); CK(hipEventCreate(&eO1)); CK(hipEventCreate(&eJoin)); CK(hipEventCreate(&eFork));
  int can = -1; hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0); printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  unsigned epoch = 1;
  auto report = [&](const char* tag, float msTot, float msO) {
    std::vector<long long> hP(2 * NB), hO(3 * NE); unsigned hb[2];
    hipMemcpy(hP.data(), tlP, 16 * NB, hipMemcpyDeviceToHost); hipMemcpy(hO.data(), tlO, 24 * NE, hipMemcpyDeviceToHost); hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
    long long p0 = hP[0], pend = 0; for (int b = 0; b < NB; b++) { p0 = std::min(p0, hP[2 * b]); pend = std::max(pend, hP[2 * b + 1]); }
    std::vector<long long> pe(NB); for (int b = 0; b < NB; b++) pe[b] = hP[2 * b + 1]; std::sort(pe.begin(), pe.end());
    long long o0 = hO[0], oend = 0; int early = 0, early_done = 0; double life = 0, wait = 0;
    for (int b = 0; b < NE; b++) { o0 = std::min(o0, hO[2 * b]); oend = std::max(oend, hO[2 * b + 1]); early += hO[2 * b] < pend; early_done += hO[2 * b + 1] < pend; life += (hO[2 * b + 1] - hO[2 * NE + b]) / 100.0; wait += (hO[2 * NE + b] - hO[2 * b]) / 100.0; }
    printf("%-28s total %.1f us (events)  P: first start 0, median end %.1f, last end %.1f us | O: first start %.1f, last end %.1f us; O waves started / finished before P's last wave left: %d / %d; mean O wave life %.1f us, mean wait %.1f us; events O %.1f us; wrong data %u, timeouts %u\n",
           tag, 1e3 * msTot, (pe[NB / 2] - p0) / 100.0, (pend - p0) / 100.0, (o0 - p0) / 100.0, (oend - p0) / 100.0, early, early_done, life / NE, wait / NE, 1e3 * msO, hb[0], hb[1]);
    hipMemset(bad, 0, 8);
  };
  for (int rep = 0; rep < 3; rep++) {
    float ms, msO;
    // O alone (no queue)
    CK(hipEventRecord(e0, sA)); hipLaunchKernelGGL(kO<false>, dim3(NE), dim3(64), 0, sA, data, q, epoch, chain, lat, out, tlO, bad, 0, 0ll); CK(hipEventRecord(e1, sA)); CK(hipStreamSynchronize(sA));
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("O alone: %.1f us\n", 1e3 * ms); hipMemset(bad, 0, 8);
    // sequential: P then O on one stream, O through the queue (never waits)
    for (int atomic_loads = 0; atomic_loads < 2; atomic_loads++) {
      epoch++; CK(hipMemsetAsync(cnt, 0, 4, sA));
      CK(hipEventRecord(e0, sA));
      hipLaunchKernelGGL(kP, dim3(NB), dim3(64), 0, sA, data, q, cnt, epoch, tlP, base_us, spread_us);
      CK(hipEventRecord(eO0, sA));
      if (atomic_loads) hipLaunchKernelGGL(kO<true>, dim3(NE), dim3(64), 0, sA, data, q, epoch, chain, lat, out, tlO, bad, 1, 50000ll);
      else hipLaunchKernelGGL(kO<false>, dim3(NE), dim3(64), 0, sA, data, q, epoch, chain, lat, out, tlO, bad, 1, 50000ll);
      CK(hipEventRecord(e1, sA)); CK(hipStreamSynchronize(sA));
      CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&msO, eO0, e1));
      report(atomic_loads ? "sequential, atomic loads" : "sequential, plain loads", ms, msO);
    }
    // concurrent: P on stream A, O on stream B (enqueued after P; bounded spin 500 us), joined on A
    for (int atomic_loads = 0; atomic_loads < 2; atomic_loads++) {
      epoch++; CK(hipMemsetAsync(cnt, 0, 4, sA));
      CK(hipEventRecord(e0, sA));
      CK(hipEventRecord(eFork, sA)); CK(hipStreamWaitEvent(sB, eFork, 0));
      hipLaunchKernelGGL(kP, dim3(NB), dim3(64), 0, sA, data, q, cnt, epoch, tlP, base_us, spread_us);
      CK(hipEventRecord(eO0, sB));
      if (atomic_loads) hipLaunchKernelGGL(kO<true>, dim3(NE), dim3(64), 0, sB, data, q, epoch, chain, lat, out, tlO, bad, 1, 50000ll);
      else hipLaunchKernelGGL(kO<false>, dim3(NE), dim3(64), 0, sB, data, q, epoch, chain, lat, out, tlO, bad, 1, 50000ll);
      CK(hipEventRecord(eJoin, sB)); CK(hipStreamWaitEvent(sA, eJoin, 0));
      CK(hipEventRecord(e1, sA)); CK(hipStreamSynchronize(sA)); CK(hipStreamSynchronize(sB));
      CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&msO, eO0, eJoin));
      report(atomic_loads ? "concurrent, atomic loads" : "concurrent, plain loads", ms, msO);
    }
  }
  return 0;
}
