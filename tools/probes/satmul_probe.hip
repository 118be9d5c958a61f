// probe: does v_pk_mul_f32 ... clamp saturate to [0, 1] on gfx950, and is clamp((-2^64 x) * 2^100) == (x < 0 ? 1 : 0) for every float class?
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/satmul_probe.hip -o /tmp/satmul && /tmp/satmul
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ inline f2 sat_mul(f2 a, f2 b) { f2 r; asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b)); return r; }
__global__ void k(const float* x, float* y, int n) {
  int i = threadIdx.x + blockIdx.x * blockDim.x;
  if (i >= n) return;
  f2 t = f2{x[i], x[i]} * f2{-0x1p64f, -0x1p64f};
  f2 m = sat_mul(t, f2{0x1p100f, 0x1p100f});
  y[i] = m.x;
}
int main() {
  const int n = 1 << 20;
  float* h = new float[n]; float* o = new float[n];
  unsigned s = 12345u;
  for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; unsigned b = s; if (i % 4 == 0) b &= 0x807fffffu; /* denormals / zeros */ memcpy(&h[i], &b, 4); }
  float special[] = {-1e-45f, -1e-38f, -1.f, -0.f, 0.f, 1e-45f, 1.f, NAN, INFINITY, -INFINITY, -3e38f, 3e38f, -1e-20f, 1e-20f};
  for (int i = 0; i < 14; i++) h[i] = special[i];
  float *dx, *dy; hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4);
  hipMemcpy(dx, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
  hipMemcpy(o, dy, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; i++) { float want = h[i] < 0.f ? 1.f : 0.f; if (!(o[i] == want)) { if (bad < 10) printf("x=%g (%08x) got %g want %g\n", h[i], *(unsigned*)&h[i], o[i], want); bad++; } }
  for (int i = 0; i < 14; i++) printf("x=%g -> %g\n", h[i], o[i]);
  printf("mismatches: %d of %d\n", bad, n);
  return bad != 0;
}
