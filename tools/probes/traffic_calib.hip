// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS library's access pattern (MI355X_MICROARCH.md, "HBM": the
// counters are exact only for some access widths - "calibrate on a known byte count in your own access pattern").  The step
// kernels read and write SoA rows with one 4-byte element per lane (a 256-byte line per wave and row); this kernel does exactly
// that over a known number of bytes, far beyond the L2 capacity.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/traffic_calib.hip -o alt_build/traffic_calib
//   rocprofv3 --pmc FETCH_SIZE -d out_f -o p --output-format csv -- alt_build/traffic_calib      (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void rows_rw(const float* __restrict__ in, float* __restrict__ out, int rows, long n) {
  const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (e >= n) return;
  for (int r = 0; r < rows; r++) out[r * n + e] = in[r * n + e] + 1.0f;
}
__global__ void rows_r(const float* __restrict__ in, float* __restrict__ out, int rows, long n) {
  const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (e >= n) return;
  float s = 0.f;
  for (int r = 0; r < rows; r++) s += in[r * n + e];
  if (s == 123.456f) out[e] = s;          // never true: read-only traffic
}
int main() {
  const long n = 1 << 20; const int rows = 64;           // 256 MiB read, 256 MiB written by rows_rw
  float *a, *b;
  hipMalloc(&a, rows * n * 4); hipMalloc(&b, rows * n * 4);
  hipMemset(a, 0, rows * n * 4); hipMemset(b, 0, rows * n * 4);
  hipDeviceSynchronize();
  for (int it = 0; it < 3; it++) {
    hipLaunchKernelGGL(rows_rw, dim3(n / 64), dim3(64), 0, 0, a, b, rows, n);
    hipLaunchKernelGGL(rows_r, dim3(n / 64), dim3(64), 0, 0, a, b, rows, n);
  }
  hipDeviceSynchronize();
  printf("rows_rw: %ld bytes read, %ld bytes written per launch; rows_r: %ld bytes read\n", rows * n * 4, rows * n * 4, rows * n * 4);
  return 0;
}
