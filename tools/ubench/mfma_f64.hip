// v_mfma_f64_16x16x4_f64 issue rate on gfx950: N independent accumulator chains per wave, W waves per SIMD, all CUs busy.
// Prints cycles per MFMA and SIMD (s_memtime, shader clock) and the TFLOP/s of the whole device.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/mfma_f64 tools/ubench/mfma_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* ticks, int reps) {
  double4_t acc[CH];
  for (int i = 0; i < CH; i++) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int i = 0; i < CH; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  double s = 0; for (int i = 0; i < CH; i++) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int CH> void run(int wgs, int threads, double* d, unsigned long long* dt, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  (void)0; k<CH><<<wgs, threads>>>(d, dt, reps); (void)hipDeviceSynchronize();
  hipEventRecord(e0); k<CH><<<wgs, threads>>>(d, dt, reps); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long t; hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
  const double nm = (double)reps * 8 * CH;                       // MFMAs per wave
  const double flops = nm * 2048.0 * wgs * (threads / 64);
  printf("chains %d  wgs %4d x %3d threads: %.1f ns per MFMA and wave (wave 0 of wg 0), device %.2f TFLOP/s (%.3f ms)\n", CH, wgs, threads, t * 10.0 / nm, flops / (ms * 1e-3) / 1e12, ms);
}
int main() {
  double* d; unsigned long long* dt; hipMalloc(&d, 8 * 256 * 4096); hipMalloc(&dt, 8 * 4096);
  const int reps = 2000;
  run<1>(256, 256, d, dt, reps); run<2>(256, 256, d, dt, reps); run<4>(256, 256, d, dt, reps); run<8>(256, 256, d, dt, reps);
  run<4>(512, 256, d, dt, reps); run<4>(256, 512, d, dt, reps); run<4>(1024, 256, d, dt, reps);
  run<4>(1, 64, d, dt, reps); run<4>(1, 256, d, dt, reps); run<4>(64, 256, d, dt, reps);
  return 0;
}
