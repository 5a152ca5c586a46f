// Issue rate of v_mfma_i32_16x16x64_i8 on gfx950: independent accumulators (throughput) and one dependent chain (latency),
// 1 / 2 / 4 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_i8.hip -o mfma_i8 && ./mfma_i8
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512) void k(int* out, int iters) {
  v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
  v4i c[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++) c[i] = (v4i){i, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < NACC; i++) s += c[i].x + c[i].y + c[i].z + c[i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int wg_threads, int nwg, const char* name) {
  int* d; hipMalloc(&d, (size_t)nwg * wg_threads * 4);
  const int iters = 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 20; w++) hipLaunchKernelGGL(k<NACC>, dim3(nwg), dim3(wg_threads), 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(nwg), dim3(wg_threads), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)nwg * (wg_threads / 64) * iters * NACC;
  const double waves_per_simd = (double)nwg * (wg_threads / 64) / 1024.0;
  printf("%-34s %8.3f ms  %7.1f TOPS  %6.1f cycles per MFMA and SIMD at 2.4 GHz\n", name, ms, n * 32768.0 / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (n / 1024.0));
  (void)waves_per_simd;
  hipFree(d);
}
int main() {
  run<2>(256, 512, "2 chains, 2 waves per SIMD");
  run<8>(256, 256, "8 independent, 1 wave per SIMD");
  run<8>(256, 512, "8 independent, 2 waves per SIMD");
  run<4>(256, 512, "4 independent, 2 waves per SIMD");
  run<1>(256, 256, "1 dependent chain, 1 wave per SIMD");
  run<1>(256, 1024, "1 dependent chain, 4 waves per SIMD");
  run<2>(256, 512, "2 chains, 2 waves per SIMD");
  run<4>(512, 256, "4 chains, 2 waves per SIMD, 512-thread workgroups");
  run<16>(512, 256, "16 independent, 2 waves per SIMD, 512-thread workgroups");
  return 0;
}
