// Do VALU instructions co-issue with v_mfma_i32_16x16x64_i8 on gfx950?  NV independent v_min / v_med3 per MFMA, 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 mfma_valu_mix.hip -o mfma_valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int NV, bool MF>
__global__ __launch_bounds__(512) void k(int* out, int iters) {
  v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
  v4i c[4];
  int k1[8], k2[8];
#pragma unroll
  for (int i = 0; i < 4; i++) c[i] = (v4i){i, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 8; i++) { k1[i] = 1000 + i + threadIdx.x; k2[i] = 2000 + i; }
  int x = threadIdx.x * 7;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (MF) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; v++) {
        const int j = (i * NV + v) & 7;
        x = x * 5 + 1;                                   // (a value the compiler cannot fold; 1 more VALU)
        int r;
        asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(k1[j]), "v"(k2[j]), "v"(x));
        k2[j] = r;
        k1[j] = min(k1[j], x);
      }
    }
  }
  int s = x;
#pragma unroll
  for (int i = 0; i < 4; i++) s += c[i].x + c[i].y + c[i].z + c[i].w;
#pragma unroll
  for (int i = 0; i < 8; i++) s += k1[i] + k2[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, bool MF, bool CH1, bool RD = false>
__global__ __launch_bounds__(512) void k32(int* out, int iters) {
  v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
  v16i c[2];
  int k1[8], k2[8];
#pragma unroll
  for (int i = 0; i < 2; i++) for (int e = 0; e < 16; e++) c[i][e] = i + e;
#pragma unroll
  for (int i = 0; i < 8; i++) { k1[i] = 1000 + i + threadIdx.x; k2[i] = 2000 + i; }
  int x = threadIdx.x * 7;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      if (MF) c[CH1 ? 0 : i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[CH1 ? 0 : i], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; v++) {
        const int j = (i * NV + v) & 7;
        if (RD) x = c[1 - i][(v * 2) & 15]; else x = x * 5 + 1;      // RD: read a result of the accumulator that is NOT being accumulated
        int r;
        asm volatile("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(k1[j]), "v"(k2[j]), "v"(x));
        k2[j] = r;
        k1[j] = min(k1[j], x);
      }
    }
  }
  int s = x;
#pragma unroll
  for (int i = 0; i < 2; i++) for (int e = 0; e < 16; e++) s += c[i][e];
#pragma unroll
  for (int i = 0; i < 8; i++) s += k1[i] + k2[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, bool MF, bool CH1 = false, bool RD = false>
void run32(const char* name) {
  const int nwg = 256, th = 512, iters = 2048;
  int* d; (void)hipMalloc(&d, (size_t)nwg * th * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 20; w++) hipLaunchKernelGGL((k32<NV, MF, CH1, RD>), dim3(nwg), dim3(th), 0, 0, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k32<NV, MF, CH1, RD>), dim3(nwg), dim3(th), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double slots = (double)nwg * (th / 64) * iters * 2 / 1024.0;
  printf("32x32x32: %-34s %7.3f ms  %6.1f cycles per slot and SIMD (each slot = 1 MFMA + %d x 3 VALU)\n", name, ms, ms * 1e-3 * 2.4e9 / slots, NV);
  (void)hipFree(d);
}
template <int NV, bool MF>
void run(const char* name) {
  const int nwg = 256, th = 512, iters = 2048;
  int* d; (void)hipMalloc(&d, (size_t)nwg * th * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 20; w++) hipLaunchKernelGGL((k<NV, MF>), dim3(nwg), dim3(th), 0, 0, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, MF>), dim3(nwg), dim3(th), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double slots = (double)nwg * (th / 64) * iters * 4 / 1024.0;     // MFMA slots per SIMD
  printf("%-44s %7.3f ms  %6.1f cycles per slot and SIMD (2 waves: each slot = 1 MFMA + %d x 3 VALU)\n", name, ms, ms * 1e-3 * 2.4e9 / slots, NV);
  (void)hipFree(d);
}
int main() {
  run<0, true>("MFMA only");
  run<1, true>("MFMA + 3 VALU");
  run<2, true>("MFMA + 6 VALU");
  run<2, false>("6 VALU only");
  run<4, true>("MFMA + 12 VALU");
  run<4, false>("12 VALU only");
  run32<0, true>("MFMA only");
  run32<2, true>("MFMA + 6 VALU");
  run32<4, true>("MFMA + 12 VALU");
  run32<8, true>("MFMA + 24 VALU");
  run32<8, false>("24 VALU only");
  run32<0, true, true>("MFMA only, ONE dependent chain");
  run32<2, true, true>("MFMA + 6 VALU, ONE dependent chain");
  run32<4, true, false, true>("MFMA + 8 VALU reading the other accumulator");
  run32<4, true, false, false>("MFMA + 12 VALU not reading results");
  return 0;
}
