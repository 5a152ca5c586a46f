// How fast can u8 image tiles be streamed on MI355X?  Copies 256 frames of 1241x376 (stride 1241 and stride 1280) with
// different workgroup tile shapes and per-lane access widths; reports algorithmic GB/s (read + write).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// tile TW x TH, 256 threads, each lane moves VEC bytes per access
template <int TW, int TH, int VEC>
__global__ __launch_bounds__(256) void k_tile(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int w, int h, int pitch, int tiles_x) {
  const int f = blockIdx.y, t = blockIdx.x, tx = t % tiles_x, ty = t / tiles_x;
  constexpr int LPR = TW / VEC;            // lanes per row
  constexpr int RPP = 256 / LPR;           // rows per pass
  const int lx = threadIdx.x % LPR, ly = threadIdx.x / LPR;
  const int x = tx * TW + lx * VEC;
  const size_t fo = (size_t)f * pitch * h;
  if (x + VEC > w) return;
  typedef uint32_t vec_t __attribute__((ext_vector_type(VEC / 4)));
  vec_t v[TH / RPP];
#pragma unroll
  for (int k = 0; k < TH / RPP; k++) {
    const int y = ty * TH + ly + k * RPP;
    if (y < h) __builtin_memcpy(&v[k], src + fo + (size_t)y * pitch + x, VEC);     // unaligned-safe
  }
#pragma unroll
  for (int k = 0; k < TH / RPP; k++) {
    const int y = ty * TH + ly + k * RPP;
    if (y < h) __builtin_memcpy(dst + fo + (size_t)y * pitch + x, &v[k], VEC);
  }
}
template <int TW, int TH, int VEC> void run(const uint8_t* s, uint8_t* d, int w, int h, int pitch, int nf) {
  const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + TH - 1) / TH;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; i++) k_tile<TW, TH, VEC><<<dim3(tiles_x * tiles_y, nf), 256>>>(s, d, w, h, pitch, tiles_x);
  hipEventRecord(e0);
  for (int i = 0; i < 5; i++) k_tile<TW, TH, VEC><<<dim3(tiles_x * tiles_y, nf), 256>>>(s, d, w, h, pitch, tiles_x);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("pitch %4d tile %4dx%-3d vec %2d B : %.3f ms  %.0f GB/s\n", pitch, TW, TH, VEC, ms, 2.0 * w * h * nf / ms * 1e-6);
}
int main() {
  const int w = 1241, h = 376, nf = 256;
  for (int pitch : {1241, 1280}) {
    uint8_t *s, *d; size_t n = (size_t)pitch * h * nf + 64;
    hipMalloc(&s, n); hipMalloc(&d, n); hipMemset(s, 1, n);
    run<128, 32, 4>(s, d, w, h, pitch, nf);
    run<128, 64, 4>(s, d, w, h, pitch, nf);
    run<256, 32, 4>(s, d, w, h, pitch, nf);
    run<256, 32, 8>(s, d, w, h, pitch, nf);
    run<512, 32, 8>(s, d, w, h, pitch, nf);
    run<512, 32, 16>(s, d, w, h, pitch, nf);
    run<1024, 16, 16>(s, d, w, h, pitch, nf);
    run<1024, 32, 16>(s, d, w, h, pitch, nf);
    run<256, 64, 16>(s, d, w, h, pitch, nf);
    hipFree(s); hipFree(d);
  }
  return 0;
}
