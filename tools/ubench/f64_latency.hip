// Single-wave latency of the operations on the critical path of the 32x32 diagonal factor (gfx950), in ns and in units of a
// dependent v_add_u32 (4 cycles on a 16-lane SIMD): dependent and independent v_fma_f64, v_rsq_f64, the v_readlane -> VALU
// round trip, an LDS write -> broadcast read round trip, ds_read_b128 issue.  Timed with s_memrealtime (100 MHz) over N
// repetitions of an unrolled block; the kernel runs behind a spin kernel so that the clock is the loaded one.
//   hipcc --offload-arch=gfx950 -O3 -o f64_latency f64_latency.hip && ./f64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define REP 64
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)
__global__ void k_spin(double* out, int n) {
  double a = threadIdx.x * 1e-3, b = 1.0000001;
  for (int i = 0; i < n; i++) { a = fma(a, b, 1e-9); b = fma(b, 0.9999999, 1e-9); }
  if (a == 12345.678) out[0] = a + b;
}
__global__ void k_acc(double* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  unsigned long long z = 0x9E3779B97F4A7C15ull * (i + 1); z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
  const double x = ldexp(1.0 + (double)(z >> 12) * (1.0 / 4503599627370496.0), (int)(z & 63) - 32);
  out[i] = x; out[65536 + i] = __builtin_amdgcn_rcp(x); out[2 * 65536 + i] = __builtin_amdgcn_rsq(x);
}
template <int T>
__global__ __launch_bounds__(64) void k_lat(double* out, unsigned long long* ticks) {
  __shared__ __attribute__((aligned(16))) double s_buf[128];
  const int lane = threadIdx.x;
  double a = 1.0 + lane * 1e-3, b = 0.999999, c = 1e-9, d0 = a, d1 = a + 1, d2 = a + 2, d3 = a + 3;
  unsigned u = lane, v = 3;
  s_buf[lane] = a; s_buf[64 + lane] = b;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int rep = 0; rep < REP; rep++) {
    if (T == 0) { asm volatile(R64("v_add_u32 %0, %0, %1\n") : "+v"(u) : "v"(v)); }                                  // dependent int add
    if (T == 1) { asm volatile(R64("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c)); }                        // dependent fma
    if (T == 2) { asm volatile(R16("v_fma_f64 %0, %0, %4, %5\nv_fma_f64 %1, %1, %4, %5\nv_fma_f64 %2, %2, %4, %5\nv_fma_f64 %3, %3, %4, %5\n")
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c)); }                            // 4 independent chains
    if (T == 3) { asm volatile(R64("v_rsq_f64 %0, %0\ns_nop 1\n") : "+v"(a)); }                                        // dependent rsq (+2 wait states)
    if (T == 4) { asm volatile(R64("v_readlane_b32 s20, %0, 5\ns_nop 3\nv_add_u32 %0, s20, %0\n") : "+v"(u) : : "s20"); }   // readlane -> use -> readlane
    if (T == 5) { asm volatile(R64("ds_write_b64 %1, %0\ns_waitcnt lgkmcnt(0)\nds_read_b64 %0, %2\ns_waitcnt lgkmcnt(0)\n") : "+v"(a) : "v"(lane * 8), "v"(40)); }   // LDS write -> broadcast read
    if (T == 6) { typedef double d2_t __attribute__((ext_vector_type(2))); d2_t q; asm volatile(R64("ds_read_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(q) : "v"(64)); a += q.x; }                    // b128 broadcast read issue
    if (T == 7) { asm volatile(R64("v_mul_f64 %0, %0, %1\n") : "+v"(a) : "v"(b)); }                                   // dependent mul
    if (T == 8) { asm volatile(R64("v_readlane_b32 s20, %0, 5\nv_readlane_b32 s21, %1, 5\n") : : "v"(u), "v"(v) : "s20", "s21"); }   // readlane issue only
    if (T == 9) { asm volatile(R64("v_fma_f64 %0, %0, %1, %2\ns_nop 0\n") : "+v"(a) : "v"(b), "v"(c)); }              // dependent fma with an s_nop between
    if (T == 10) { asm volatile(R64("v_rcp_f64 %0, %0\ns_nop 1\n") : "+v"(a)); }                                       // dependent rcp (+2 wait states)
    if (T == 11) { asm volatile(R64("v_rcp_f64 %0, %0\nv_fma_f64 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c)); }      // rcp -> fma -> rcp ... (per pair)
    if (T == 12) { asm volatile(R64("v_readlane_b32 s20, %0, 5\nv_readlane_b32 s21, %1, 5\nv_fma_f64 %2, s[20:21], %3, %2\nv_cvt_u32_f64 %0, %2\n") : "+v"(u), "+v"(v), "+v"(a) : "v"(b) : "s20", "s21"); }   // 2 readlanes -> fma with the SGPR pair -> feeds the next readlane
    if (T == 13) { typedef double d4_t __attribute__((ext_vector_type(4))); d4_t q = {a, a, a, a}; asm volatile(R16("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n") : "+v"(q) : "v"(b), "v"(c)); a = q.x; }      // dependent 16x16x4 (16 per rep)
    if (T == 14) { asm volatile(R64("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n") : "+v"(a) : "v"(b), "v"(c)); }          // dependent 4x4x4 (4 blocks)
    if (T == 15) { asm volatile(R64("v_rsq_f64 %0, %0\nv_fma_f64 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c)); }      // rsq -> fma -> rsq ... (per pair)
    if (T == 16) { asm volatile(R64("v_mov_b32_dpp %0, %0 row_shr:1\n") : "+v"(u)); }                                    // dependent DPP move
    if (T == 17) { asm volatile(R64("v_fma_f64 %0, %0, %2, %3\nv_fma_f64 %1, %1, %2, %3\n") : "+v"(a), "+v"(d0) : "v"(b), "v"(c)); }   // 2 independent chains (per pair)
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) ticks[0] = t1 - t0;
  out[lane] = a + d0 + d1 + d2 + d3 + u;
}
template <int T> static double run(const char* name, double* d, unsigned long long* dt, double per_block, double base) {
  for (int w = 0; w < 2; w++) { k_spin<<<2048, 256>>>(d, 300000); k_lat<T><<<1, 64>>>(d, dt); hipDeviceSynchronize(); }
  unsigned long long t; hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
  const double ns = t * 10.0 / (REP * per_block);
  printf("%-44s %7.2f ns per op", name, ns);
  if (base > 0) printf("   = %5.1f cycles (4-cycle int add = %.2f ns)", ns / base * 4.0, base);
  printf("\n");
  return ns;
}
int main() {
  double* d; unsigned long long* dt; hipMalloc(&d, 4096); hipMalloc(&dt, 64);
  const double base = run<0>("dependent v_add_u32", d, dt, 64, 0);
  run<1>("dependent v_fma_f64", d, dt, 64, base);
  run<2>("v_fma_f64, 4 independent chains (per op)", d, dt, 64, base);
  run<7>("dependent v_mul_f64", d, dt, 64, base);
  run<9>("dependent v_fma_f64 + s_nop 0", d, dt, 64, base);
  run<3>("dependent v_rsq_f64 (+ s_nop 1)", d, dt, 64, base);
  run<4>("v_readlane -> s_nop 3 -> v_add (round trip)", d, dt, 64, base);
  run<8>("v_readlane_b32 issue (per op)", d, dt, 128, base);
  run<5>("LDS write -> wait -> broadcast read -> wait", d, dt, 64, base);
  run<6>("ds_read_b128 broadcast issue (per op)", d, dt, 64, base);
  run<10>("dependent v_rcp_f64 (+ s_nop 1)", d, dt, 64, base);
  run<11>("v_rcp_f64 -> v_fma_f64 (per pair)", d, dt, 64, base);
  run<15>("v_rsq_f64 -> v_fma_f64 (per pair)", d, dt, 64, base);
  run<12>("2 x v_readlane -> v_fma_f64 s[..] -> v_mov (per round)", d, dt, 64, base);
  run<13>("dependent v_mfma_f64_16x16x4_f64", d, dt, 16, base);
  run<14>("dependent v_mfma_f64_4x4x4_4b_f64", d, dt, 64, base);
  run<16>("dependent v_mov_b32_dpp row_shr:1", d, dt, 64, base);
  run<17>("v_fma_f64, 2 independent chains (per pair)", d, dt, 64, base);
  // accuracy of the hardware estimates
  {
    double* dx; hipMalloc(&dx, 3 * 65536 * 8);
    k_acc<<<256, 256>>>(dx);
    std::vector<double> h(3 * 65536); hipMemcpy(h.data(), dx, h.size() * 8, hipMemcpyDeviceToHost);
    double mr = 0, ms = 0;
    for (int i = 0; i < 65536; i++) { const long double x = h[i]; mr = fmax(mr, fabs((double)((long double)h[65536 + i] * x - 1.0L))); ms = fmax(ms, fabs((double)((long double)h[2 * 65536 + i] * (long double)h[2 * 65536 + i] * x - 1.0L))); }
    printf("v_rcp_f64 max |r x - 1| = %.3e (2^%.1f); v_rsq_f64 max |y^2 x - 1| = %.3e (2^%.1f)\n", mr, log2(mr), ms, log2(ms));
  }
  return 0;
}
