// VALU issue-rate microbenchmark (gfx950): lane-ops/s of plain integer, packed 16-bit, dot4/dot2 and mad_u32_u16 instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
template <int OP> __global__ __launch_bounds__(256) void k(unsigned* out, unsigned seed) {
  unsigned a[8];
  for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 8 + i;
  unsigned b = seed ^ 0x9e3779b9u;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) a[i] = a[i] * 3u + b;                                   // v_mad_u32_u24? (32-bit mul: slower) -> use add/xor
      if (OP == 1) a[i] = (a[i] ^ b) + a[(i + 1) & 7];                      // 2 plain VALU (xor, add) [may fuse to v_xad_u32]
      if (OP == 2) { ushort2_t x = __builtin_bit_cast(ushort2_t, a[i]), y = __builtin_bit_cast(ushort2_t, b); x = __builtin_elementwise_min(x, y) + y; a[i] = __builtin_bit_cast(unsigned, x); }   // v_pk_min_u16 + v_pk_add_u16
      if (OP == 3) a[i] = __builtin_amdgcn_udot4(a[i], b, a[(i + 1) & 7], false);   // v_dot4_u32_u8
      if (OP == 4) a[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, a[i]), __builtin_bit_cast(ushort2_t, b), a[(i + 1) & 7], false);   // v_dot2_u32_u16
      if (OP == 5) a[i] = __builtin_amdgcn_alignbyte(a[i], a[(i + 1) & 7], 1);     // v_alignbyte_b32
      if (OP == 6) a[i] = __builtin_amdgcn_sad_u8(a[i], b, a[(i + 1) & 7]);        // v_sad_u8
      if (OP == 7) a[i] = __builtin_popcount(a[i] ^ b) + a[(i + 1) & 7];            // v_xor + v_bcnt_u32_b32 (with add)
    }
  }
  unsigned s = 0;
  for (int i = 0; i < 8; i++) s ^= a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char* name, double instr_per_step) {
  unsigned* d; hipMalloc(&d, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<4096, 256>>>(d, 1); hipDeviceSynchronize();
  hipEventRecord(e0); k<OP><<<4096, 256>>>(d, 2); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double lane_instr = 4096.0 * 256 * ITER * 8 * instr_per_step;
  printf("%-28s %.3f ms  %.1f T lane-instr/s (assuming %.0f instr per step)\n", name, ms, lane_instr / ms * 1e-9, instr_per_step);
  hipFree(d);
}
int main() {
  run<0>("mul_lo+add (u32 mad)", 1);
  run<1>("xor+add", 2);
  run<2>("pk_min_u16+pk_add_u16", 2);
  run<3>("dot4_u32_u8", 1);
  run<4>("dot2_u32_u16", 1);
  run<5>("alignbyte", 1);
  run<6>("sad_u8", 1);
  run<7>("xor+bcnt(add)", 2);
  return 0;
}
