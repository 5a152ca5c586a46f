// Transposing wave reduction for gfx950: 36 fp64 values per lane -> 36 totals, one per lane (36 of the 64 lanes end up owning one).
// A plain butterfly pays 6 steps per VALUE (k_ba_schur's 36 sums: 36 x (6 x (2 v_mov_dpp + v_add_f64) + 2 v_readlane) ~ 800 instructions and
// 72 registers of partial sums in flight).  Here every step halves the number of values a lane carries: the lane keeps the half its
// partner does not, so the work is 18 + 9 + 5 + 3 + 2 + 1 = 38 additions, and the two widest steps use gfx950's
// v_permlane32_swap / v_permlane16_swap (upper half of one register <-> lower half of another: both operands of the addition land in
// place, no selects): ~160 instructions, the live values shrink 36 -> 18 -> 9 ...
// Summation order of one value over the lanes (fixed, the same for every value): (l, l^32), then ^16, ^8, ^1, ^2, ^4.
#pragma once
#include <hip/hip_runtime.h>

namespace orbhip {

// v_permlane32_swap: lanes 32..63 of a <-> lanes 0..31 of b
__device__ __forceinline__ void wr_swap32(double& a, double& b) {
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  const u2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const u2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi[0], (int)lo[0]); b = __hiloint2double((int)hi[1], (int)lo[1]);
}
// v_permlane16_swap: rows 1, 3 (lanes 16..31, 48..63) of a <-> rows 0, 2 of b
__device__ __forceinline__ void wr_swap16(double& a, double& b) {
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  const u2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const u2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi[0], (int)lo[0]); b = __hiloint2double((int)hi[1], (int)lo[1]);
}
template <int CTRL>
__device__ __forceinline__ double wr_dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// one step inside a row of 16 lanes: the lane whose `bit` is clear keeps a and receives its partner's a, the other keeps b
template <int CTRL>
__device__ __forceinline__ double wr_step(double a, double b, bool bit) {
  const double keep = bit ? b : a, send = bit ? a : b;
  return keep + wr_dpp<CTRL>(send);
}
// The value index (0..35) whose total this lane holds after wave_reduce36, or -1.
__device__ __forceinline__ int wave_reduce36_slot(int lane) {
  const int i5 = ((lane >> 2) & 1) + 2 * ((lane >> 1) & 1);
  const int i4 = i5 + 3 * (lane & 1);
  const int i3 = i4 + 5 * ((lane >> 3) & 1);
  const bool ok = i5 < 3 && i4 < 5 && i3 < 9;
  return ok ? i3 + 9 * ((lane >> 4) & 1) + 18 * ((lane >> 5) & 1) : -1;
}
__device__ __forceinline__ double wave_reduce36(double (&v)[36], int lane) {
  // lanes l, l ^ 32: the lower half keeps values 0..17, the upper 18..35
#pragma unroll
  for (int j = 0; j < 18; j++) { wr_swap32(v[j], v[18 + j]); v[j] += v[18 + j]; }
  // l ^ 16: rows 0, 2 keep 0..8 (of the 18), rows 1, 3 keep 9..17
#pragma unroll
  for (int j = 0; j < 9; j++) { wr_swap16(v[j], v[9 + j]); v[j] += v[9 + j]; }
  // l ^ 8 (row_ror:8): 9 -> 5 (+ one padding zero)
  const bool b3 = (lane >> 3) & 1, b0 = lane & 1, b1 = (lane >> 1) & 1, b2 = (lane >> 2) & 1;
  double y[6];
#pragma unroll
  for (int j = 0; j < 4; j++) y[j] = wr_step<0x128>(v[j], v[5 + j], b3);
  y[4] = wr_step<0x128>(v[4], 0.0, b3); y[5] = 0.0;
  // l ^ 1 (quad_perm [1,0,3,2]): 5 -> 3
  double z[4];
#pragma unroll
  for (int j = 0; j < 3; j++) z[j] = wr_step<0xB1>(y[j], y[3 + j], b0);
  z[3] = 0.0;
  // l ^ 2 (quad_perm [2,3,0,1]): 3 -> 2
  double q[2];
#pragma unroll
  for (int j = 0; j < 2; j++) q[j] = wr_step<0x4E>(z[j], z[2 + j], b1);
  // l ^ 4: 2 -> 1
  const double keep = b2 ? q[1] : q[0], send = b2 ? q[0] : q[1];
  return keep + __shfl_xor(send, 4);
}

}  // namespace orbhip
