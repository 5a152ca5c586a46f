// Phase timing of the 32-wide Cholesky panel step (diagonal factor / inverse / L21 on MFMA) with in-kernel
// s_memrealtime stamps (100 MHz).  Mirrors k_chol_panel of csrc/ba_solver.hip on a synthetic SPD block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define NB 32
typedef double double4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double bcast_lane(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane); hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rsqrt_f64(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}
__device__ __noinline__ int diag_factor_wave(double (*s_L)[NB + 1], double* s_dinv) {
  const int r = threadIdx.x & 31;
  double row[NB];
#pragma unroll
  for (int c = 0; c < NB; c++) row[c] = s_L[r][c];
  int fail = 0;
#pragma unroll
  for (int j = 0; j < NB; j++) {
    double piv = bcast_lane(row[j], j);
    if (!(piv > 0.0) || !isfinite(piv)) { fail = 1; piv = 1.0; }
    const double dinv = rsqrt_f64(piv);
    if (threadIdx.x == j) s_dinv[j] = dinv;
    row[j] = (r == j) ? piv * dinv : row[j] * dinv;
#pragma unroll
    for (int c = j + 1; c < NB; c++) {
      const double lcj = bcast_lane(row[j], c);
      row[c] = fma(-row[j], lcj, row[c]);
      if (((c - j) & 3) == 0) __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (threadIdx.x < NB) {
#pragma unroll
    for (int c = 0; c < NB; c++) s_L[r][c] = (c <= r) ? row[c] : 0.0;
  }
  return fail;
}
__device__ __noinline__ void diag_invert_wave(double (*s_L)[NB + 1], double (*s_X)[NB + 1], const double* s_dinv) {
  const int c = threadIdx.x & 31;
  double x[NB];
#pragma unroll
  for (int rr = 0; rr < NB; rr++) {
    double sum = (rr == c) ? 1.0 : 0.0;
#pragma unroll
    for (int m = 0; m < rr; m++) sum = fma(-s_L[rr][m], x[m], sum);
    x[rr] = sum * s_dinv[rr];
  }
  if (threadIdx.x < NB) {
#pragma unroll
    for (int rr = 0; rr < NB; rr++) s_X[rr][c] = x[rr];
  }
}

// ---- new variants: column broadcast through LDS (b128 broadcast reads), only the next-pivot term via v_readlane; blocked inverse
__device__ __noinline__ int diag_factor_wave2(double (*s_L)[NB + 1], double (*s_T)[NB], double* s_dinv) {
  const int r = threadIdx.x & 31;
  double row[NB];
#pragma unroll
  for (int c = 0; c < NB; c++) row[c] = s_L[r][c];
  int bad = 0;
#pragma unroll
  for (int j = 0; j < NB; j++) {
    const double piv = bcast_lane(row[j], j);
    bad |= (!(piv > 0.0) || !isfinite(piv)) ? 1 : 0;
    const double dinv = rsqrt_f64(piv);
    if (threadIdx.x == j) s_dinv[j] = dinv;
    row[j] = row[j] * dinv;                               // lane j: piv * dinv = sqrt(piv)
    s_T[j][r] = row[j];
    if (j + 1 < NB) { const double l = bcast_lane(row[j], j + 1); row[j + 1] = fma(-row[j], l, row[j + 1]); }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    int c = j + 2;
    if (c < NB && (c & 1)) { const double l = s_T[j][c]; row[c] = fma(-row[j], l, row[c]); c++; }
#pragma unroll
    for (; c + 1 < NB; c += 2) {
      const double2 l = *(const double2*)&s_T[j][c];
      row[c] = fma(-row[j], l.x, row[c]);
      row[c + 1] = fma(-row[j], l.y, row[c + 1]);
    }
  }
  if (threadIdx.x < NB) {
#pragma unroll
    for (int c = 0; c < NB; c++) s_L[r][c] = (c <= r) ? row[c] : 0.0;
  }
  return bad;
}
// X = L^-1 blockwise (16x16 blocks): X11, X22 by lanes (one column each), X21 = -X22 * (L21 * X11) on the matrix cores
__device__ __noinline__ void diag_invert_wave2(double (*s_L)[NB + 1], double (*s_X)[NB + 1], double (*s_T)[NB], const double* s_dinv) {
  const int lane = threadIdx.x & 63;
  const int c = lane & 15, b = lane & 16;
  {
    double x[16];
#pragma unroll
    for (int rr = 0; rr < 16; rr++) {
      double sum = (rr == c) ? 1.0 : 0.0;
#pragma unroll
      for (int m = 0; m < rr; m++) sum = fma(-s_L[b + rr][b + m], x[m], sum);
      x[rr] = sum * s_dinv[b + rr];
    }
    if (lane < 32) {
#pragma unroll
      for (int rr = 0; rr < 16; rr++) { s_X[b + rr][b + c] = x[rr]; if (b == 0) s_X[rr][16 + c] = 0.0; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
  const int li = lane & 15, lk = lane >> 4;
  double4_t t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < 4; ks++) t = __builtin_amdgcn_mfma_f64_16x16x4f64(s_L[16 + li][4 * ks + lk], s_X[4 * ks + lk][li], t, 0, 0, 0);
#pragma unroll
  for (int rg = 0; rg < 4; rg++) s_T[lk + 4 * rg][li] = t[rg];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
  double4_t u = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < 4; ks++) u = __builtin_amdgcn_mfma_f64_16x16x4f64(s_X[16 + li][16 + 4 * ks + lk], s_T[4 * ks + lk][li], u, 0, 0, 0);
#pragma unroll
  for (int rg = 0; rg < 4; rg++) s_X[16 + lk + 4 * rg][li] = -u[rg];
}
#define STAMP(i) do { if (tid == 0 && blockIdx.x == 0) stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
template <int V> __global__ __launch_bounds__(256) void k_panel(double* S, double* Dinv, int np, int k, unsigned long long* stamps) {
  __shared__ double s_L[NB][NB + 1];
  __shared__ double s_X[NB][NB + 1];
  __shared__ double s_dinv[NB];
  __shared__ __attribute__((aligned(16))) double s_T[NB][NB];
  __shared__ int s_fail;
  const int tid = threadIdx.x;
  STAMP(0);
  for (int i = tid; i < NB * NB; i += 256) { int r = i / NB, c = i % NB; s_L[r][c] = (c <= r) ? S[(size_t)(k + r) * np + k + c] : 0.0; }
  if (tid == 0) s_fail = 0;
  __syncthreads();
  STAMP(1);
  if (tid < 64) {
    const int fail = V ? diag_factor_wave2(s_L, s_T, s_dinv) : diag_factor_wave(s_L, s_dinv);
    if (fail && tid == 0) s_fail = 1;
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    STAMP(2);
    if (V) diag_invert_wave2(s_L, s_X, s_T, s_dinv); else diag_invert_wave(s_L, s_X, s_dinv);
  }
  __syncthreads();
  STAMP(3);
  if (blockIdx.x == 0) {
    double* Di = Dinv + (size_t)(k / NB) * NB * NB;
    for (int i = tid; i < NB * NB; i += 256) {
      int r = i / NB, c = i % NB;
      if (c <= r) S[(size_t)(k + r) * np + k + c] = s_L[r][c];
      Di[i] = s_X[r][c];
    }
  }
  STAMP(4);
  const int w = tid >> 6, lane = tid & 63;
  const int row0 = k + NB + (blockIdx.x * 4 + w) * 16;
  if (row0 > np) return;
  const int li = lane & 15, lk = lane >> 4;
  const int arow = row0 + li;
  const bool rvalid = arow <= np;
  double a[8];
#pragma unroll
  for (int ks = 0; ks < 8; ks++) a[ks] = rvalid ? S[(size_t)arow * np + k + 4 * ks + lk] : 0.0;
  double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < 8; ks++) {
    const double b0 = s_X[li][4 * ks + lk], b1 = s_X[16 + li][4 * ks + lk];
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b1, acc1, 0, 0, 0);
  }
#pragma unroll
  for (int rg = 0; rg < 4; rg++) {
    const int orow = row0 + (lane >> 4) + 4 * rg;
    if (orow <= np) {
      S[(size_t)orow * np + k + (lane & 15)] = acc0[rg];
      S[(size_t)orow * np + k + 16 + (lane & 15)] = acc1[rg];
    }
  }
  STAMP(5);
}
int main() {
  const int np = 608;
  std::vector<double> h((size_t)(np + 1) * np, 0.0);
  for (int i = 0; i < np; i++) for (int j = 0; j <= i; j++) h[(size_t)i * np + j] = (i == j) ? 40.0 + (i % 7) : std::cos(0.37 * i + 1.3 * j) / (1.0 + 0.3 * (i - j));
  double *S, *Di; unsigned long long* st;
  hipMalloc(&S, h.size() * 8); hipMalloc(&Di, 19 * NB * NB * 8); hipMalloc(&st, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<double> out[2], dout[2];
  for (int v = 0; v < 2; v++) {
    for (int rep = 0; rep < 3; rep++) {
      hipMemcpy(S, h.data(), h.size() * 8, hipMemcpyHostToDevice);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      if (v) k_panel<1><<<(np - NB + 1 + 63) / 64, 256>>>(S, Di, np, 0, st); else k_panel<0><<<(np - NB + 1 + 63) / 64, 256>>>(S, Di, np, 0, st);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long s[6]; hipMemcpy(s, st, 48, hipMemcpyDeviceToHost);
      printf("v%d launch %.1f us | load %.2f  factor %.2f  invert %.2f  store %.2f  L21 %.2f us (100 MHz stamps)\n", v, ms * 1e3,
             (s[1] - s[0]) * 0.01, (s[2] - s[1]) * 0.01, (s[3] - s[2]) * 0.01, (s[4] - s[3]) * 0.01, (s[5] - s[4]) * 0.01);
    }
    out[v].resize(h.size()); dout[v].resize(NB * NB);
    hipMemcpy(out[v].data(), S, h.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(dout[v].data(), Di, NB * NB * 8, hipMemcpyDeviceToHost);
  }
  double dS = 0, dD = 0, mS = 0;
  for (size_t i = 0; i < h.size(); i++) { dS = std::fmax(dS, std::fabs(out[0][i] - out[1][i])); mS = std::fmax(mS, std::fabs(out[0][i])); }
  for (int i = 0; i < NB * NB; i++) dD = std::fmax(dD, std::fabs(dout[0][i] - dout[1][i]));
  printf("max |S0 - S1| = %.3e (max |S| %.3e), max |Dinv0 - Dinv1| = %.3e\n", dS, mS, dD);
  return 0;
}
