// One wave factoring and inverting a 32x32 SPD block out of LDS: variants of diag_factor_wave / diag_invert_wave of
// csrc/ba_solver.hip, timed with in-kernel s_memrealtime stamps (100 MHz) over REPS repetitions, and compared with each other
// (the factor variants must agree BIT FOR BIT: same operations per element in the same order, only the broadcast path differs).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o diag_factor diag_factor.hip && ./diag_factor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <cmath>
#define NB 32
#define REPS 64
typedef double double4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double bcast_lane(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane); hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rsqrt_f64(double x) {
  const double hx = 0.5 * x;
  double y = __builtin_amdgcn_rsq(x);
  double r = fma(-(hx * y), y, 0.5);
  y = fma(y, r, y);
  r = fma(-(hx * y), y, 0.5);
  y = fma(y, r, y);
  return y;
}
// AHEAD = how many of the next columns get their l(c, j) by v_readlane (the rest come back from LDS as broadcast reads)
template <int AHEAD>
__device__ __noinline__ int diag_factor_wave(double (*s_L)[NB + 1], double (*s_T)[NB], double* s_dinv) {
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
    row[j] = row[j] * dinv;
    s_T[j][r] = row[j];
#pragma unroll
    for (int a = 1; a <= AHEAD; a++)
      if (j + a < NB) { const double l = bcast_lane(row[j], j + a); row[j + a] = fma(-row[j], l, row[j + a]); }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    int c = j + 1 + AHEAD;
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

// PIPELINED: the compiler schedules the rank-1 updates of the variants above LAZILY - row[c] collects its c pending updates
// as one dependent FMA chain right before column c's pivot is read (a j-deep chain in the critical path of column j).  Here
// the order is pinned with sched_barriers: the LDS-borne part of column j-1's update (c >= j + 1) is issued in the latency
// shadow of column j's rsqrt chain, U of them per chain operation; the pivot row's own entry still travels by v_readlane.
#define SB() __builtin_amdgcn_sched_barrier(0)
// an empty volatile asm with the value as in/out operand: the value must exist BEFORE this point and later uses wait for it, so a
// pure operation cannot drift across the following sched_barrier at IR level (sched_barrier only pins the machine scheduler)
#define PIN(x) asm volatile("" : "+v"(x))
template <int J, int C0, int N>
__device__ __forceinline__ void pend_upd(double (&row)[NB], const double (&lp)[NB], const double rp) {   // column J-1 applied to columns C0 .. C0+N-1
  if constexpr (J > 0) {
#pragma unroll
    for (int q = 0; q < N; q++) if (C0 + q < NB) row[C0 + q] = fma(-rp, lp[C0 + q], row[C0 + q]);
  }
}
template <int J, int U, int SKIP>
struct FactorCol {
  static __device__ __forceinline__ void run(double (&row)[NB], double (&lp)[NB], double& rp, int& bad, double& dsave, double (*s_T)[NB], const int r) {
    constexpr int c0 = J + 1;
    constexpr int u0 = SKIP > 0 ? 0 : U, u1 = SKIP > 1 ? 0 : U;
    const double piv = bcast_lane(row[J], J);
    bad |= (!(piv > 0.0) || !isfinite(piv)) ? 1 : 0;
    const double hx = 0.5 * piv;
    double y = __builtin_amdgcn_rsq(piv);
    SB(); pend_upd<J, c0, u0>(row, lp, rp); SB();
    double t = hx * y;
    SB(); pend_upd<J, c0 + u0, u1>(row, lp, rp); SB();
    double e = fma(-t, y, 0.5);
    SB(); pend_upd<J, c0 + u0 + u1, U>(row, lp, rp); SB();
    y = fma(y, e, y);
    SB(); pend_upd<J, c0 + u0 + u1 + U, U>(row, lp, rp); SB();
    t = hx * y;
    SB(); pend_upd<J, c0 + u0 + u1 + 2 * U, U>(row, lp, rp); SB();
    e = fma(-t, y, 0.5);
    SB(); pend_upd<J, c0 + u0 + u1 + 3 * U, U>(row, lp, rp); SB();
    y = fma(y, e, y);
    SB(); pend_upd<J, c0 + u0 + u1 + 4 * U, U>(row, lp, rp); SB();
    dsave = (r == J) ? y : dsave;                         // (a conditional LDS store here would split the basic block: the
                                                          //  code sinker then moves the pending updates behind the whole chain)
    row[J] = row[J] * y;
    SB(); pend_upd<J, c0 + u0 + u1 + 5 * U, NB>(row, lp, rp); SB();           // whatever is left
    s_T[J][r] = row[J];
    if constexpr (J + 1 < NB) { const double l = bcast_lane(row[J], J + 1); row[J + 1] = fma(-row[J], l, row[J + 1]); }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    rp = row[J];
    if constexpr (J + 2 < NB) {
      constexpr int ce = (J + 2) + ((J + 2) & 1);                              // first even column >= J + 2
      if constexpr (((J + 2) & 1) != 0) lp[J + 2] = s_T[J][J + 2];
#pragma unroll
      for (int c = ce; c + 1 < NB; c += 2) { const double2 l = *(const double2*)&s_T[J][c]; lp[c] = l.x; lp[c + 1] = l.y; }
    }
    SB();
    if constexpr (J + 1 < NB) FactorCol<J + 1, U, SKIP>::run(row, lp, rp, bad, dsave, s_T, r);
  }
};
template <int U, int SKIP>
__device__ __noinline__ int diag_factor_pipe(double (*s_L)[NB + 1], double (*s_T)[NB], double* s_dinv) {
  const int r = threadIdx.x & 31;
  double row[NB], lp[NB];
#pragma unroll
  for (int c = 0; c < NB; c++) { row[c] = s_L[r][c]; lp[c] = 0.0; }
  int bad = 0;
  double rp = 0.0;                                          // row[j - 1] (scaled): this lane's multiplier of the pending update
  double dsave = 0.0;
  FactorCol<0, U, SKIP>::run(row, lp, rp, bad, dsave, s_T, r);
  if (threadIdx.x < NB) {
    s_dinv[r] = dsave;
#pragma unroll
    for (int c = 0; c < NB; c++) s_L[r][c] = (c <= r) ? row[c] : 0.0;
  }
  return bad;
}
// FUSED factor + inverse.  Inverting L by forward substitution applies to the columns of I exactly the operations the
// right-looking factor applies to the rows of A (scale entry j by 1/sqrt(pivot j), subtract l(c, j) times it from entry c > j),
// so lanes 32..63 - idle in the factor - carry one column of the identity each through the SAME instructions and end up holding
// L^-1: no separate inverse phase.  The loop-carried chain is kept off the lanes: the next pivot is formed from two values read
// ahead of time (sa = a(j+1, j), sb = a(j+1, j+1), uniform), pivot' = sb - (sa y)^2, bit-identical to what lane j+1 computes.
template <int J, int C0, int N>
__device__ __forceinline__ void cur_upd(double (&acc)[NB], const double (&lp)[NB]) {                   // column J applied to columns C0 .. C0+N-1
#pragma unroll
  for (int q = 0; q < N; q++) if (C0 + q < NB) acc[C0 + q] = fma(-acc[J], lp[C0 + q], acc[C0 + q]);
}
template <int J, int C0, int N, bool OFF = false>
__device__ __forceinline__ void prev_upd(double (&acc)[NB], const double (&lp)[NB], const double mp) {     // column J-1 applied to columns C0 .. C0+N-1
  if constexpr (J > 0 && !OFF) {
#pragma unroll
    for (int q = 0; q < N; q++) if (C0 + q < NB) acc[C0 + q] = fma(-mp, lp[C0 + q], acc[C0 + q]);
  }
}
// body J: y = 1/sqrt(pivot J); sa = a(J+1, J), sb = a(J+1, J+1) (uniform); lp[c] = l(c, J-1) for c >= J+2, requested from LDS
// one body earlier (a 105-cycle round trip, f64_latency.hip: consumed in the same body it would stall the in-order wave), mp =
// this lane's scaled entry J-1.  Columns J+1 and J+2 get column J's update through v_readlane, the rest through LDS one body later.
template <int J, int U, int MODE>
struct FusedCol {
  static constexpr bool CUBIC = (MODE & 1) != 0, NOLDS = (MODE & 2) != 0, NORL = (MODE & 4) != 0, NOFMA = (MODE & 8) != 0;
  static __device__ __forceinline__ void run(double (&acc)[NB], const double (&lp)[NB], const double mp, const double y, const double sa, const double sb,
                                             int& bad, double (*s_T)[64], const int lane) {
    double l = 0.0, pivn = 1.0, hxn = 0.0, yn = 0.0;
    if constexpr (J + 1 < NB) {
      l = sa * y;
      pivn = fma(-l, l, sb);
      hxn = 0.5 * pivn;
      yn = __builtin_amdgcn_rsq(pivn);
      PIN(yn); PIN(hxn);
      bad |= (!(pivn > 0.0) || !isfinite(pivn)) ? 1 : 0;
    }
    SB();
    acc[J] = acc[J] * y;
    if constexpr (!NOLDS) s_T[J][lane] = acc[J];
    if constexpr (J + 1 < NB) acc[J + 1] = fma(-acc[J], l, acc[J + 1]);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    double san = 0.0, sbn = 0.0;
    if constexpr (J + 2 < NB) {
      prev_upd<J, J + 2, 1>(acc, lp, mp);
      if constexpr (!NORL) {
      const double l2 = bcast_lane(acc[J], J + 2);
      acc[J + 2] = fma(-acc[J], l2, acc[J + 2]);
      san = bcast_lane(acc[J + 1], J + 2);
      sbn = bcast_lane(acc[J + 2], J + 2);
      } else { san = acc[J + 1] * 1e-3; sbn = fabs(acc[J + 2]) + 40.0; }
    }
    double lpn[NB];
    if constexpr (NOLDS) {
#pragma unroll
      for (int c = 0; c < NB; c++) lpn[c] = 1e-3 * c;
    } else if constexpr (J + 3 < NB) {
      constexpr int ce = (J + 3) + ((J + 3) & 1);
      if constexpr (((J + 3) & 1) != 0) lpn[J + 3] = s_T[J][J + 3];
#pragma unroll
      for (int c = ce; c + 1 < NB; c += 2) { const double2 v = *(const double2*)&s_T[J][c]; lpn[c] = v.x; lpn[c + 1] = v.y; }
    }
    constexpr int c0 = J + 3;
    if constexpr (J + 1 < NB) {
      if constexpr (!CUBIC) {
        SB(); double t = hxn * yn;
        SB(); prev_upd<J, c0, U, NOFMA>(acc, lp, mp);
        SB(); double e = fma(-t, yn, 0.5);
        SB(); prev_upd<J, c0 + U, U, NOFMA>(acc, lp, mp);
        SB(); yn = fma(yn, e, yn);
        SB(); prev_upd<J, c0 + 2 * U, U, NOFMA>(acc, lp, mp);
        SB(); t = hxn * yn;
        SB(); prev_upd<J, c0 + 3 * U, U, NOFMA>(acc, lp, mp);
        SB(); e = fma(-t, yn, 0.5);
        SB(); prev_upd<J, c0 + 4 * U, U, NOFMA>(acc, lp, mp);
        SB(); yn = fma(yn, e, yn); PIN(yn);
        SB(); prev_upd<J, c0 + 5 * U, NB, NOFMA>(acc, lp, mp);
        SB();
      } else {
        // one third-order step: e = 1/2 - (x/2) y^2 = (1 - x y^2) / 2;  y' = y + y e (1 + 3/2 e)   (error^3: 2^-23 -> 2^-69)
        SB(); double t = hxn * yn;
        SB(); prev_upd<J, c0, 2 * U, NOFMA>(acc, lp, mp);
        SB(); double e = fma(-t, yn, 0.5);
        SB(); prev_upd<J, c0 + 2 * U, 2 * U, NOFMA>(acc, lp, mp);
        SB(); const double p = fma(1.5, e, 1.0); const double q = yn * e;
        SB(); prev_upd<J, c0 + 4 * U, U, NOFMA>(acc, lp, mp);
        SB(); yn = fma(q, p, yn); PIN(yn);
        SB(); prev_upd<J, c0 + 5 * U, NB, NOFMA>(acc, lp, mp);
        SB();
      }
      FusedCol<J + 1, U, MODE>::run(acc, lpn, acc[J], yn, san, sbn, bad, s_T, lane);
    }
  }
};
template <int U, int MODE>
__device__ __noinline__ int diag_factor_fused(double (*s_L)[NB + 1], double (*s_X)[NB + 1], double (*s_Tu)[64]) {
  double (*s_T)[64] = (double (*)[64])__builtin_assume_aligned(s_Tu, 16);      // b128 broadcast reads with immediate offsets
  const int lane = threadIdx.x & 63, r = lane & 31;
  double acc[NB];
#pragma unroll
  for (int c = 0; c < NB; c++) acc[c] = (lane < 32) ? s_L[r][c] : ((c == r) ? 1.0 : 0.0);
  int bad = 0;
  const double piv = bcast_lane(acc[0], 0);
  bad |= (!(piv > 0.0) || !isfinite(piv)) ? 1 : 0;
  const double y0 = rsqrt_f64(piv);
  const double sa = bcast_lane(acc[0], 1), sb = bcast_lane(acc[1], 1);
  double lp0[NB];
#pragma unroll
  for (int c = 0; c < NB; c++) lp0[c] = 0.0;
  if constexpr ((MODE & 16) == 0) FusedCol<0, U, MODE>::run(acc, lp0, 0.0, y0, sa, sb, bad, s_T, lane);
  else acc[1] += y0 * sa + sb;
  if (lane < 32) {
#pragma unroll
    for (int c = 0; c < NB; c++) s_L[r][c] = (c <= r) ? acc[c] : 0.0;          // (the product does not need L11 back in LDS; kept for the comparison)
  } else {
#pragma unroll
    for (int rr = 0; rr < NB; rr++) s_X[rr][r] = acc[rr];                        // column r of L^-1 (zeros above the diagonal come out by themselves)
  }
  return bad;
}
// phase split of the inverse: (a) the two 16x16 diagonal inverses by lanes, (b) the two MFMA stages
__device__ __noinline__ void diag_invert_a(double (*s_L)[NB + 1], double (*s_X)[NB + 1], const double* s_dinv) {
  const int lane = threadIdx.x & 63;
  const int c = lane & 15, b = lane & 16;
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
// variant: the column c of a 16x16 inverse is zero above row c - start the substitution at row c (uniform loop, predicated
// start is not possible per lane, so instead: lanes keep the full loop but the L reads are done ONCE per row into registers
// shared by all lanes through broadcast b128 reads of a TRANSPOSED copy (s_T holds L^T after the factor: s_T[j][r] = l(r, j))
__device__ __noinline__ void diag_invert_a2(double (*s_T)[NB], double (*s_X)[NB + 1], const double* s_dinv) {
  const int lane = threadIdx.x & 63;
  const int c = lane & 15, b = lane & 16;
  double x[16];
  // x[rr] = (e - sum_{m<rr} l(b+rr, b+m) x[m]) * dinv;  l(b+rr, b+m) = s_T[b+m][b+rr]: for fixed m the rr run is contiguous,
  // so organise by COLUMN m (right-looking): after x[m] is final, subtract l(rr, m) x[m] from every later sum
  double sum[16];
#pragma unroll
  for (int rr = 0; rr < 16; rr++) sum[rr] = (rr == c) ? 1.0 : 0.0;
#pragma unroll
  for (int m = 0; m < 16; m++) {
    x[m] = sum[m] * s_dinv[b + m];
    int rr = m + 1;
    if (rr < 16 && (rr & 1)) { sum[rr] = fma(-s_T[b + m][b + rr], x[m], sum[rr]); rr++; }
#pragma unroll
    for (; rr + 1 < 16; rr += 2) {
      const double2 l = *(const double2*)&s_T[b + m][b + rr];
      sum[rr] = fma(-l.x, x[m], sum[rr]);
      sum[rr + 1] = fma(-l.y, x[m], sum[rr + 1]);
    }
  }
  if (lane < 32) {
#pragma unroll
    for (int rr = 0; rr < 16; rr++) { s_X[b + rr][b + c] = x[rr]; if (b == 0) s_X[rr][16 + c] = 0.0; }
  }
}
__device__ __noinline__ void diag_invert_b(double (*s_L)[NB + 1], double (*s_X)[NB + 1], double (*s_T)[NB]) {
  const int lane = threadIdx.x & 63;
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

template <int AHEAD, int INV>
__global__ __launch_bounds__(256) void k_diag(const double* A, double* Lout, double* Xout, unsigned long long* ticks) {
  __shared__ double s_L[NB][NB + 1];
  __shared__ double s_X[NB][NB + 1];
  __shared__ double s_dinv[NB];
  __shared__ __attribute__((aligned(16))) double s_T[NB][NB];
  __shared__ __attribute__((aligned(16))) double s_T2[NB][NB];
  __shared__ __attribute__((aligned(16))) double s_T64[NB][64];
  const int tid = threadIdx.x;
  unsigned long long tf = 0, ta = 0, tb = 0;
  for (int rep = 0; rep < REPS; rep++) {
    for (int i = tid; i < NB * NB; i += 64) { int r = i / NB, c = i % NB; s_L[r][c] = (c <= r) ? A[i] : 0.0; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (AHEAD >= 1000) diag_factor_fused<(AHEAD / 1000), (AHEAD % 100)>(s_L, s_X, s_T64);
    else if (AHEAD < 100) diag_factor_wave<(AHEAD < 100 ? AHEAD : 1)>(s_L, s_T, s_dinv); else diag_factor_pipe<(AHEAD < 1000 ? AHEAD / 100 : 1), (AHEAD % 100)>(s_L, s_T, s_dinv);
    __threadfence_block(); __builtin_amdgcn_wave_barrier();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (AHEAD >= 1000) { }
    else if (INV == 0) diag_invert_a(s_L, s_X, s_dinv); else diag_invert_a2(s_T, s_X, s_dinv);
    __threadfence_block(); __builtin_amdgcn_wave_barrier();
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    if (AHEAD < 1000) diag_invert_b(s_L, s_X, INV == 0 ? s_T : s_T2);
    __threadfence_block(); __builtin_amdgcn_wave_barrier();
    const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
    tf += t1 - t0; ta += t2 - t1; tb += t3 - t2;
    __syncthreads();
  }
  for (int i = tid; i < NB * NB; i += 64) { Lout[i] = s_L[i / NB][i % NB]; Xout[i] = s_X[i / NB][i % NB]; }
  if (tid == 0) { ticks[0] = tf; ticks[1] = ta; ticks[2] = tb; }
}

__global__ void k_spin(double* out, int n) {                 // keeps every CU busy so that the timed single-wave kernel runs at the loaded clock
  double a = threadIdx.x * 1e-3, b = 1.0000001;
  for (int i = 0; i < n; i++) { a = fma(a, b, 1e-9); b = fma(b, 0.9999999, 1e-9); }
  if (a == 12345.678) out[0] = a + b;
}
template <int AHEAD, int INV>
static void run(const char* name, const double* dA, double* dL, double* dX, unsigned long long* dt, std::vector<double>& L, std::vector<double>& X) {
  for (int w = 0; w < 2; w++) { k_spin<<<2048, 256>>>(dX, 400000); k_diag<AHEAD, INV><<<1, 64>>>(dA, dL, dX, dt); hipDeviceSynchronize(); }
  unsigned long long t[3]; hipMemcpy(t, dt, 24, hipMemcpyDeviceToHost);
  L.resize(NB * NB); X.resize(NB * NB);
  hipMemcpy(L.data(), dL, NB * NB * 8, hipMemcpyDeviceToHost); hipMemcpy(X.data(), dX, NB * NB * 8, hipMemcpyDeviceToHost);
  printf("%-28s factor %.2f us   inverse: lanes %.2f us + matrix cores %.2f us\n", name, t[0] * 10.0 / REPS * 1e-3, t[1] * 10.0 / REPS * 1e-3, t[2] * 10.0 / REPS * 1e-3);
}
int main() {
  std::vector<double> h(NB * NB);
  for (int i = 0; i < NB; i++) for (int j = 0; j < NB; j++) h[i * NB + j] = (i == j) ? 40.0 + (i % 7) : std::cos(0.37 * std::max(i, j) + 1.3 * std::min(i, j)) / (1.0 + 0.3 * std::abs(i - j));
  double *dA, *dL, *dX; unsigned long long* dt;
  hipMalloc(&dA, NB * NB * 8); hipMalloc(&dL, NB * NB * 8); hipMalloc(&dX, NB * NB * 8); hipMalloc(&dt, 64);
  hipMemcpy(dA, h.data(), NB * NB * 8, hipMemcpyHostToDevice);
  std::vector<double> L[24], X[24];
  run<1, 0>("ahead 1 (product, round 2)", dA, dL, dX, dt, L[0], X[0]);
  run<2, 0>("ahead 2", dA, dL, dX, dt, L[1], X[1]);
  run<3, 0>("ahead 3", dA, dL, dX, dt, L[2], X[2]);
  run<4, 0>("ahead 4", dA, dL, dX, dt, L[3], X[3]);
  run<6, 0>("ahead 6", dA, dL, dX, dt, L[4], X[4]);
  run<2, 1>("ahead 2, column-order inverse", dA, dL, dX, dt, L[5], X[5]);
  run<200, 0>("pipelined, 2 per chain op", dA, dL, dX, dt, L[6], X[6]);
  run<300, 0>("pipelined, 3 per chain op", dA, dL, dX, dt, L[7], X[7]);
  run<400, 0>("pipelined, 4 per chain op", dA, dL, dX, dt, L[8], X[8]);
  run<301, 0>("pipelined, 3, skip 1", dA, dL, dX, dt, L[9], X[9]);
  run<402, 0>("pipelined, 4, skip 2", dA, dL, dX, dt, L[10], X[10]);
  run<602, 0>("pipelined, 6, skip 2", dA, dL, dX, dt, L[11], X[11]);
  run<3000, 0>("FUSED factor+inverse, 3", dA, dL, dX, dt, L[12], X[12]);
  run<4000, 0>("FUSED factor+inverse, 4", dA, dL, dX, dt, L[13], X[13]);
  run<5000, 0>("FUSED factor+inverse, 5", dA, dL, dX, dt, L[14], X[14]);
  run<6000, 0>("FUSED factor+inverse, 6", dA, dL, dX, dt, L[15], X[15]);
  run<4001, 0>("FUSED, 4, cubic rsqrt step", dA, dL, dX, dt, L[16], X[16]);
  run<6001, 0>("FUSED, 6, cubic rsqrt step", dA, dL, dX, dt, L[17], X[17]);
  run<4002, 0>("(timing only) FUSED 4, no LDS", dA, dL, dX, dt, L[18], X[18]);
  run<4004, 0>("(timing only) FUSED 4, no readlanes", dA, dL, dX, dt, L[19], X[19]);
  run<4008, 0>("(timing only) FUSED 4, no deferred FMAs", dA, dL, dX, dt, L[20], X[20]);
  run<4014, 0>("(timing only) FUSED 4, chain only", dA, dL, dX, dt, L[21], X[21]);
  run<4006, 0>("(timing only) FUSED 4, no LDS, no RL", dA, dL, dX, dt, L[22], X[22]);
  run<4010, 0>("(timing only) FUSED 4, no LDS no FMA", dA, dL, dX, dt, L[23], X[23]);
  run<4016, 0>("(timing only) FUSED, no columns at all", dA, dL, dX, dt, L[22], X[22]);
  for (int v = 1; v < 18; v++)
    printf("variant %d vs 0: L %s (max diff %.3e), X max diff %.3e\n", v, memcmp(L[v].data(), L[0].data(), NB * NB * 8) ? "DIFFERS" : "bit-identical",
           [&] { double d = 0; for (int i = 0; i < NB * NB; i++) d = std::fmax(d, std::fabs(L[v][i] - L[0][i])); return d; }(),
           [&] { double d = 0; for (int i = 0; i < NB * NB; i++) d = std::fmax(d, std::fabs(X[v][i] - X[0][i])); return d; }());
  // residual of variant 0 against the input: max |L L^T - A|, max |X L - I|
  double e1 = 0, e2 = 0;
  for (int i = 0; i < NB; i++) for (int j = 0; j <= i; j++) {
    double s = 0; for (int k = 0; k <= j; k++) s += L[0][i * NB + k] * L[0][j * NB + k];
    e1 = std::fmax(e1, std::fabs(s - h[i * NB + j]));
    double t = 0; for (int k = j; k <= i; k++) t += X[0][i * NB + k] * L[0][k * NB + j];
    e2 = std::fmax(e2, std::fabs(t - (i == j ? 1.0 : 0.0)));
  }
  printf("residuals: |L L^T - A| %.3e   |X L - I| %.3e\n", e1, e2);
  return 0;
}
