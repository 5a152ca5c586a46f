// ============================================================================
// ba_solver.hip -- MI355X (gfx950) bundle adjustment behind the C ABI of
// include/orbslam_hip.h: drop-in for CeresOptimizer::{PoseOptimization,
// BundleAdjustment / GlobalBundleAdjustemnt, LocalBundleAdjustment, CheckOutlier(s)}
// (reference src/CeresOptimizer.cc:49-599) including the Ceres solve underneath
// (trust-region Levenberg-Marquardt with Ceres' default options, Huber loss with the
// Triggs corrector, EigenQuaternionParameterization, Jacobi scaling).
// Everything is fp64.
//
//   k_pose_lm        PoseOptimization: the WHOLE LM loop of one frame inside one
//                    workgroup (residual + 2x6 Jacobian per observation, 6x6 normal
//                    equations reduced in LDS, 6x6 Cholesky, step test) - no host
//                    round trip per iteration; batched one workgroup per frame.
//   BA (poses+points) per LM iteration, all on device, LM control in a device-side state:
//     k_ba_eval<mode>  residuals + the analytic SE(3) Jacobians per observation, stored FACTORED: {W = Q^T Q, r = 2 RX} (64 bytes,
//                      camera-major) and h = Q^T res - the comment above ld_rec8
//     k_ba_cam_blocks  6x6 pose blocks from a camera's records (one workgroup per camera), 3x3 landmark blocks behind them
//     k_ba_schur_prep  per point: (C+D)^-1, N = S_p (C+D)^-1 S_p; zero fill of S
//     k_ba_schur       reduced camera system S = B + D - E (C+D)^-1 E^T, one workgroup per block row, 128-pair segments per wave
//     k_chol_*         dense blocked Cholesky of S on the FP64 matrix cores (v_mfma_f64_16x16x4_f64 - the only MFMA user, as the
//                      dense reduced block is the only GEMM here): k_chol_la (one launch per 32-column step), k_chol_persist /
//                      k_chol_persist_blk (flag-linked persistent launches of single solves), k_chol_wg (one workgroup per
//                      problem of a lockstep batch); forward substitution rides the factorisation (augmented row)
//     k_chol_bsolve_*  backward substitution
//     k_ba_backsub     landmark back-substitution, candidate point, model cost change
//     k_ba_iter_begin / k_ba_after_eval / k_ba_iter_end   Ceres' step acceptance / radius update / convergence tests
// The exact Schur solve is mathematically identical to the reference's
// SPARSE_NORMAL_CHOLESKY.  No CPU fallback exists in this file.
// ============================================================================
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#include <numeric>
#include <chrono>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <string>
#include <atomic>
#include <cstdlib>
#include <climits>

#include "common.h"
#include "ba_math.h"
#include "handoff.h"
#include "wave_reduce.h"
#include "sim3_math.h"

namespace orbhip {

// ---------------------------------------------------------------------------- block reductions
template <int V>
__device__ __forceinline__ void block_reduce(double (&acc)[V], double* s_red /*[4*V]*/, double* s_out /*[V]*/) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < V; k++) {
    double v = acc[k];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) s_red[w * V + k] = v;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < V; k += blockDim.x) s_out[k] = (s_red[k] + s_red[V + k]) + (s_red[2 * V + k] + s_red[3 * V + k]);
  __syncthreads();
}

// the same for workgroups of up to 16 waves (s_red holds 16 * V values); fixed summation order
template <int V>
__device__ __forceinline__ void block_reduce_wide(double (&acc)[V], double* s_red /*[16*V]*/, double* s_out /*[V]*/) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int k = 0; k < V; k++) {
    double v = acc[k];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) s_red[w * V + k] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < V) { double t = 0.0; for (int i = 0; i < nw; i++) t += s_red[i * V + threadIdx.x]; s_out[threadIdx.x] = t; }
  __syncthreads();
}

// Wave sum on the VALU only (DPP row permutations + row broadcasts, no LDS crossbar): 6 steps x (2 v_mov_dpp + v_add_f64).
// The __shfl_xor tree costs 12 ds_bpermute per value; with 36 values per thread the LDS pipe, not the loads, bounded
// k_ba_schur.  Fixed order: quads, 8, 16 inside each row of 16 lanes, then rows (0+1), (2+3), ((2+3)+(0+1)); the total is
// valid in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_take<0xB1, 0xF>(v);      // quad_perm [1,0,3,2]
  v += dpp_take<0x4E, 0xF>(v);      // quad_perm [2,3,0,1]
  v += dpp_take<0x141, 0xF>(v);     // row_half_mirror
  v += dpp_take<0x140, 0xF>(v);     // row_mirror: every lane of a row holds the row total
  v += dpp_take<0x142, 0xA>(v);     // row_bcast15 into rows 1 and 3
  v += dpp_take<0x143, 0xC>(v);     // row_bcast31 into rows 2 and 3
  return v;
}
// value of lane `src` (wave-uniform, a constant after unrolling: two v_readlane_b32)
__device__ __forceinline__ double lane_bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
template <int V>
__device__ __forceinline__ void block_reduce_dpp(double (&acc)[V], double* s_red /*[(blockDim.x/64)*V]*/, double* s_out /*[V]*/) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int k = 0; k < V; k++) {
    const double t = wave_sum_dpp(acc[k]);
    if (lane == 63) s_red[w * V + k] = t;
  }
  __syncthreads();
  if ((int)threadIdx.x < V) { double t = 0.0; for (int i = 0; i < nw; i++) t += s_red[i * V + threadIdx.x]; s_out[threadIdx.x] = t; }
  __syncthreads();
}

// in-place Cholesky + solve of a tiny SPD system (n <= 6), row-major; returns false if not PD
__device__ bool small_chol_solve(double* A, double* b, int n) {
  for (int j = 0; j < n; j++) {
    double d = A[j * n + j];
    for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0.0) || !isfinite(d)) return false;
    d = sqrt(d);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[i * n + j];
      for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= A[i * n + k] * b[k]; b[i] = s / A[i * n + i]; }
  for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= A[k * n + i] * b[k]; b[i] = s / A[i * n + i]; }
  return true;
}

#include "ba_small_lm.inc"   // PoseOptimization and OptimizeSim3

// ============================================================================ general BA
struct BaState {
  double radius, decrease_factor, x_cost, x_norm, initial_cost, cand_cost, model_cost_change, step_norm2, gmax;
  int iteration, successful_steps, termination, done, need_eval, first, valid, invalid_steps, chol_fail, accepted, max_iters;
  int e_dirty;     // (unused since the factored records are written by k_ba_eval; kept for the layout of the collected state)
};

struct BaDev {            // device pointers of one problem
  int ncam, npts, nobs, nfc, n6, npad;
  const double* K4; const unsigned char* cam_fixed; const int* cam_col;
  double* poses; double* pts; double* cand_poses; double* cand_pts;
  const int* obs_cam; const int* obs_pt; const double* obs_uv; const double* obs_w; const unsigned char* obs_robust;
  const int* pt_off;                 // [npts+1] observations grouped by point
  const int* cam_off; const int* cam_obs; const int* cam_obs_pt;   // per-camera lists (sorted by point)
  const int* cam_pos;                // [nobs] position of an observation inside its camera's list (inverse of cam_obs)
  double* Hc;                        // [nobs][3] camera-major: h = Q^T res of every observation (with the E record: all the block kernels need)
  double* B; double* gc; double* C; double* gp;       // unscaled blocks: B[nfc][21], gc[nfc][6], C[npts][6], gp[npts][3]
  double* scale_c; double* scale_p;  // Jacobi scaling [nfc][6], [npts][3]
  double* Cinv; double* gps; double* E;   // Cinv[npts][6], gps[npts][3], E[nobs] factored 64-byte records in camera-major order (ld_rec8), written by k_ba_eval
  double* Ng;                        // [npts][9] {N = S_p (C_s+D)^-1 S_p (6, symmetric), g_p (3)}: what k_ba_schur needs of a point, one gather
  double* t3;                        // [nobs][3] E_i^T y_cam of the landmark back-substitution
  double* ae_part; unsigned int* ae_ticket;      // k_ba_after_eval: [AE_NS][4] slice results {cost (slice 0), |x|^2, gradient max-norm} and the arrival counter of its workgroups
  double* camrec;                    // [ncam][16] what k_ba_backsub needs of a camera in ONE 128-byte line: {R (9), S_c y (6), 1 = free camera with a valid step | 0} (k_ba_cam_update)
  double* S; double* rhs;            // reduced system S[npad+1][npad] (lower; row npad = rhs^T), rhs/yc [npad]
  double* Dinv;                      // inverse of every 32x32 diagonal Cholesky block [npad/32][32][32]
  double* Mb;                        // persistent Cholesky: M_k = X_k P_k of every step [npad/32][32][32]
  int* cflags;                       // persistent Cholesky: hand-off flags of this problem [ncflags], zeroed by k_ba_iter_begin
  int ncflags;
  const int* pair_i; const int* pair_j;   // Schur pair lists, block after block in (a, b) order, a <= b (pair_i: POSITION of camera a's observation in its camera's list, pair_j: camera b's observation as its index in camera-major order = E record)
  const int4* row_meta;              // [nfc][2] block row a: {first, end of the diagonal block's pairs, first, end of the row's segments}, {first entry, length of camera a's list, camera index, -}
  const int4* seg;                   // the off-diagonal blocks cut into SEGMENTS of <= SR_SEG pairs: {first pair, end, column b, 1 = first | 2 = last segment of its block | camera index of b << 2}
  const int* free_cams;              // [nfc] reduced column -> camera index
  const int* tile_first;             // [npad / 32 + 1] skyline of S by 32-row tiles: first tile column with a structural non-zero (entry npad / 32: the rhs row, 0)
  double* part;                      // partial sums: [3][nparts]
  int nparts; int fix_points;
  const unsigned char* cam_local; unsigned char* erase;   // LocalBA classification (k_ba_classify): local flags [ncam], result [nobs] (device order)
  int pt_in_eval;                    // 1: the 3x3 landmark blocks are summed by k_ba_eval<0> itself (every point has <= PT_MAXRUN observations), 0: by k_ba_cam_blocks' landmark workgroups
  int chol_la;                       // 1: this problem's reduced system is factored by the look-ahead family (npad <= 1024, or larger with a narrow skyline), 0: two-level blocking
  int wg_rows;                       // most active rows (+ the rhs row) any column has: k_chol_wg's list holds NB + 2
  int band;                          // widest envelope of a tile row, in tiles (max over i of i - tile_first[i])
  int persist_ring_rows;             // first row that is NOT walked by the ring (rows from here on keep a workgroup set each: the wide envelopes of a loop-closed map)
  int persist_ring;                  // > 0: k_chol_persist's row workgroups walk the rows slot, slot + ring, ... (narrow skyline: 2 ring + 3 workgroups in all)
  int persist_nwg;                   // workgroups of k_chol_persist for this system (the chain, a producer and its consumers per block row, the rhs row's two)
  double huber;
  const volatile unsigned char* stop_dev;   // device-visible mirror of the caller's stop flag (pinned host byte of the calling thread)
  unsigned char* out;                // this problem's slice of the batch's output block: {BaState | poses | pts | erase}, 256-byte aligned parts (k_ba_collect)
  BaState* st;
};

// The early-exit flags of a problem, read with UNCONDITIONAL loads: `if (st->done || !st->valid || st->chol_fail) return;` makes
// the compiler fetch and wait for each flag in turn (short-circuit semantics) - two or three serial round trips at the top of
// every one of the ~45 kernels of an LM iteration.
struct StFlags { int done, need_eval, valid, chol_fail, accepted; };
__device__ __forceinline__ StFlags ld_flags(const BaState* st) {
  StFlags f; f.done = st->done; f.need_eval = st->need_eval; f.valid = st->valid; f.chol_fail = st->chol_fail; f.accepted = st->accepted;
  // (all five requested together and waited for HERE: left alone, the compiler loads `done` and `valid`, waits, branches, and fetches the
  // others behind the branch where they are used - a round trip each on kernels that are made of round trips)
  asm volatile("" : "+v"(f.done), "+v"(f.need_eval), "+v"(f.valid), "+v"(f.chol_fail), "+v"(f.accepted));
  return f;
}

#define BA_TPB 256
#define NB 32                          /* tile edge of the reduced system's factorisation (ba_cholesky.inc) */

// The per-observation block E = (Jc S_c)^T (Jp S_p) (6x3) is never stored: it FACTORS.  With Q = sqrt(rho') w dpi/dX_c (2x3, four
// non-zero entries), the rotated point RX and the camera's rotation R (ba_math.h: Jc = Q [I | -2 [RX]x], Jp = Q R),
//     E = S_c [W; [r]x W] R S_p,    W = Q^T Q (symmetric 3x3 with W01 = 0: five numbers),  r = 2 RX,
// so an observation keeps {w00, w11, w02, w12, w22, r0, r1, r2}: 64 bytes - half a cache line, aligned - instead of the 144 of the
// 18 products, and what depends on the camera only (R, S_c) or on the point only (S_p) is applied once per block / per point:
//     E_a (C_s+D)^-1 E_b^T = S_c,a [ G_a (R_a N R_b^T) G_b^T ] S_c,b,   G = [W; [r]x W] (6x3),  N = S_p (C_s+D)^-1 S_p.
// (Round 4: 18-double records made k_ba_schur move 2 GB per launch of a 64-problem batch through a 4 MB L2 per XCD; a problem's
// records are now 3.2 MB.)  The algebra is exact whatever the norm of the quaternion (Jc and Jp are built from the same RX and R).
// Records are stored in CAMERA-MAJOR order (record index = cam_pos[i], the position of the observation in the concatenated
// per-camera lists): k_ba_schur streams camera a's records, and those it gathers from a camera b ascend inside b's contiguous run.
// They depend on the iterate only - not on the LM radius, not on the scaling: k_ba_eval writes them with the Jacobians (mode 0), beside
// h = Q^T r (Hc), and k_ba_cam_blocks forms Jc^T Jc and Jc^T r from the same records (round 4 kept 14 doubles of Jc and r per
// observation for it, and a kernel of its own, k_ba_E, rewrote 18-double E records whenever the iterate had changed).
// "These doubles are needed HERE": an empty asm the values pass through.  The compiler otherwise moves a load down to its first use - behind
// a branch, a select it turns into a branch, a barrier - and ends every divergent block with a wait for what the block requested; pinned,
// the loads issued before the pin leave together and are waited for once.
__device__ __forceinline__ void pin8(double* x) {
  asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
}
__device__ __forceinline__ void pin4(double* x) { asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])); }
__device__ __forceinline__ void pin4i(int& a, int& b, int& c, int& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
// Loads through ADDRESS-SPACE-1 pointers (round 6).  The pointers of the BaDev struct are generic, so the compiler emits FLAT loads, and a
// flat load counts as LDS traffic too (lgkmcnt): in a kernel that gathers records while it reads its LDS - k_ba_schur - every wait for an
// LDS read also waited for the gathers in flight.  ldg / ld_rec8g are global_load instructions: vmcnt only.
#define ORB_AS1 __attribute__((address_space(1)))
typedef double v2d_t __attribute__((ext_vector_type(2)));
typedef int v4i_t __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ T ldg(const T* p) { return *(const T ORB_AS1*)p; }
__device__ __forceinline__ int4 ldg4(const int4* p) { const v4i_t v = *(const v4i_t ORB_AS1*)p; return make_int4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void ld_rec8g(const double* base, size_t q, double* c) {
  const v2d_t ORB_AS1* m = (const v2d_t ORB_AS1*)(base + 8 * q);
#pragma unroll
  for (int k = 0; k < 4; k++) { const v2d_t v = m[k]; c[2 * k] = v.x; c[2 * k + 1] = v.y; }
}
__device__ __forceinline__ void ld_rec8(const double* __restrict__ base, size_t q, double* c) {
  const double2* m = (const double2*)(base + 8 * q);
#pragma unroll
  for (int k = 0; k < 4; k++) { const double2 v = m[k]; c[2 * k] = v.x; c[2 * k + 1] = v.y; }
}
// One observation's share of its landmark's blocks, Jp = Q R:  Jp^T Jp = R^T W R (6 numbers),  Jp^T res = R^T h (3), from the factored
// record {w00, w11, w02, w12, w22} / h and the camera's rotation.  ONE function for both places that form it (ba_pt_blocks_body from the
// stored record, k_ba_eval<0> from the registers the record was written from): the same operations on the same doubles.
__device__ __forceinline__ void pt_terms(double w00, double w11, double w02, double w12, double w22, double h0, double h1, double h2, const double* Rc, double* o) {
  double V[9];                                                          // V = W R
#pragma unroll
  for (int k = 0; k < 3; k++) {
    V[k] = fma(w02, Rc[6 + k], w00 * Rc[k]);
    V[3 + k] = fma(w12, Rc[6 + k], w11 * Rc[3 + k]);
    V[6 + k] = fma(w22, Rc[6 + k], fma(w12, Rc[3 + k], w02 * Rc[k]));
  }
  o[0] = fma(Rc[6], V[6], fma(Rc[3], V[3], Rc[0] * V[0])); o[1] = fma(Rc[6], V[7], fma(Rc[3], V[4], Rc[0] * V[1])); o[2] = fma(Rc[6], V[8], fma(Rc[3], V[5], Rc[0] * V[2]));
  o[3] = fma(Rc[7], V[7], fma(Rc[4], V[4], Rc[1] * V[1])); o[4] = fma(Rc[7], V[8], fma(Rc[4], V[5], Rc[1] * V[2]));
  o[5] = fma(Rc[8], V[8], fma(Rc[5], V[5], Rc[2] * V[2]));
  o[6] = fma(Rc[6], h2, fma(Rc[3], h1, Rc[0] * h0)); o[7] = fma(Rc[7], h2, fma(Rc[4], h1, Rc[1] * h0)); o[8] = fma(Rc[8], h2, fma(Rc[5], h1, Rc[2] * h0));
}
#define PT_MAXRUN 112                  /* observations of one landmark up to which k_ba_eval<0> sums its blocks: (256 + 111) x 9 doubles fit the record staging area */
// ---- residuals + Jacobians at x (mode 0) or cost only at the candidate (mode 1) -------------------
template <int mode>      // (a template parameter: the cost-only instance carries neither the staging LDS nor the Jacobian registers)
__global__ __launch_bounds__(BA_TPB) void k_ba_eval(const BaDev* __restrict__ Dv) {
  const BaDev D = Dv[blockIdx.y];
  __shared__ double s_red[4], s_out[1];
  const BaState* st = D.st;
  if ((int)blockIdx.x * BA_TPB >= max(D.nobs, 1)) return;               // batched launch: grid.x is the maximum over the problems
  const int i = blockIdx.x * BA_TPB + threadIdx.x;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __shared__ double s_rec[mode == 0 ? BA_TPB / 64 : 1][mode == 0 ? 64 : 1][13];   // mode 0: the wave's records {W (5), r (3), h (3)} on their way out (+ pad: odd pitch)
  __shared__ int s_q[mode == 0 ? BA_TPB / 64 : 1][mode == 0 ? 64 : 1];            // ... and their places (camera-major position, -1: none)
  const bool have = i < D.nobs;
  const bool ptphase = mode == 0 && D.pt_in_eval && !D.fix_points && D.nobs > 0;
  const int b0 = blockIdx.x * BA_TPB, iend = min(b0 + BA_TPB, D.nobs);
  // Round 6: the loads leave in TWO groups, each waited for once (pin4 / pin8).  (A) what hangs off the observation index: camera, landmark,
  // camera-major place, measurement, weight - and the same of the observation this thread re-evaluates when the last landmark's run
  // continues behind the workgroup (i2), and the last landmark's index; (B) what hangs off those: intrinsics, pose, point, the camera's
  // column, the landmark's range.  Requested at their uses they made eleven dependent round trips (cam_pos behind `want`, the landmark phase
  // four of its own behind the barrier).
  // No branches around the loads - the compiler ends a divergent block with a wait for everything it requested: a thread without an
  // observation reads the workgroup's first one, always there when the problem has any, and its indices are zeroed behind the pin.
  const int i2 = iend + (int)threadIdx.x;
  const bool have2 = ptphase && (int)threadIdx.x < PT_MAXRUN - 1 && i2 < D.nobs;
  int c_mine = 0, p_mine = 0, q_mine = -1, rob = 0, c2 = 0, p2 = 0, rob2 = 0, plast = 0;
  double m4[4] = {0.0, 0.0, 0.0, 0.0}, n4[4] = {0.0, 0.0, 0.0, 0.0};           // {u, v, weight, -} of i and of i2
  const int ia = have ? i : b0, ib = have2 ? i2 : b0;
  c_mine = D.obs_cam[ia]; p_mine = D.obs_pt[ia]; rob = D.obs_robust[ia];
  if (mode == 0) q_mine = D.cam_pos[ia];
  m4[0] = D.obs_uv[2 * (size_t)ia]; m4[1] = D.obs_uv[2 * (size_t)ia + 1]; m4[2] = D.obs_w[ia];
  if (mode == 0) {
    c2 = D.obs_cam[ib]; p2 = D.obs_pt[ib]; rob2 = D.obs_robust[ib];
    n4[0] = D.obs_uv[2 * (size_t)ib]; n4[1] = D.obs_uv[2 * (size_t)ib + 1]; n4[2] = D.obs_w[ib];
    plast = D.obs_pt[max(iend - 1, 0)];
  }
  const StFlags F = ld_flags(st);
  pin4i(c_mine, p_mine, q_mine, rob); pin4(m4);
  if (mode == 0) { pin4i(c2, p2, rob2, plast); pin4(n4); }
  if (F.done) return;
  if (mode == 0 && !F.need_eval) return;
  if (mode == 1 && !F.valid) return;
  if (!have) { c_mine = 0; p_mine = 0; q_mine = -1; }
  if (!have2) { c2 = 0; p2 = 0; }
  if (!ptphase) plast = 0;
  const double* poses = mode ? D.cand_poses : D.poses;
  const double* pts = mode ? D.cand_pts : D.pts;
  double kp[16] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};       // K4 (4), pose (7), X (3), -, -
  double kp2[16] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int col = -1, po0 = 0, po1 = 0, run_end = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) kp[k] = D.K4[4 * (size_t)c_mine + k];
#pragma unroll
  for (int k = 0; k < 7; k++) kp[4 + k] = poses[7 * (size_t)c_mine + k];
#pragma unroll
  for (int k = 0; k < 3; k++) kp[11 + k] = pts[3 * (size_t)p_mine + k];
  if (mode == 0) {
    col = D.cam_col[c_mine]; po0 = D.pt_off[p_mine]; po1 = D.pt_off[p_mine + 1];
    run_end = D.pt_off[plast + 1];                              // (uniform) where the run of the workgroup's last landmark ends
  }
  pin8(kp); pin8(kp + 8);
  if (mode == 0) pin4i(col, po0, po1, run_end);
  double acc[1] = {0.0};
  double rec[11] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};        // this observation's record {W (5), r (3), h (3)}
  if (have) {
    double r[2], Jc[12], RX[3];
    // (an observation by a FIXED camera still feeds its landmark's block: its record is written too unless the landmarks are fixed)
    const bool want = (mode == 0) && (col >= 0 || !D.fix_points);
    // (Jc as a compile-time choice: `want ? Jc : nullptr` kept Jc and r in SCRATCH memory, 112 bytes per lane written and read back - 40 % of
    // what this kernel wrote to HBM; 171 -> 127 us per launch of a 64-problem C4 batch)
    double rho = reproj_eval(kp, kp + 4, kp + 11, m4[0], m4[1], m4[2], rob, D.huber, r, mode == 0 ? Jc : nullptr, nullptr, RX);
    acc[0] = 0.5 * rho;
    if (mode == 0) {
      if (want) {
        // The Jacobians leave in FACTORED form, grouped by camera (the comment above ld_rec8): {W = Q^T Q, r = 2 RX} is the record
        // k_ba_schur / k_ba_backsub / the block kernels work from, h = Q^T res the gradients' share - 88 bytes per observation, the
        // only thing this kernel writes (it is bound by the HBM WRITE rate, ~2.7 TB/s of scattered 64- and 24-byte pieces: the 2x6 and
        // 2x3 Jacobians were 160 bytes)
        const double q00 = Jc[0], q02 = Jc[2], q11 = Jc[7], q12 = Jc[8];
        rec[0] = q00 * q00; rec[1] = q11 * q11; rec[2] = q00 * q02; rec[3] = q11 * q12; rec[4] = q02 * q02 + q12 * q12;
        rec[5] = 2.0 * RX[0]; rec[6] = 2.0 * RX[1]; rec[7] = 2.0 * RX[2];
        rec[8] = q00 * r[0]; rec[9] = q11 * r[1]; rec[10] = q02 * r[0] + q12 * r[1];
        double* t = s_rec[w][lane];
#pragma unroll
        for (int k = 0; k < 11; k++) t[k] = rec[k];
      } else q_mine = -1;
    }
  }
  if (mode == 0) {
    // (the re-evaluated observation's intrinsics, pose and point: requested here, behind the first evaluation - 32 registers that would
    // otherwise be live through it and cost the kernel a wave per SIMD -, in flight while the records leave)
#pragma unroll
    for (int k = 0; k < 4; k++) kp2[k] = D.K4[4 * (size_t)c2 + k];
#pragma unroll
    for (int k = 0; k < 7; k++) kp2[4 + k] = D.poses[7 * (size_t)c2 + k];
#pragma unroll
    for (int k = 0; k < 3; k++) kp2[11 + k] = D.pts[3 * (size_t)p2 + k];
    // The records go to scattered places (the wave's observations are a run of the point-major order): four lanes write one record's
    // 64 bytes per store instruction, three its h - whole runs, not 64 sixteen-byte pieces of 64 lines (the L2 sees a third of the requests)
    s_q[w][lane] = q_mine;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int idx = lane + 64 * t, rr = idx >> 2, part = idx & 3;
      const int q = s_q[w][rr];
      if (q >= 0) *(double2*)(D.E + 8 * (size_t)q + 2 * part) = make_double2(s_rec[w][rr][2 * part], s_rec[w][rr][2 * part + 1]);
    }
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const int idx = lane + 64 * t, rr = idx / 3, part = idx - 3 * rr;
      const int q = s_q[w][rr];
      if (q >= 0) D.Hc[3 * (size_t)q + part] = s_rec[w][rr][8 + part];
    }
    // ---- the 3x3 landmark blocks (round 5; they were a pass of their own over the records, gathered by point: ba_pt_blocks_body).  The
    // observations of a landmark are consecutive, so the workgroup that holds a landmark's FIRST observation sums its run in order from
    // the nine terms every thread leaves in LDS (the staging area, free once the records are out); a run that continues behind the
    // workgroup's last observation is evaluated again by the first threads (<= PT_MAXRUN - 1 observations).  Same terms, same order as
    // the separate pass: the bits of C and g_p do not change.
    if (ptphase) {
      __syncthreads();                                          // every wave's records have left the staging area
      double* s_pt = &s_rec[0][0][0];                           // [256 + PT_MAXRUN - 1][9]
      if (have) {
        double Rc[9], o[9];
        quat_to_R(kp + 7, Rc);
        pt_terms(rec[0], rec[1], rec[2], rec[3], rec[4], rec[8], rec[9], rec[10], Rc, o);
#pragma unroll
        for (int k = 0; k < 9; k++) s_pt[9 * threadIdx.x + k] = o[k];
      }
      pin8(kp2); pin8(kp2 + 8);
      if ((int)threadIdx.x < run_end - iend) {                  // (have2 holds: the run ends inside the observations and is <= PT_MAXRUN long)
        double r2[2], J2[12], RX2[3], Rc[9], o[9];
        reproj_eval(kp2, kp2 + 4, kp2 + 11, n4[0], n4[1], n4[2], rob2, D.huber, r2, J2, nullptr, RX2);
        const double q00 = J2[0], q02 = J2[2], q11 = J2[7], q12 = J2[8];
        quat_to_R(kp2 + 7, Rc);
        pt_terms(q00 * q00, q11 * q11, q00 * q02, q11 * q12, q02 * q02 + q12 * q12, q00 * r2[0], q11 * r2[1], q02 * r2[0] + q12 * r2[1], Rc, o);
#pragma unroll
        for (int k = 0; k < 9; k++) s_pt[9 * (BA_TPB + threadIdx.x) + k] = o[k];
      }
      __syncthreads();
      if (have && i == po0) {
        const int hi = po1;
        double C[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
        for (int ii = i; ii < hi; ii++) {
          const double* tq = s_pt + 9 * (ii - b0);
#pragma unroll
          for (int k = 0; k < 6; k++) C[k] += tq[k];
#pragma unroll
          for (int k = 0; k < 3; k++) g[k] += tq[6 + k];
        }
        for (int k = 0; k < 6; k++) D.C[6 * (size_t)p_mine + k] = C[k];
        for (int k = 0; k < 3; k++) D.gp[3 * (size_t)p_mine + k] = g[k];
      }
    }
  }
  block_reduce<1>(acc, s_red, s_out);
  if (threadIdx.x == 0) D.part[(mode ? 1 : 0) * D.nparts + blockIdx.x] = s_out[0];
}

// ---- 6x6 pose blocks: B_c = sum Jc^T Jc, g_c = sum Jc^T r over the camera's observations ----------
__device__ __forceinline__ void ba_pt_blocks_body(const BaDev& D, int bx);
// k_ba_cam_blocks also carries the 3x3 landmark blocks (workgroups behind the `ncam_grid` camera workgroups): the two were
// separate launches of ~7 us each on a single solve's dependent chain, and neither depends on the other.
__global__ __launch_bounds__(BA_TPB) void k_ba_cam_blocks(const BaDev* __restrict__ Dv, int ncam_grid) {
  const BaDev D = Dv[blockIdx.y];
  __shared__ double s_red[4 * 27], s_out[27];
  const BaState* st = D.st;
  // (the camera's column and list range travel with the state flags: one round trip, they were three)
  const int c = blockIdx.x;
  const bool camwg = (int)blockIdx.x < ncam_grid && c < D.ncam;
  int cc = D.cam_col[camwg ? c : 0], lo = D.cam_off[camwg ? c : 0], hi = D.cam_off[camwg ? c + 1 : 0], pad_ = 0;
  const StFlags F = ld_flags(st);
  pin4i(cc, lo, hi, pad_);
  if (F.done || !F.need_eval) return;
  if ((int)blockIdx.x >= ncam_grid) { ba_pt_blocks_body(D, (int)blockIdx.x - ncam_grid); return; }
  if (c >= D.ncam) return;
  if (cc < 0) return;
  double acc[27];
#pragma unroll
  for (int k = 0; k < 27; k++) acc[k] = 0.0;
  // streamed: the list order is the record order.  Jc = Q [I | -[r]x]:  Jc^T Jc = [W, -K; -K^T, L] with K = W [r]x, L = -[r]x K;  Jc^T res = [h; r x h]
  auto one = [&](const double* c8, const double* hp, bool v) {
    const double h0 = v ? hp[0] : 0.0, h1 = v ? hp[1] : 0.0, h2 = v ? hp[2] : 0.0;           // (an entry beyond the list: zero weight, the sums keep their bits)
    const double w00 = v ? c8[0] : 0.0, w11 = v ? c8[1] : 0.0, w02 = v ? c8[2] : 0.0, w12 = v ? c8[3] : 0.0, w22 = v ? c8[4] : 0.0, r0 = c8[5], r1 = c8[6], r2 = c8[7];
    const double W[3][3] = {{w00, 0.0, w02}, {0.0, w11, w12}, {w02, w12, w22}};
    double K[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++) { K[i][0] = W[i][1] * r2 - W[i][2] * r1; K[i][1] = W[i][2] * r0 - W[i][0] * r2; K[i][2] = W[i][0] * r1 - W[i][1] * r0; }
    acc[sym6(0, 0)] += w00; acc[sym6(0, 2)] += w02; acc[sym6(1, 1)] += w11; acc[sym6(1, 2)] += w12; acc[sym6(2, 2)] += w22;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) acc[sym6(i, 3 + j)] -= K[i][j];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      acc[sym6(3, 3 + j)] += r2 * K[1][j] - r1 * K[2][j];              // L = -[r]x K, upper triangle
      if (j >= 1) acc[sym6(4, 3 + j)] += r0 * K[2][j] - r2 * K[0][j];
      if (j >= 2) acc[sym6(5, 3 + j)] += r1 * K[0][j] - r0 * K[1][j];
    }
    acc[21] += h0; acc[22] += h1; acc[23] += h2;
    acc[24] += r1 * h2 - r2 * h1; acc[25] += r2 * h0 - r0 * h2; acc[26] += r0 * h1 - r1 * h0;
  };
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (hi > lo) {                                                          // two list entries per thread requested before either is used
    const int e0 = lo + tid, e1 = lo + tid + BA_TPB;
    const bool v0 = e0 < hi, v1 = e1 < hi;
    double a8[8], b8[8], ha[3], hb[3];
    ld_rec8(D.E, (size_t)(v0 ? e0 : lo), a8); ld_rec8(D.E, (size_t)(v1 ? e1 : lo), b8);
#pragma unroll
    for (int k = 0; k < 3; k++) { ha[k] = D.Hc[3 * (size_t)(v0 ? e0 : lo) + k]; hb[k] = D.Hc[3 * (size_t)(v1 ? e1 : lo) + k]; }
    one(a8, ha, v0); one(b8, hb, v1);
    for (int e = lo + tid + 2 * BA_TPB; e < hi; e += BA_TPB) {
      double c8[8];
      ld_rec8(D.E, (size_t)e, c8);
      one(c8, D.Hc + 3 * (size_t)e, true);
    }
  }
  {                                                                       // 27 sums of the workgroup: transposing wave reduction, then the four waves in order
    double a36[36];
#pragma unroll
    for (int k = 0; k < 36; k++) a36[k] = k < 27 ? acc[k] : 0.0;
    const double t = wave_reduce36(a36, lane);
    const int sl = wave_reduce36_slot(lane);
    if (sl >= 0 && sl < 27) s_red[w * 27 + sl] = t;
    __syncthreads();
    if (tid < 27) s_out[tid] = (s_red[tid] + s_red[27 + tid]) + (s_red[2 * 27 + tid] + s_red[3 * 27 + tid]);
    __syncthreads();
  }
  if (threadIdx.x < 21) D.B[21 * (size_t)cc + threadIdx.x] = s_out[threadIdx.x];
  if (threadIdx.x < 6) D.gc[6 * (size_t)cc + threadIdx.x] = s_out[21 + threadIdx.x];
}

// ---- 3x3 landmark blocks ------------------------------------------------------------------------------
__device__ __forceinline__ void ba_pt_blocks_body(const BaDev& D, int bx) {
  if (D.fix_points || D.pt_in_eval) return;                            // (pt_in_eval: k_ba_eval<0> has summed them)
  const int p = bx * BA_TPB + threadIdx.x;
  if (p >= D.npts) return;
  // Jp = Q R:  Jp^T Jp = R^T W R,  Jp^T res = R^T h  from the observation's factored record (camera-major: gathered) and its camera's rotation
  // (tried: four observations at a time with every load of the four requested before the first is used - 60 % more slots than
  // observations at five views per landmark and 120 registers of operands: the launch went from 132 to 166 us)
  double C[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
  for (int i = D.pt_off[p]; i < D.pt_off[p + 1]; i++) {
    const size_t q = (size_t)D.cam_pos[i];
    double c8[8], Rc[9];
    ld_rec8(D.E, q, c8);
    const double h0 = D.Hc[3 * q], h1 = D.Hc[3 * q + 1], h2 = D.Hc[3 * q + 2];
    quat_to_R(D.poses + 7 * (size_t)D.obs_cam[i] + 3, Rc);
    double o[9];
    pt_terms(c8[0], c8[1], c8[2], c8[3], c8[4], h0, h1, h2, Rc, o);
#pragma unroll
    for (int k = 0; k < 6; k++) C[k] += o[k];
#pragma unroll
    for (int k = 0; k < 3; k++) g[k] += o[6 + k];
  }
  for (int k = 0; k < 6; k++) D.C[6 * (size_t)p + k] = C[k];
  for (int k = 0; k < 3; k++) D.gp[3 * (size_t)p + k] = g[k];
}

// ---- start of an evaluation: x_cost, Jacobi scaling (first time), gradient max-norm, |x| ----------
#define AE_TPB 1024
#define AE_NS 8         // workgroups per problem: one 1024-thread workgroup walking every camera and landmark was bound by ONE CU's bandwidth (38 us at C5 size)
__device__ __forceinline__ void ba_iter_begin_body(const BaDev& D);
// AE_NS workgroups per problem take a slice of the cameras and landmarks each (slice s: indices tid + 1024 s, + 1024 AE_NS, ...); the LAST
// one to arrive (a ticket per problem) adds the slices' |x|^2 in slice order - the same partition whatever the batch, so a problem's bits do
// not depend on how it is called -, takes the maximum of the gradient norms and the cost that slice 0 summed exactly as the one-workgroup
// kernel did, and then does the BEGINNING of the next iteration (k_ba_iter_begin's body: flag reset, stop flag, iteration cap, minimum
// radius: the two were back-to-back one-workgroup launches on every solve's dependent chain).
// (gridDim.x == 1 - lockstep batches of >= 8 problems, which fill the device by themselves: ONE workgroup walks the AE_NS slices one after
// the other - the same slices, the same per-slice sums, the same order of adding them: the same bits)
__global__ __launch_bounds__(AE_TPB) void k_ba_after_eval(const BaDev* __restrict__ Dv) {
  const BaDev D = Dv[blockIdx.y];
  __shared__ double s_red[16 * 3], s_out[3];
  __shared__ double s_max[16];
  __shared__ int s_last;
  BaState* st = D.st;
  const StFlags F = ld_flags(st);
  if (F.done) return;
  const int tid = threadIdx.x, stride = AE_TPB * AE_NS;
  const bool solo = gridDim.x == 1;
  if (!F.need_eval) { if (blockIdx.x == 0) ba_iter_begin_body(D); return; }      // (a rejected step: nothing to evaluate, the next iteration still begins)
  const bool first = st->first != 0;
  double cost = 0.0, x2 = 0.0, gm_all = 0.0;                   // thread 0 of a solo workgroup: the slices' results as they come
  for (int sl = solo ? 0 : (int)blockIdx.x; sl < (solo ? AE_NS : (int)blockIdx.x + 1); sl++) {
    if (first) {
      for (int j = tid + AE_TPB * sl; j < 6 * D.nfc; j += stride) D.scale_c[j] = 1.0 / (1.0 + sqrt(D.B[21 * (size_t)(j / 6) + sym6(j % 6, j % 6)]));
      if (!D.fix_points) {
        const int dg[3] = {0, 3, 5};
        for (int j = tid + AE_TPB * sl; j < 3 * D.npts; j += stride) D.scale_p[j] = 1.0 / (1.0 + sqrt(D.C[6 * (size_t)(j / 3) + dg[j % 3]]));
      }
    }
    double acc[3] = {0.0, 0.0, 0.0};                  // cost (slice 0), |x|^2 of the slice, (unused)
    if (sl == 0) for (int b = tid; b < D.nparts; b += AE_TPB) acc[0] += D.part[b];
    double gmax = 0.0;
    // (round 6: a camera's column and pose in one group, its gradient in a second; a landmark's range, position and gradient in one - read
    // where they were used, behind the `continue`s, they were four dependent round trips per camera and per landmark, and a lockstep
    // batch's one workgroup per problem walks eight slices of them one after the other)
    for (int c = tid + AE_TPB * sl; c < D.ncam; c += stride) {
      int cc = D.cam_col[c];
      double x[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k = 0; k < 7; k++) x[k] = D.poses[7 * (size_t)c + k];
      asm volatile("" : "+v"(cc)); pin8(x);
      double g[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k = 0; k < 6; k++) g[k] = D.gc[6 * (size_t)max(cc, 0) + k];
      pin8(g);
      if (cc < 0) continue;
      for (int k = 0; k < 7; k++) acc[1] += x[k] * x[k];
      for (int k = 0; k < 3; k++) gmax = fmax(gmax, fabs(g[k]));
      double d[3] = {-g[3], -g[4], -g[5]}, qn[4];
      quat_plus(x + 3, d, qn);
      for (int k = 0; k < 4; k++) gmax = fmax(gmax, fabs(x[3 + k] - qn[k]));
    }
    if (!D.fix_points)
      for (int p = tid + AE_TPB * sl; p < D.npts; p += stride) {
        int o0 = D.pt_off[p], o1 = D.pt_off[p + 1], z0_ = 0, z1_ = 0;
        double pg[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 3; k++) { pg[k] = D.pts[3 * (size_t)p + k]; pg[4 + k] = D.gp[3 * (size_t)p + k]; }
        pin4i(o0, o1, z0_, z1_); pin8(pg);
        if (o1 == o0) continue;                             // unused point: not in the reduced program
        for (int k = 0; k < 3; k++) { double v = pg[k]; acc[1] += v * v; gmax = fmax(gmax, fabs(pg[4 + k])); }
      }
    acc[2] = 0.0;
    // max-reduce gmax through the sum tree by bit tricks is not possible: do a separate max tree
    double m = gmax;
    for (int o = 32; o >= 1; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    if ((tid & 63) == 0) s_max[tid >> 6] = m;
    block_reduce_wide<3>(acc, s_red, s_out);
    if (tid == 0) {
      double gm = 0.0;
      for (int i = 0; i < AE_TPB / 64; i++) gm = fmax(gm, s_max[i]);
      if (solo) { if (sl == 0) cost = s_out[0]; x2 += s_out[1]; gm_all = fmax(gm_all, gm); }
      else {
        double* mine = D.ae_part + 4 * sl;
        __hip_atomic_store(mine + 0, s_out[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 1, s_out[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 2, gm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();                                          // (s_max, s_red, s_out are free for the next slice)
  }
  if (!solo) {
    if (tid == 0) {
      __threadfence();
      s_last = (__hip_atomic_fetch_add(D.ae_ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == AE_NS - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    if (tid == 0) {
      __threadfence();
      cost = __hip_atomic_load(D.ae_part + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int q = 0; q < AE_NS; q++) {
        x2 += __hip_atomic_load(D.ae_part + 4 * q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gm_all = fmax(gm_all, __hip_atomic_load(D.ae_part + 4 * q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      }
      __hip_atomic_store(D.ae_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (tid == 0) {
    st->x_cost = cost;
    st->x_norm = sqrt(x2);
    st->gmax = gm_all;
    if (first) st->initial_cost = cost;
    st->first = 0;
    st->need_eval = 0;
    st->e_dirty = 1;
    if (st->gmax <= 1e-10) { st->termination = 1; st->done = 1; }
  }
  ba_iter_begin_body(D);
}

// ---- iteration begin: iteration cap / minimum radius ------------------------------------------------
__device__ __forceinline__ void ba_iter_begin_body(const BaDev& D) {
  BaState* st = D.st;
  const int iteration = st->iteration, max_iters = st->max_iters;      // (all reads first: one round trip)
  const double radius = st->radius;
  if (D.cflags) for (int i = threadIdx.x; i < D.ncflags; i += blockDim.x) D.cflags[i] = 0;      // (any block size; the rest is thread 0's; harmless when the solve has just ended)
  if (threadIdx.x != 0) return;
  if (st->done) return;                                         // (thread 0's own write when the evaluation above has just ended the solve)
  st->valid = 0; st->accepted = 0; st->chol_fail = 0; st->e_dirty = 0;
  // StopFlagCallback (include/CeresOptimizer.h:332-349) runs after every iteration, before the iteration-cap test: the
  // host keeps copying the caller's flag into this pinned byte while the enqueued iterations drain
  if (D.stop_dev && __atomic_load_n(D.stop_dev, __ATOMIC_RELAXED)) { st->termination = 4; st->done = 1; return; }
  if (iteration >= max_iters) { st->termination = 0; st->done = 1; return; }
  if (radius <= 1e-32) { st->termination = 6; st->done = 1; return; }
  st->iteration = iteration + 1;
  st->valid = 1;          // provisional; cleared by a failed factorisation / non-positive model change
}
__global__ void k_ba_iter_begin(const BaDev* __restrict__ Dv) {      // (the pose-graph solver's launch; bundle adjustment: inside k_ba_after_eval)
  const BaDev D = Dv[blockIdx.y];
  if (ld_flags(D.st).done) return;
  ba_iter_begin_body(D);
}

// ---- per point: (C_s + D)^-1, scaled gradient, E and E (C_s+D)^-1 per observation -----------------
__device__ __forceinline__ bool inv3_sym6(const double* C, double* Ci) {   // C = [c00,c01,c02,c11,c12,c22]
  const double a = C[0], b = C[1], c = C[2], d = C[3], e = C[4], f = C[5];
  const double A = d * f - e * e, Bc = -(b * f - c * e), Cc = b * e - c * d;
  const double det = a * A + b * Bc + c * Cc;
  if (!(det != 0.0) || !isfinite(det)) return false;
  const double id = 1.0 / det;
  Ci[0] = A * id; Ci[1] = Bc * id; Ci[2] = Cc * id; Ci[3] = (a * f - c * c) * id; Ci[4] = -(a * e - b * c) * id; Ci[5] = (a * d - b * b) * id;
  return true;
}

// (the workgroups behind the `npt_grid` landmark workgroups zero the reduced system: k_ba_zero_S was a launch of its own)
// envelope != 0 (lockstep batches factored by k_chol_wg, which never writes a tile outside the skyline - tile_first): only the tiles
// INSIDE the envelope are cleared; the rest was zeroed once at the start of the solve (k_ba_zero_S) and nothing has touched it since.
// At C4 size that is 55 tiles of 8 KB per problem instead of 2.9 MB (185 MB per 64-problem launch).
__global__ __launch_bounds__(BA_TPB) void k_ba_schur_prep(const BaDev* __restrict__ Dv, int npt_grid, int envelope) {
  const BaDev D = Dv[blockIdx.y];
  BaState* st = D.st;
  const StFlags F = ld_flags(st);
  if (F.done || !F.valid) return;
  if ((int)blockIdx.x >= npt_grid) {
    if (envelope && D.chol_la) {
      const int nbz = D.npad / NB, np = D.npad;
      for (int i = (int)blockIdx.x - npt_grid; i < nbz; i += (int)gridDim.x - npt_grid) {
        const int f = min(D.tile_first[i], i), wd = (i - f + 1) * NB;
        for (int e = threadIdx.x; e < NB * wd; e += BA_TPB) {
          const int r = NB * i + e / wd, c = NB * f + e % wd;
          if (r < D.n6) D.S[(size_t)r * np + c] = 0.0;
        }
      }
      return;
    }
    const size_t tot = (size_t)D.n6 * D.npad, nz = (size_t)(gridDim.x - npt_grid) * BA_TPB;
    for (size_t i = (size_t)((int)blockIdx.x - npt_grid) * BA_TPB + threadIdx.x; i < tot; i += nz) D.S[i] = 0.0;
    return;
  }
  if (D.fix_points) return;
  const int p = blockIdx.x * BA_TPB + threadIdx.x;
  if (p >= D.npts) return;
  // (round 6: every input into registers first - one group, waited for once -, the stores at the end.  Written as loads and stores
  // interleaved, the compiler re-read scale_p behind every store - pointers out of the same struct may alias -: sixteen dependent round
  // trips per thread in a kernel that is 6 % of a batched iteration.  Same operations on the same doubles.)
  int po[4] = {D.pt_off[p], D.pt_off[p + 1], 0, 0};
  double in[16];                                               // scale_p (3), C (6), g_p (3), radius, -
#pragma unroll
  for (int k = 0; k < 16; k++) in[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 3; k++) { in[k] = D.scale_p[3 * (size_t)p + k]; in[9 + k] = D.gp[3 * (size_t)p + k]; }
#pragma unroll
  for (int k = 0; k < 6; k++) in[3 + k] = D.C[6 * (size_t)p + k];
  in[12] = st->radius;
  pin4i(po[0], po[1], po[2], po[3]); pin8(in); pin8(in + 8);
  if (po[0] == po[1]) return;
  const double* sp = in; const double* Cu = in + 3; const double* gpv = in + 9;
  double Cs[6] = {Cu[0] * sp[0] * sp[0], Cu[1] * sp[0] * sp[1], Cu[2] * sp[0] * sp[2], Cu[3] * sp[1] * sp[1], Cu[4] * sp[1] * sp[2], Cu[5] * sp[2] * sp[2]};
  const double radius = in[12];
  Cs[0] += fmin(fmax(Cs[0], 1e-6), 1e32) / radius;
  Cs[3] += fmin(fmax(Cs[3], 1e-6), 1e32) / radius;
  Cs[5] += fmin(fmax(Cs[5], 1e-6), 1e32) / radius;
  double Ci[6];
  if (!inv3_sym6(Cs, Ci)) { st->chol_fail = 1; for (int k = 0; k < 6; k++) Ci[k] = 0.0; }
  double o[18];                                                // Cinv (6), scaled g_p (3), N (6) + g_p (3)
#pragma unroll
  for (int k = 0; k < 6; k++) o[k] = Ci[k];
#pragma unroll
  for (int k = 0; k < 3; k++) o[6 + k] = gpv[k] * sp[k];
  o[9] = Ci[0] * sp[0] * sp[0]; o[10] = Ci[1] * sp[0] * sp[1]; o[11] = Ci[2] * sp[0] * sp[2];
  o[12] = Ci[3] * sp[1] * sp[1]; o[13] = Ci[4] * sp[1] * sp[2]; o[14] = Ci[5] * sp[2] * sp[2];
#pragma unroll
  for (int k = 0; k < 3; k++) o[15 + k] = gpv[k];
#pragma unroll
  for (int k = 0; k < 6; k++) D.Cinv[6 * (size_t)p + k] = o[k];
#pragma unroll
  for (int k = 0; k < 3; k++) D.gps[3 * (size_t)p + k] = o[6 + k];
  double* ng = D.Ng + 9 * (size_t)p;
#pragma unroll
  for (int k = 0; k < 9; k++) ng[k] = o[9 + k];
}

// X' = G (R_a N) of one observation of camera a is [Y; [r]x Y] with Y = W (R_a N) (3x3): twelve numbers {Y, r} stand for the 6x3 block.
// c = the observation's record, Ra row-major, N6 = {n00,n01,n02,n11,n12,n22}; out: yr[0..8] = Y (row-major), yr[9..11] = r
__device__ __forceinline__ void make_yr(const double* __restrict__ c, const double* __restrict__ Ra, const double* __restrict__ N6, double* __restrict__ yr) {
  const double w00 = c[0], w11 = c[1], w02 = c[2], w12 = c[3], w22 = c[4];
  double M[9];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double a0 = Ra[3 * i], a1 = Ra[3 * i + 1], a2 = Ra[3 * i + 2];
    M[3 * i] = fma(a2, N6[2], fma(a1, N6[1], a0 * N6[0]));
    M[3 * i + 1] = fma(a2, N6[4], fma(a1, N6[3], a0 * N6[1]));
    M[3 * i + 2] = fma(a2, N6[5], fma(a1, N6[4], a0 * N6[2]));
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double m0 = M[k], m1 = M[3 + k], m2 = M[6 + k];
    yr[k] = fma(w02, m2, w00 * m0);
    yr[3 + k] = fma(w12, m2, w11 * m1);
    yr[6 + k] = fma(w22, m2, fma(w12, m1, w02 * m0));
  }
  yr[9] = c[5]; yr[10] = c[6]; yr[11] = c[7];
}
// P = Y R_b^T G_b^T (3x6) of one pair: the upper three rows of the pair's 6x6 contribution; the lower three are [r_a]x P
__device__ __forceinline__ void pair_P(const double* __restrict__ yr, const double* __restrict__ Rb, const double* __restrict__ c, double (&P)[3][6]) {
  const double w00 = c[0], w11 = c[1], w02 = c[2], w12 = c[3], w22 = c[4], r0 = c[5], r1 = c[6], r2 = c[7];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double x0 = yr[3 * i], x1 = yr[3 * i + 1], x2 = yr[3 * i + 2];
    const double t0 = fma(x2, Rb[2], fma(x1, Rb[1], x0 * Rb[0]));
    const double t1 = fma(x2, Rb[5], fma(x1, Rb[4], x0 * Rb[3]));
    const double t2 = fma(x2, Rb[8], fma(x1, Rb[7], x0 * Rb[6]));
    const double z0 = fma(t2, w02, t0 * w00), z1 = fma(t2, w12, t1 * w11), z2 = fma(t2, w22, fma(t1, w12, t0 * w02));
    P[i][0] = z0; P[i][1] = z1; P[i][2] = z2;
    P[i][3] = fma(z2, r1, -(z1 * r2)); P[i][4] = fma(z0, r2, -(z2 * r0)); P[i][5] = fma(z1, r0, -(z0 * r1));
  }
}
// a36[6u+v] += the pair's 6x6 contribution (u = parameter of camera a, v = parameter of camera b)
__device__ __forceinline__ void pair_acc(const double* __restrict__ yr, const double* __restrict__ Rb, const double* __restrict__ c, double* __restrict__ a36) {
  double P[3][6];
  pair_P(yr, Rb, c, P);
  const double r0 = yr[9], r1 = yr[10], r2 = yr[11];
#pragma unroll
  for (int v = 0; v < 6; v++) {
    a36[v] += P[0][v]; a36[6 + v] += P[1][v]; a36[12 + v] += P[2][v];
    a36[18 + v] = fma(r1, P[2][v], fma(-r2, P[1][v], a36[18 + v]));
    a36[24 + v] = fma(r2, P[0][v], fma(-r0, P[2][v], a36[24 + v]));
    a36[30 + v] = fma(r0, P[1][v], fma(-r1, P[0][v], a36[30 + v]));
  }
}
// the lower triangle (21 sums, row u, column v <= u) of one diagonal pair
__device__ __forceinline__ void pair_acc_lower(const double* __restrict__ yr, const double* __restrict__ Ra, const double* __restrict__ c, double* __restrict__ acc) {
  double P[3][6];
  pair_P(yr, Ra, c, P);
  const double r0 = yr[9], r1 = yr[10], r2 = yr[11];
#pragma unroll
  for (int u = 0; u < 3; u++)
#pragma unroll
    for (int v = 0; v <= u; v++) acc[u * (u + 1) / 2 + v] += P[u][v];
#pragma unroll
  for (int v = 0; v < 6; v++) {
    if (v <= 3) acc[6 + v] = fma(r1, P[2][v], fma(-r2, P[1][v], acc[6 + v]));
    if (v <= 4) acc[10 + v] = fma(r2, P[0][v], fma(-r0, P[2][v], acc[10 + v]));
    acc[15 + v] = fma(r0, P[1][v], fma(-r1, P[0][v], acc[15 + v]));
  }
}

// ---- reduced camera system S = B_s + D - sum E (C_s+D)^-1 E^T over the non-empty block pairs (a <= b) ----------------
// ONE WORKGROUP PER BLOCK ROW (free camera a).  Every pair of the row couples an observation i of camera a with the observation j of
// the same point by a camera b >= a:  block (a, b) += X'_i R_b^T G_j^T  with  X'_i = G_i (R_a N_p)  (the factored records above).
// The workgroup forms X' ONCE per observation of camera a (camera a's records are a contiguous run: streamed), keeps the 18 values
// in LDS by list position (pair_i) and gathers only the 64-byte records of the partners.
//
// The kernel is bound by the latency of its dependent loads - at 78 KB of LDS per workgroup a SIMD holds two waves, and a gather of
// 64 scattered records takes ~3 us under load (tools/schur_prof.py, in-kernel stamps of a 64-problem C4 batch; the first version of
// this kernel: 52 us per workgroup, of which 16 the pass over camera a's list, 8 the diagonal block, 26 the off-diagonal blocks on
// the wave that drew the longest lists - the pair lists of a SLAM graph are skewed: half of C4's blocks hold <= 67 pairs, the
// neighbouring keyframes' 350 ... 490).  So the structure follows the round trips, not the flops:
//  (1) the pass over camera a's list builds the X' records, the rhs  rhs_a = g_s - S_c sum X'_i g_p  AND the diagonal block (both
//      records of a diagonal pair belong to the SAME observation - the pair list of block (a, a) is the camera's list unless the
//      camera sees a point twice; such rows walk the literal list afterwards), three list entries per thread in flight, one
//      reduction of 27 sums;
//  (2) the off-diagonal blocks are cut into SEGMENTS of <= SR_SEG = 128 pairs (host: seg[]), dealt round-robin to the four waves:
//      round r, wave w takes segment 4 r + w - a 490-pair block is four waves' work, not one's.  Both gathers of a segment
//      (2 x 64 lanes) are in flight together, the next round's go out before this round's arithmetic, the indices two rounds
//      ahead; lanes without a pair read a zero record from LDS and a valid partner (no branches around the loads).
//      A wave leaves its 36 sums (transposing wave reduction, wave_reduce.h) in a mailbox; behind the round's barrier wave 0 adds
//      the segments of a block in list order (fixed order: the result does not depend on timing) and stores block (b, a) = -(sum)^T.
// Records beyond SR_CH observations of one camera do not fit the LDS and are formed from global memory where they are used.
#ifdef ORBHIP_SCHUR_PROF
extern __device__ unsigned long long g_chol_prof[128][10];
#define SR_STAMP(col, cond) do { if (sr_prof && (cond)) g_chol_prof[a & 127][col] += __builtin_amdgcn_s_memrealtime() - sr_t0; } while (0)
#else
#define SR_STAMP(col, cond) do { } while (0)
#endif
#define SC_TPB 256
#define SR_CH 736                      /* records in LDS: (736 + 1) x 13 doubles = 76.6 KB, two workgroups per CU */
#define SR_PITCH 13                    /* {Y (9), r (3)} + 1: an odd pitch spreads the lanes' records over the banks */
#define SR_REC 12
#define SR_SEG 128                     /* pairs per segment: two gathers of 64 lanes (host: the seg[] list) */
#define SR_LDS_BYTES ((SR_CH + 1) * SR_PITCH * sizeof(double))
__global__ __launch_bounds__(SC_TPB) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ba_schur(const BaDev* __restrict__ Dv) {
  // Workgroup -> (problem, block row).  The records a row gathers belong to the cameras that share points with camera a - in a
  // SLAM map mostly the next few keyframes -, i.e. to rows that run at about the same time; the dispatcher deals consecutive
  // workgroup ids out over the 8 XCDs (one L2 each), so with the plain mapping those rows meet eight different L2s.  When the batch
  // has a multiple of 8 problems, XCD k takes the problems k, k + 8, ... whole, row after row: a problem's records then pass
  // through ONE L2 and the gathers hit it.
  int prob = blockIdx.y, a = blockIdx.x;
  if ((gridDim.y & 7) == 0) {
    const int n = blockIdx.y * gridDim.x + blockIdx.x, k = n & 7, m = n >> 3;
    prob = (m / (int)gridDim.x) * 8 + k; a = m % (int)gridDim.x;
  }
  const BaDev D = Dv[prob];
  const BaState* st = D.st;
  if (a >= D.nfc) return;
  // (round 6) NO BRANCHES AROUND LOADS in this kernel.  The compiler closes every `if` block that requested something with a wait for all
  // of it - `if (m.z >= 0) { indices of the round after next }` behind "next round's gathers" made every round wait for those gathers
  // BEFORE its arithmetic, the prologue's groups each waited for themselves (seven round trips where the dependence depth is four).  Every
  // load below is unconditional on a clamped index (a segment that does not exist reads the entry behind the list, a lane without a pair
  // its segment's first one), what it returned is sanitised where it is USED, a round or a phase later, and a group is waited for once
  // (pin8 / pin4i).  The arithmetic is untouched.
  int4 rm = ldg4(D.row_meta + 2 * a), rm2 = ldg4(D.row_meta + 2 * a + 1);          // (one round trip: not free_cams -> cam_off -> list; with the state flags)
  const StFlags F = ld_flags(st);
  pin4i(rm.x, rm.y, rm.z, rm.w); pin4i(rm2.x, rm2.y, rm2.z, rm2.w);
  if (F.done || !F.valid) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int np = D.npad;
#ifdef ORBHIP_SCHUR_PROF
  const bool sr_prof = prob == 0 && lane == 0; const unsigned long long sr_t0 = __builtin_amdgcn_s_memrealtime();
  if (sr_prof && w == 0) g_chol_prof[a & 127][9] += 1;
#endif
  extern __shared__ __attribute__((aligned(16))) double s_ec[];       // [SR_CH + 1][SR_PITCH]; record SR_CH = zeros
  __shared__ double s_w[2 * 4 * 36];                                  // (1): s_red [4][27], s_out [27]; (2): the mailboxes [2][4][36]
  __shared__ int s_mf[2][4][2];                                       // mailbox labels: {column b or -1, segment flags}
  double* s_red = s_w; double* s_out = s_w + 4 * 27;
  // (the row's labels are the same for every thread: scalars, so that what depends on them alone is a scalar decision)
  const int lo_a = __builtin_amdgcn_readfirstlane(rm2.x), n_a = __builtin_amdgcn_readfirstlane(rm2.y), ca = __builtin_amdgcn_readfirstlane(rm2.z);
  const int d_lo = __builtin_amdgcn_readfirstlane(rm.x), d_hi = __builtin_amdgcn_readfirstlane(rm.y), s_lo = __builtin_amdgcn_readfirstlane(rm.z),
            s_hi = __builtin_amdgcn_readfirstlane(rm.w);
  const bool fuse = (d_hi - d_lo) == n_a;                              // the diagonal block's pair list IS the camera's list
  const int R = (s_hi - s_lo + 3) >> 2;                                // rounds of (2)
  if (tid < SR_PITCH) s_ec[SR_CH * SR_PITCH + tid] = 0.0;
  // ---- segment bookkeeping of (2) ----
  struct Meta { int4 v; bool ok; };                                    // a segment's label as loaded; ok (a scalar): the segment exists
  struct Idx { int pj0, pj1, pi0, pi1; bool v0, v1; };                 // pair indices of a segment (two per lane) as loaded; v: the lane has that pair
  struct Cam { double qb[4]; double sb; };                             // camera b of a segment: quaternion, S_c,b of this lane's sum
  const int slot = wave_reduce36_slot(lane);                   // which of a segment's 36 sums this lane ends up holding (or -1)
  const int slot_u = (slot < 0 ? 0 : slot) / 6, slot_v = (slot < 0 ? 0 : slot) - 6 * slot_u;
  auto ld_meta = [&](int r) -> Meta {
    const int g = s_lo + w + 4 * r;
    Meta m; m.ok = g < s_hi;
    m.v = ldg4(D.seg + (m.ok ? g : s_lo));                                      // (s_lo <= the list's length: the entry behind the list is allocated)
    return m;
  };
  auto ld_idx = [&](const Meta& m) -> Idx {
    const int mx = m.ok ? m.v.x : 0, my = m.ok ? m.v.y : 0;
    const int e0 = mx + lane, e1 = mx + 64 + lane;
    Idx x; x.v0 = e0 < my; x.v1 = e1 < my;
    x.pj0 = ldg(D.pair_j + (x.v0 ? e0 : mx)); x.pj1 = ldg(D.pair_j + (x.v1 ? e1 : mx));     // (a lane without a pair gathers the segment's first partner: finite wherever the block is)
    x.pi0 = ldg(D.pair_i + (x.v0 ? e0 : mx)); x.pi1 = ldg(D.pair_i + (x.v1 ? e1 : mx));
    return x;
  };
  auto ld_cam = [&](const Meta& m) -> Cam {
    Cam x;
    const double* qb = D.poses + 7 * (size_t)(m.ok ? (m.v.w >> 2) : 0) + 3;
#pragma unroll
    for (int k = 0; k < 4; k++) x.qb[k] = ldg(qb + k);
    x.sb = ldg(D.scale_c + 6 * (size_t)(m.ok ? m.v.z : 0) + slot_v);
    return x;
  };
  auto ld_y = [&](const Meta& m, const Idx& ix, double* ya, double* yb) {
    ld_rec8g(D.E, (size_t)(m.ok ? ix.pj0 : 0), ya); ld_rec8g(D.E, (size_t)(m.ok ? ix.pj1 : 0), yb);
  };
  // ---- (1) camera a's records -> LDS, the rhs of camera a and the diagonal block (a, a) ----
  // Request order = dependence depth.  Group A hangs off the row's labels: camera a's first three list entries per thread (record + point
  // index), its quaternion, the segment labels of (2).  Group B off those: the landmarks' N and g of the first two entries, the pair indices
  // of round 0.
  const bool p1 = !D.fix_points && n_a > 0;
  const bool v0 = p1 && tid < n_a, v1 = p1 && tid + SC_TPB < n_a, v2 = p1 && tid + 2 * SC_TPB < n_a;
  double c0[8], c1[8], c2[8], g0[9], g1[9], g2[9], qa[4];
  int pt0, pt1, pt2;
  {
    const int e0 = lo_a + (v0 ? tid : 0), e1 = lo_a + (v1 ? tid + SC_TPB : 0), e2 = lo_a + (v2 ? tid + 2 * SC_TPB : 0);
    pt0 = ldg(D.cam_obs_pt + e0); pt1 = ldg(D.cam_obs_pt + e1); pt2 = ldg(D.cam_obs_pt + e2);
    ld_rec8g(D.E, (size_t)e0, c0); ld_rec8g(D.E, (size_t)e1, c1); ld_rec8g(D.E, (size_t)e2, c2);
#pragma unroll
    for (int k = 0; k < 4; k++) qa[k] = ldg(D.poses + 7 * (size_t)ca + 3 + k);
  }
  Meta m0 = ld_meta(0), m1 = ld_meta(1), m2 = ld_meta(2);
  { int z = 0; pin4i(pt0, pt1, pt2, z); }
  pin8(c0); pin8(c1); pin8(c2); pin4(qa);
  pin4i(m0.v.x, m0.v.y, m0.v.z, m0.v.w); pin4i(m1.v.x, m1.v.y, m1.v.z, m1.v.w); pin4i(m2.v.x, m2.v.y, m2.v.z, m2.v.w);
  double Ra[9];
  quat_to_R(qa, Ra);
  auto get_x = [&](int pos, double* x) {                       // (a camera with more observations than the LDS holds)
    const int e = lo_a + pos;
    double c[8];
    ld_rec8(D.E, (size_t)e, c);
    make_yr(c, Ra, D.Ng + 9 * (size_t)D.cam_obs_pt[e], x);
  };
  double acc[27];                                              // 21 lower-triangle sums of the diagonal block, 6 of the rhs
#pragma unroll
  for (int k = 0; k < 27; k++) acc[k] = 0.0;
  auto one_obs = [&](int t, bool v, const double* c, const double* ng) {
    double x[SR_REC];
    make_yr(c, Ra, ng, x);
    if (v && t < SR_CH) {
#pragma unroll
      for (int k = 0; k < SR_REC; k++) s_ec[t * SR_PITCH + k] = x[k];
    }
    // a thread without a list entry has loaded entry 0 (no branch around the loads): its Y becomes 0, the sums keep their bits
#pragma unroll
    for (int k = 0; k < 9; k++) x[k] = v ? x[k] : 0.0;
    // rhs: X' g_p = [Y g; r x (Y g)]
    const double h0 = fma(x[2], ng[8], fma(x[1], ng[7], x[0] * ng[6])), h1 = fma(x[5], ng[8], fma(x[4], ng[7], x[3] * ng[6])), h2 = fma(x[8], ng[8], fma(x[7], ng[7], x[6] * ng[6]));
    acc[21] += h0; acc[22] += h1; acc[23] += h2;
    acc[24] += fma(x[10], h2, -(x[11] * h1)); acc[25] += fma(x[11], h0, -(x[9] * h2)); acc[26] += fma(x[9], h1, -(x[10] * h0));
    if (fuse) pair_acc_lower(x, Ra, c, acc);
  };
#pragma unroll
  for (int k = 0; k < 9; k++) { g0[k] = ldg(D.Ng + 9 * (size_t)pt0 + k); g1[k] = ldg(D.Ng + 9 * (size_t)pt1 + k); }
  Idx i0 = ld_idx(m0);
  pin8(g0); pin8(g1); { double t4[4] = {g0[8], g1[8], 0.0, 0.0}; pin4(t4); g0[8] = t4[0]; g1[8] = t4[1]; }
  if (p1) {
    one_obs(tid, v0, c0, g0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 9; k++) g2[k] = ldg(D.Ng + 9 * (size_t)pt2 + k);         // (into the registers the first entry has left)
    __builtin_amdgcn_sched_barrier(0);
    one_obs(tid + SC_TPB, v1, c1, g1);
    one_obs(tid + 2 * SC_TPB, v2, c2, g2);
    for (int t = tid + 3 * SC_TPB; t < n_a; t += SC_TPB) {
      const int e = lo_a + t;
      double c[8], ng[9];
      ld_rec8(D.E, (size_t)e, c);
      const double* gp = D.Ng + 9 * (size_t)D.cam_obs_pt[e];
#pragma unroll
      for (int k = 0; k < 9; k++) ng[k] = gp[k];
      one_obs(t, true, c, ng);
    }
  }
  SR_STAMP(0, w == 0);
  __builtin_amdgcn_sched_barrier(0);
  // what the stores behind the reduction need (21 threads a diagonal entry, 6 an rhs entry): requested here, used behind the reduction
  int o_u = 0, o_v = 0;
  if (tid < 21) { while ((o_u + 1) * (o_u + 2) / 2 <= tid) o_u++; o_v = tid - o_u * (o_u + 1) / 2; }
  else if (tid < 27) o_u = tid - 21;
  const double* sca = D.scale_c + 6 * (size_t)a;
  const double ld_B = ldg(D.B + 21 * (size_t)a + (tid < 21 ? sym6(o_u, o_v) : 0)), ld_g = ldg(D.gc + 6 * (size_t)a + o_u);     // (both, every thread: no branch)
  const double ld_s = ldg(sca + o_u), ld_s2 = ldg(sca + o_v), ld_r = ldg(&st->radius);
  const double sa = ldg(sca + slot_u);                                // S_c,a of this lane's sum in (2)
  // the partner records of round 0 (requested here: the reduction below hides their round trip)
  double y0[8], y1[8];
  ld_y(m0, i0, y0, y1);
  Cam k0 = ld_cam(m0);
  Idx i1 = ld_idx(m1);
  const double o_b = tid < 21 ? ld_B : ld_g, o_s = ld_s, o_s2 = tid < 21 ? ld_s2 : 1.0, o_r = ld_r;
  if (!fuse) {                                                 // a camera that sees a point twice: the literal pair list (cross terms)
    __syncthreads();
    for (int e = d_lo + tid; e < d_hi; e += SC_TPB) {
      const int pos = D.pair_i[e];
      double x[SR_REC], y[8];
      ld_rec8(D.E, (size_t)D.pair_j[e], y);
      if (pos < SR_CH) {
#pragma unroll
        for (int k = 0; k < SR_REC; k++) x[k] = s_ec[pos * SR_PITCH + k];
      } else get_x(pos, x);
      pair_acc_lower(x, Ra, y, acc);
    }
  }
  {                                                            // the 27 sums of the workgroup (the barriers also publish the records)
    double a36[36];
#pragma unroll
    for (int k = 0; k < 36; k++) a36[k] = k < 27 ? acc[k] : 0.0;
    const double t = wave_reduce36(a36, lane);
    if (slot >= 0 && slot < 27) s_red[w * 27 + slot] = t;
    __syncthreads();
    if (tid < 27) s_out[tid] = (s_red[tid] + s_red[27 + tid]) + (s_red[2 * 27 + tid] + s_red[3 * 27 + tid]);
    __syncthreads();
  }
  SR_STAMP(1, w == 0);
  if (tid < 21) {
    double bs = o_b * o_s * o_s2;
    if (o_u == o_v) bs += fmin(fmax(bs, 1e-6), 1e32) / o_r;
    D.S[(size_t)(6 * a + o_u) * np + 6 * a + o_v] = bs - o_s * o_s2 * s_out[tid];
  } else if (tid < 27) {
    const double rv = o_b * o_s - o_s * s_out[tid];
    D.rhs[6 * a + o_u] = rv;
    D.S[(size_t)np * np + 6 * a + o_u] = rv;                   // augmented row: forward substitution rides the factorisation
  }
  __syncthreads();                                             // (s_out is read; the mailboxes share its memory)
  SR_STAMP(2, w == 0);
  // ---- (2) the off-diagonal blocks, a segment per wave and round ----
  double carry = 0.0;                                          // wave 0: the running sum of the block whose segments are arriving
  for (int r = 0; r < R; r++) {
    const int buf = r & 1;
    double mine = 0.0;
    if (m0.ok) {                                               // (a scalar)
      // next round's gathers (their indices arrived a round ago) and the indices of the round after it go out first
      double z0[8], z1[8];
      ld_y(m1, i1, z0, z1);
      const Cam k1 = ld_cam(m1);
      const Idx i2 = ld_idx(m2);
      const Meta m3 = ld_meta(r + 3);
      __builtin_amdgcn_sched_barrier(0);
      double Rb[9];
      quat_to_R(k0.qb, Rb);
      double a36[36];
#pragma unroll
      for (int k = 0; k < 36; k++) a36[k] = 0.0;
      auto pair_prod = [&](int pos, const double* y) {
        double x[SR_REC];
        if (pos < SR_CH) {
          const int rec = (pos < 0 ? SR_CH : pos) * SR_PITCH;
#pragma unroll
          for (int k = 0; k < SR_REC; k++) x[k] = s_ec[rec + k];
        } else get_x(pos, x);
        pair_acc(x, Rb, y, a36);
      };
      pair_prod(i0.v0 ? i0.pi0 : -1, y0);
      pair_prod(i0.v1 ? i0.pi1 : -1, y1);
      mine = wave_reduce36(a36, lane) * (sa * k0.sb);            // (the lane with slot k holds the total of sum k)
      if (lane == 0) { s_mf[buf][w][0] = m0.v.z; s_mf[buf][w][1] = m0.v.w; }
      m0 = m1; m1 = m2; m2 = m3; i0 = i1; i1 = i2; k0 = k1;
#pragma unroll
      for (int k = 0; k < 8; k++) { y0[k] = z0[k]; y1[k] = z1[k]; }
    } else {
      if (lane == 0) { s_mf[buf][w][0] = -1; s_mf[buf][w][1] = 0; }
      m0 = m1; m1 = m2; m2.ok = false;                         // (a wave's segments end at most one round before the row's)
    }
    if (slot >= 0) s_w[(buf * 4 + w) * 36 + slot] = mine;
    __syncthreads();
    if (w == 0 && lane < 36) {
#pragma unroll
      for (int sl = 0; sl < 4; sl++) {
        const int b = s_mf[buf][sl][0], fl = s_mf[buf][sl][1];
        if (b < 0) continue;
        const double v = s_w[(buf * 4 + sl) * 36 + lane];
        carry = (fl & 1) ? v : carry + v;
        if (fl & 2) {
          const int u = lane / 6, vv = lane - 6 * u;
          D.S[(size_t)(6 * b + vv) * np + 6 * a + u] = -carry;     // lower triangle: block (b, a) = -(sum)^T
        }
      }
    }
  }
  SR_STAMP(4 + w, true);
}

// zero the lower triangle rows of the real block (the factorisation overwrote S in place)
__global__ __launch_bounds__(256) void k_ba_zero_S(const BaDev* __restrict__ Dv, int always) {      // always: at the start of a solve (the state is not valid yet)
  const BaDev D = Dv[blockIdx.y];
  const BaState* st = D.st;
  const StFlags F = ld_flags(st);
  if (!always && (F.done || !F.valid)) return;
  const size_t tot = (size_t)D.n6 * D.npad;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) D.S[i] = 0.0;
}

// padding rows of S (identity): written once per solve - the factorisation maps them to themselves
__global__ void k_ba_pad(const BaDev* __restrict__ Dv) {
  const BaDev D = Dv[blockIdx.y];
  const int np = D.npad, n6 = D.n6;
  const int i = n6 + blockIdx.x;
  if (i >= np) return;
  for (int j = threadIdx.x; j < np; j += blockDim.x) D.S[(size_t)i * np + j] = (i == j) ? 1.0 : 0.0;
  if (threadIdx.x == 0) { D.rhs[i] = 0.0; D.S[(size_t)np * np + i] = 0.0; }
}

#include "ba_cholesky.inc"   // dense blocked Cholesky of the reduced camera system

// ---- candidate cameras: x+ = Plus(x, -y * scale); partial |dx|^2 -------------------------------------------
__global__ __launch_bounds__(BA_TPB) void k_ba_cam_update(const BaDev* __restrict__ Dv) {
  const BaDev D = Dv[blockIdx.y];
  __shared__ double s_red[4 * 2], s_out[2];
  const BaState* st = D.st;
  if ((int)blockIdx.x * BA_TPB >= D.ncam) return;
  const int c = blockIdx.x * BA_TPB + threadIdx.x;
  const bool have = c < D.ncam;
  // Two groups of loads, each waited for once (round 6): the pose and the camera's column with the state flags; then the step, the scaling,
  // the block and the gradient of that column.  Everything is computed in registers and stored at the end: with the stores to cand_poses
  // between the reads of poses (pointers the compiler cannot tell apart) every quantity was re-read behind them - ~30 waits in an 8.6 us
  // kernel on every iteration's chain.  Same operations on the same doubles.
  double x[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int cc = -1;
  {
    const size_t ca = have ? c : 0;
#pragma unroll
    for (int k = 0; k < 7; k++) x[k] = D.poses[7 * ca + k];
    cc = D.cam_col[ca];
  }
  const StFlags F = ld_flags(st);
  pin8(x); asm volatile("" : "+v"(cc));
  if (F.done || !F.valid) return;
  const bool live = have && cc >= 0 && !F.chol_fail;
  double v[40];                                       // y (6), S_c (6), g_c (6), -, -, B (21), -
#pragma unroll
  for (int k = 0; k < 40; k++) v[k] = 0.0;
  {
    const size_t cz = live ? cc : 0;
#pragma unroll
    for (int k = 0; k < 6; k++) { v[k] = D.rhs[6 * cz + k]; v[6 + k] = D.scale_c[6 * cz + k]; v[12 + k] = D.gc[6 * cz + k]; }
#pragma unroll
    for (int k = 0; k < 21; k++) v[18 + k] = D.B[21 * cz + k];
  }
  pin8(v); pin8(v + 8); pin8(v + 16); pin8(v + 24); pin8(v + 32);
  double acc[2] = {0.0, 0.0};                 // |dx|^2 of the cameras; their share of the model cost change (see k_ba_backsub)
  if (have) {
    double* xc = D.cand_poses + 7 * c;
    // the camera's record for k_ba_backsub (round 5: every observation used to walk obs_cam -> cam_col -> {y, S_c, quaternion} and
    // rebuild R and S_c y itself - one dependent level and ~40 instructions more per observation); the same products, formed once
    double* cr = D.camrec + 16 * (size_t)c;
    if (!live) {
      cr[15] = 0.0;
#pragma unroll
      for (int k = 0; k < 7; k++) xc[k] = x[k];
    } else {
      const double* y = v; const double* sc = v + 6; const double* g = v + 12; const double* Bu = v + 18;
      double Rc[9], xn[7];
      quat_to_R(x + 3, Rc);
#pragma unroll
      for (int k = 0; k < 3; k++) xn[k] = x[k] + (-y[k]) * sc[k];
      double d[3] = {(-y[3]) * sc[3], (-y[4]) * sc[4], (-y[5]) * sc[5]};
      quat_plus(x + 3, d, xn + 3);
#pragma unroll
      for (int k = 0; k < 7; k++) { double e = x[k] - xn[k]; acc[0] += e * e; }
      // -(g_c . s + s^T B_s s / 2) with the scaled step s = -y, the scaled gradient and the scaled block WITHOUT the damping
      double gs = 0.0, q = 0.0;
#pragma unroll
      for (int u = 0; u < 6; u++) {
        const double su = -y[u];
        gs += g[u] * sc[u] * su;
        double row = 0.0;
#pragma unroll
        for (int w = 0; w < 6; w++) row += Bu[sym6(u, w)] * sc[u] * sc[w] * (-y[w]);
        q += su * row;
      }
      acc[1] = -(gs + q / 2);
#pragma unroll
      for (int k = 0; k < 9; k++) cr[k] = Rc[k];
#pragma unroll
      for (int k = 0; k < 6; k++) cr[9 + k] = y[k] * sc[k];
      cr[15] = 1.0;
#pragma unroll
      for (int k = 0; k < 7; k++) xc[k] = xn[k];
    }
  }
  block_reduce<2>(acc, s_red, s_out);
  if (threadIdx.x == 0) { D.part[2 * D.nparts + blockIdx.x] = s_out[0]; D.part[5 * D.nparts + blockIdx.x] = s_out[1]; }
}

// ---- landmark back-substitution, candidate points, model cost change and |dx|^2 partials ---------------------
// One 1024-thread workgroup per 256 points.  Observations are grouped by point, so the workgroup owns the contiguous
// observation range of its points and works in three phases: (1) per observation t_i = E_i^T y_cam (all loads of the
// dependent chain obs_cam -> cam_col -> y in flight at once), (2) per point the ordered sum over its t_i, the 3x3 solve and
// the candidate point, (3) per observation the model residual.  (The first version walked each point's observations in a
// serial loop of dependent loads: 33 us per launch at C4 size.)
#ifndef BS_PTS
#define BS_PTS 256
#endif
#ifndef BS_TPB
#define BS_TPB 1024
#endif
#define BS_LDS_OBS 2048                /* observations of a workgroup's 256 points whose t_i fit its LDS (48 KB) */
__global__ __launch_bounds__(BS_TPB) void k_ba_backsub(const BaDev* __restrict__ Dv, int part_off) {
  const BaDev D = Dv[blockIdx.y];
  __shared__ double s_red[16 * 2], s_out[2];
  const BaState* st = D.st;
  const StFlags F = ld_flags(st);
  if (F.done || !F.valid) return;
  if ((int)blockIdx.x * BS_PTS >= max(D.npts, 1)) return;
  const int tid = threadIdx.x;
  const int p0 = blockIdx.x * BS_PTS, p1 = min(p0 + BS_PTS, D.npts);
  const bool ok = !st->chol_fail;
  const int olo = (p1 > p0) ? D.pt_off[p0] : 0, ohi = (p1 > p0) ? D.pt_off[p1] : 0;
  // t_i waits for phase (2) in LDS when the workgroup's observations fit (round 5: it went through global memory - 77 MB written and
  // read back per 64-problem launch at C4 size); the same doubles either way
  __shared__ double s_t3[3 * BS_LDS_OBS];
  const bool in_lds = ohi - olo <= BS_LDS_OBS;
  double acc[2] = {0.0, 0.0};              // model cost change, |dx|^2
  if (ok && !D.fix_points)
    for (int i = olo + tid; i < ohi; i += BS_TPB) {
      const int c = D.obs_cam[i];
      double t[3] = {0.0, 0.0, 0.0};
      double cr[16], e[8];
      {                                                       // the camera's record (one line) and the observation's, requested together
        const double2* m = (const double2*)(D.camrec + 16 * (size_t)c);
#pragma unroll
        for (int k = 0; k < 8; k++) { const double2 v = m[k]; cr[2 * k] = v.x; cr[2 * k + 1] = v.y; }
        ld_rec8(D.E, (size_t)D.cam_pos[i], e);
      }
      if (cr[15] != 0.0) {
        // E^T y of the factored record (k_ba_eval): S_p R^T W (yt - r x yw), y~ = S_c y; the S_p factor is applied per point below
        const double* Rc = cr;
        const double yt0 = cr[9], yt1 = cr[10], yt2 = cr[11], yw0 = cr[12], yw1 = cr[13], yw2 = cr[14];
        const double r0 = e[5], r1 = e[6], r2 = e[7];
        const double d0 = yt0 - (r1 * yw2 - r2 * yw1), d1 = yt1 - (r2 * yw0 - r0 * yw2), d2 = yt2 - (r0 * yw1 - r1 * yw0);
        const double q0 = fma(e[2], d2, e[0] * d0), q1 = fma(e[3], d2, e[1] * d1), q2 = fma(e[4], d2, fma(e[3], d1, e[2] * d0));
#pragma unroll
        for (int v = 0; v < 3; v++) t[v] = fma(Rc[6 + v], q2, fma(Rc[3 + v], q1, Rc[v] * q0));
      }
      if (in_lds) { s_t3[3 * (i - olo)] = t[0]; s_t3[3 * (i - olo) + 1] = t[1]; s_t3[3 * (i - olo) + 2] = t[2]; }
      else { D.t3[3 * (size_t)i] = t[0]; D.t3[3 * (size_t)i + 1] = t[1]; D.t3[3 * (size_t)i + 2] = t[2]; }
    }
  __syncthreads();
  if (tid < BS_PTS) {
    const int p = p0 + tid;
    if (p < p1) {
      const int lo = D.pt_off[p], hi = D.pt_off[p + 1];
      if (ok && !D.fix_points && lo < hi) {
        const double g0 = D.gps[3 * (size_t)p], g1 = D.gps[3 * (size_t)p + 1], g2 = D.gps[3 * (size_t)p + 2];
        double T0 = 0.0, T1 = 0.0, T2 = 0.0;                    // sum over the point's observations of E_i^T y_cam, in observation order
        double t[3] = {g0, g1, g2};                             // g_p - sum: subtracted one by one, as before
        const double* spp = D.scale_p + 3 * (size_t)p;
        for (int i = lo; i < hi; i++) {
          const double* tq = in_lds ? s_t3 + 3 * (i - olo) : D.t3 + 3 * (size_t)i;
          const double a0 = tq[0] * spp[0], a1 = tq[1] * spp[1], a2 = tq[2] * spp[2];
          t[0] -= a0; t[1] -= a1; t[2] -= a2; T0 += a0; T1 += a1; T2 += a2;
        }
        const double* Ci = D.Cinv + 6 * (size_t)p;
        const double yp0 = Ci[0] * t[0] + Ci[1] * t[1] + Ci[2] * t[2];
        const double yp1 = Ci[1] * t[0] + Ci[3] * t[1] + Ci[4] * t[2];
        const double yp2 = Ci[2] * t[0] + Ci[4] * t[1] + Ci[5] * t[2];
        const double stp[3] = {-yp0, -yp1, -yp2};
        const double* sp = D.scale_p + 3 * (size_t)p;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const double xo = D.pts[3 * (size_t)p + k], xn = xo + stp[k] * sp[k];
          D.cand_pts[3 * (size_t)p + k] = xn;
          const double e = xo - xn; acc[1] += e * e;
        }
        // The point's share of the model cost change -(g . s + s^T H s / 2) (Ceres: -sum over the residual blocks of
        // m . (r + m / 2), m = J s - the same quadratic form; round 4: it was summed per observation from 160 bytes of stored
        // Jacobians each): g_p . s_p + s_p^T C_s s_p / 2 + s_p . sum_i E_i^T s_c(i), with the scaled step s_p, the scaled landmark
        // block WITHOUT the damping and s_c = -y, i.e. the last term is -s_p . T.  The cameras' share comes from k_ba_cam_update.
        const double* Cu = D.C + 6 * (size_t)p;
        const double c00 = Cu[0] * sp[0] * sp[0], c01 = Cu[1] * sp[0] * sp[1], c02 = Cu[2] * sp[0] * sp[2], c11 = Cu[3] * sp[1] * sp[1], c12 = Cu[4] * sp[1] * sp[2],
                     c22 = Cu[5] * sp[2] * sp[2];
        const double gs = g0 * stp[0] + g1 * stp[1] + g2 * stp[2];
        const double q = stp[0] * (c00 * stp[0] + c01 * stp[1] + c02 * stp[2]) + stp[1] * (c01 * stp[0] + c11 * stp[1] + c12 * stp[2]) +
                         stp[2] * (c02 * stp[0] + c12 * stp[1] + c22 * stp[2]);
        const double cross = -(stp[0] * T0 + stp[1] * T1 + stp[2] * T2);
        acc[0] = -(gs + q / 2 + cross);
      } else {
        for (int k = 0; k < 3; k++) D.cand_pts[3 * (size_t)p + k] = D.pts[3 * (size_t)p + k];
      }
    }
  }
  block_reduce_wide<2>(acc, s_red, s_out);
  if (tid == 0) { D.part[3 * D.nparts + blockIdx.x] = s_out[0]; D.part[4 * D.nparts + blockIdx.x] = s_out[1]; }
}

// ---- iteration end: Ceres' step evaluation ------------------------------------------------------
__global__ __launch_bounds__(BA_TPB) void k_ba_iter_end(const BaDev* __restrict__ Dv) {
  const BaDev D = Dv[blockIdx.y];
  const int nb_obs = max((D.nobs + BA_TPB - 1) / BA_TPB, 1), nb_cam = (D.ncam + BA_TPB - 1) / BA_TPB, nb_pt = max((D.npts + BA_TPB - 1) / BA_TPB, 1);
  __shared__ double s_red[4 * 3], s_out[3];
  BaState* st = D.st;
  // (round 6: what thread 0 needs of the state is requested with the flags - read where it was used, behind the stores to the same
  // struct, every field was a round trip of its own: twelve in a 4.8 us kernel on every iteration's chain)
  int invalid_steps = st->invalid_steps, successful_steps = st->successful_steps;
  double sv[4] = {st->radius, st->decrease_factor, st->x_norm, st->x_cost};
  const StFlags F = ld_flags(st);
  asm volatile("" : "+v"(invalid_steps), "+v"(successful_steps)); pin4(sv);
  if (F.done || !F.valid) return;
  const int tid = threadIdx.x;
  double acc[3] = {0.0, 0.0, 0.0};           // candidate cost, model cost change, |dx|^2
  for (int b = tid; b < nb_obs; b += BA_TPB) acc[0] += D.part[D.nparts + b];
  for (int b = tid; b < nb_pt; b += BA_TPB) { acc[1] += D.part[3 * D.nparts + b]; acc[2] += D.part[4 * D.nparts + b]; }
  for (int b = tid; b < nb_cam; b += BA_TPB) { acc[2] += D.part[2 * D.nparts + b]; acc[1] += D.part[5 * D.nparts + b]; }
  block_reduce<3>(acc, s_red, s_out);
  if (tid != 0) return;
  const double radius = sv[0], decrease_factor = sv[1], x_norm = sv[2], x_cost = sv[3];
  const double mcc = s_out[1];
  if (F.chol_fail == 2) {               // a wait inside a persistent factorisation ran out of time: a scheduling problem, not arithmetic
    st->valid = 0; st->termination = 7; st->done = 1;                   // (the iterate and the radius stay as they are; the entry point returns ORBHIP_ETIMEOUT)
    return;
  }
  if (F.chol_fail || !(mcc > 0.0)) {                                    // HandleInvalidStep
    st->valid = 0;
    st->invalid_steps = invalid_steps + 1;
    if (invalid_steps + 1 >= 5) { st->termination = 5; st->done = 1; }
    st->radius = radius / decrease_factor; st->decrease_factor = decrease_factor * 2;
    return;
  }
  st->invalid_steps = 0;
  double cand_cost = s_out[0];
  if (!isfinite(cand_cost)) cand_cost = DBL_MAX;
  st->cand_cost = cand_cost; st->model_cost_change = mcc; st->step_norm2 = s_out[2];
  if (sqrt(s_out[2]) <= 1e-8 * (x_norm + 1e-8)) { st->termination = 2; st->done = 1; return; }
  const double cost_change = x_cost - cand_cost;
  if (fabs(cost_change) <= 1e-6 * x_cost) { st->termination = 3; st->done = 1; return; }
  const double rel = cost_change / mcc;
  if (rel > 1e-3) {
    st->accepted = 1; st->successful_steps = successful_steps + 1; st->need_eval = 1;
    st->radius = fmin(1e16, radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3)));
    st->decrease_factor = 2.0;
  } else {
    st->radius = radius / decrease_factor; st->decrease_factor = decrease_factor * 2.0;
  }
}

__global__ __launch_bounds__(BA_TPB) void k_ba_apply(const BaDev* __restrict__ Dv) {
  const BaDev D = Dv[blockIdx.y];
  const BaState* st = D.st;
  const StFlags F = ld_flags(st);
  if (F.done || !F.accepted) return;
  const int i = blockIdx.x * BA_TPB + threadIdx.x;
  const bool a = i < 7 * D.ncam, b = i < 3 * D.npts;
  double v2[4] = {D.cand_poses[a ? i : 0], (D.npts > 0) ? D.cand_pts[b ? i : 0] : 0.0, 0.0, 0.0};      // (both loads before either store)
  pin4(v2);
  if (a) D.poses[i] = v2[0];
  if (b) D.pts[i] = v2[1];
}

// ---- LocalBA outlier classification on the final poses / points (src/CeresOptimizer.cc:529-567): chi2 > 5.991 or
// non-positive depth, only for observations of local keyframes.  Same check_outlier() as the host path (no contraction:
// bit-identical decisions); flags are written in device (point-grouped) order.
__global__ __launch_bounds__(BA_TPB) void k_ba_classify(const BaDev* __restrict__ Dv) {
  const BaDev D = Dv[blockIdx.y];
  const int i = blockIdx.x * BA_TPB + threadIdx.x;
  if (i >= D.nobs || !D.erase) return;
  const int c = D.obs_cam[i];
  unsigned char e = 0;
  if (D.cam_local[c]) {
    double depth;
    const int out = check_outlier(D.K4 + 4 * c, D.poses + 7 * c, D.pts + 3 * (size_t)D.obs_pt[i], D.obs_uv[2 * (size_t)i], D.obs_uv[2 * (size_t)i + 1],
                                  D.obs_w[i], 5.991, &depth);
    e = (out || depth <= 0) ? 1 : 0;
  }
  D.erase[i] = e;
}

// launched: the iterations the host has enqueued - the beginning of the NEXT one is the tail of an iteration's last kernel (k_ba_after_eval),
// so a solve stopped by the host may have begun an iteration that never ran: it does not count
__global__ void k_ba_user_stop(const BaDev* __restrict__ Dv, int launched) {
  const BaDev D = Dv[blockIdx.y];
  BaState* st = D.st;
  if (!st->done) { if (st->iteration > launched) st->iteration = launched; st->termination = 4; st->done = 1; }
}


#include "ba_posegraph.inc"   // OptimizeEssentialGraph and the batch bookkeeping kernels

}  // namespace orbhip

using namespace orbhip;

#include "ba_host.inc"   // host driver

extern "C" {

int ba_check_outlier(const double* K4, const double* pose7, const double* Xw, const double* uv, double inv_sigma2,
                     double thres, double* depth) {
  return check_outlier(K4, pose7, Xw, uv[0], uv[1], inv_sigma2, thres, depth);
}

int ba_solve(const double* K4, double* poses7, const uint8_t* cam_fixed, int ncam, double* pts3, int npts,
             const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv, const double* obs_weight,
             const uint8_t* obs_robust, int nobs, const ba_options* opts, ba_summary* summary) {
  return ba_solve_impl(K4, poses7, cam_fixed, ncam, pts3, npts, obs_cam, obs_pt, obs_uv, obs_weight, obs_robust, nobs, opts, summary);
}

int ba_pose_optimization_batch_device(const double* d_K4, double* d_pose7, const double* d_Xw, const double* d_uv,
                                      const float* d_inv_sigma2, const int32_t* d_offsets, int nproblems,
                                      uint8_t* d_outlier, int32_t* d_n_inliers, ba_summary* d_summary, void* stream) {
  ORBHIP_REQUIRE(nproblems >= 0, ORBHIP_EINVAL, "bad size");
  if (nproblems == 0) return 0;
  ORBHIP_REQUIRE(d_K4 && d_pose7 && d_Xw && d_uv && d_inv_sigma2 && d_offsets && d_outlier && d_n_inliers, ORBHIP_EINVAL, "NULL argument");
  hipLaunchKernelGGL(k_pose_lm, dim3(nproblems), dim3(256), 0, (hipStream_t)stream, d_K4, d_pose7, d_Xw, d_uv, d_inv_sigma2,
                     d_offsets, d_outlier, d_n_inliers, d_summary, 100, sqrt(5.991));   // :296, :300
  ORBHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

int ba_pose_optimization(const double* K4, double* pose7, const double* Xw, const double* uv, const float* inv_sigma2,
                         int n, uint8_t* outlier, int* n_inliers, ba_summary* summary) {
  ORBHIP_REQUIRE(K4 && pose7 && n_inliers && n >= 0, ORBHIP_EINVAL, "NULL argument");
  *n_inliers = 0;
  if (n < 3) return 0;                                         // pose untouched (:330)
  ORBHIP_REQUIRE(Xw && uv && inv_sigma2 && outlier, ORBHIP_EINVAL, "NULL argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (the HIP path has no CPU fallback)"); return ORBHIP_ENODEV; }
  if (int rcd = use_default_device()) return rcd;
  HostBA H; int rc = 0;
  int offs[2] = {0, n};
  double* dK = H.upload(K4, 4, &rc); double* dP = H.upload(pose7, 7, &rc); double* dX = H.upload(Xw, 3 * (size_t)n, &rc);
  double* dU = H.upload(uv, 2 * (size_t)n, &rc); float* dS = H.upload(inv_sigma2, n, &rc); int* dO = H.upload(offs, 2, &rc);
  uint8_t* dOut = H.alloc<uint8_t>(n, &rc); int* dN = H.alloc<int>(1, &rc); ba_summary* dSum = H.alloc<ba_summary>(1, &rc);
  if (rc) return rc;
  rc = ba_pose_optimization_batch_device(dK, dP, dX, dU, dS, dO, 1, dOut, dN, dSum, nullptr);
  if (rc) return rc;
  ORBHIP_CHECK_HIP(hipMemcpy(pose7, dP, 7 * sizeof(double), hipMemcpyDeviceToHost));
  ORBHIP_CHECK_HIP(hipMemcpy(outlier, dOut, n, hipMemcpyDeviceToHost));
  ORBHIP_CHECK_HIP(hipMemcpy(n_inliers, dN, sizeof(int), hipMemcpyDeviceToHost));
  if (summary) ORBHIP_CHECK_HIP(hipMemcpy(summary, dSum, sizeof(ba_summary), hipMemcpyDeviceToHost));
  return 0;
}

int ba_optimize_sim3_batch_device(const double* d_K1, const double* d_K2, double* d_s12, const double* d_P3D2c, const double* d_obs1,
                                  const float* d_inv_sigma2_1, const double* d_P3D1c, const double* d_obs2,
                                  const float* d_inv_sigma2_2, const int32_t* d_offsets, const double* d_th2, int nproblems,
                                  uint8_t* d_outlier, int32_t* d_n_inliers, ba_summary* d_summary, void* stream) {
  ORBHIP_REQUIRE(nproblems >= 0, ORBHIP_EINVAL, "bad size");
  if (nproblems == 0) return 0;
  ORBHIP_REQUIRE(d_K1 && d_K2 && d_s12 && d_offsets && d_th2 && d_n_inliers, ORBHIP_EINVAL, "NULL argument");
  hipLaunchKernelGGL(k_sim3_lm, dim3(nproblems), dim3(256), 0, (hipStream_t)stream, d_K1, d_K2, d_s12, d_P3D2c, d_obs1, d_inv_sigma2_1,
                     d_P3D1c, d_obs2, d_inv_sigma2_2, d_offsets, d_th2, d_outlier, d_n_inliers, d_summary, 100);   // :625
  ORBHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

int ba_optimize_sim3(const double* K1, const double* K2, double* s12, const double* P3D2c, const double* obs1,
                     const float* inv_sigma2_1, const double* P3D1c, const double* obs2, const float* inv_sigma2_2, int n,
                     double th2, int fix_scale, uint8_t* outlier, int* n_inliers, ba_summary* summary) {
  (void)fix_scale;                                             // the reference never reads bFixScale (:604)
  ORBHIP_REQUIRE(K1 && K2 && s12 && n_inliers && n >= 0 && th2 > 0.0, ORBHIP_EINVAL, "NULL argument");
  ORBHIP_REQUIRE(n == 0 || (P3D2c && obs1 && inv_sigma2_1 && P3D1c && obs2 && inv_sigma2_2), ORBHIP_EINVAL, "NULL correspondence arrays");
  *n_inliers = 0;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (the HIP path has no CPU fallback)"); return ORBHIP_ENODEV; }
  if (int rcd = use_default_device()) return rcd;
  HostBA H; int rc = 0;
  int offs[2] = {0, n};
  double* dK1 = H.upload(K1, 4, &rc); double* dK2 = H.upload(K2, 4, &rc); double* dS = H.upload(s12, 7, &rc);
  double* dP2 = H.upload(P3D2c, 3 * (size_t)n, &rc); double* dO1 = H.upload(obs1, 2 * (size_t)n, &rc); float* dW1 = H.upload(inv_sigma2_1, n, &rc);
  double* dP1 = H.upload(P3D1c, 3 * (size_t)n, &rc); double* dO2 = H.upload(obs2, 2 * (size_t)n, &rc); float* dW2 = H.upload(inv_sigma2_2, n, &rc);
  int* dOff = H.upload(offs, 2, &rc); double* dTh = H.upload(&th2, 1, &rc);
  uint8_t* dOut = H.alloc<uint8_t>(n, &rc); int* dN = H.alloc<int>(1, &rc); ba_summary* dSum = H.alloc<ba_summary>(1, &rc);
  if (rc) return rc;
  rc = ba_optimize_sim3_batch_device(dK1, dK2, dS, dP2, dO1, dW1, dP1, dO2, dW2, dOff, dTh, 1, dOut, dN, dSum, nullptr);
  if (rc) return rc;
  ORBHIP_CHECK_HIP(hipMemcpy(s12, dS, 7 * sizeof(double), hipMemcpyDeviceToHost));
  if (outlier && n) ORBHIP_CHECK_HIP(hipMemcpy(outlier, dOut, n, hipMemcpyDeviceToHost));
  ORBHIP_CHECK_HIP(hipMemcpy(n_inliers, dN, sizeof(int), hipMemcpyDeviceToHost));
  if (summary) ORBHIP_CHECK_HIP(hipMemcpy(summary, dSum, sizeof(ba_summary), hipMemcpyDeviceToHost));
  return 0;
}

// Sim(3) helpers a host binding needs to cross the Sophus boundary without Sophus (host arithmetic, no device work).
int ba_sim3_exp(const double* tangent7, double* s12_out) { ORBHIP_REQUIRE(tangent7 && s12_out, ORBHIP_EINVAL, "NULL argument"); s3_exp(tangent7, s12_out); return 0; }
int ba_sim3_log(const double* s12, double* tangent7_out) { ORBHIP_REQUIRE(s12 && tangent7_out, ORBHIP_EINVAL, "NULL argument"); s3_log(s12, tangent7_out); return 0; }
int ba_sim3_mul(const double* a, const double* b, double* out) {
  ORBHIP_REQUIRE(a && b && out, ORBHIP_EINVAL, "NULL argument");
  double r[7]; s3_mul(a, b, r); for (int k = 0; k < 7; k++) out[k] = r[k]; return 0;
}
int ba_sim3_inverse(const double* a, double* out) {
  ORBHIP_REQUIRE(a && out, ORBHIP_EINVAL, "NULL argument");
  double r[7]; s3_inverse(a, r); for (int k = 0; k < 7; k++) out[k] = r[k]; return 0;
}

// LocalBundleAdjustment's optimisation core for a batch of independent local maps: pass 1 of every problem in one
// lockstep batch, host-side classification, pass 2 likewise (src/CeresOptimizer.cc:408-598 per problem).
int ba_local_bundle_adjustment_batch(const ba_local_problem* problems, int nproblems, const volatile uint8_t* stop_flag,
                                     int duplicate_blocks, int* aborted, ba_summary* pass1, ba_summary* pass2) {
  ORBHIP_REQUIRE(problems && aborted && nproblems > 0, ORBHIP_EINVAL, "NULL argument");
  *aborted = 0;
  const bool timing = std::getenv("ORBHIP_BA_TIMING") != nullptr;
  const double tt0 = ba_now_ms();
  struct Work {
    std::vector<double> P0, X0, uv, w; std::vector<int32_t> oc, op; std::vector<uint8_t> rob, erase;
  };
  std::vector<Work> W(nproblems);
  std::vector<BaInputs> in(nproblems);
  for (int q = 0; q < nproblems; q++) {
    const ba_local_problem& L = problems[q];
    ORBHIP_REQUIRE(L.K4 && L.poses7 && L.cam_fixed && L.cam_local && L.obs_erase && L.ncam > 0 && L.nobs >= 0 && L.npts >= 0, ORBHIP_EINVAL, "NULL argument");
    ORBHIP_REQUIRE(L.nobs == 0 || (L.obs_cam && L.obs_pt && L.obs_uv && L.obs_inv_sigma2), ORBHIP_EINVAL, "NULL observation arrays");
    ORBHIP_REQUIRE(L.npts == 0 || L.pts3, ORBHIP_EINVAL, "NULL argument");
    Work& w = W[q];
    w.P0.assign(L.poses7, L.poses7 + 7 * (size_t)L.ncam); w.X0.assign(L.pts3, L.pts3 + 3 * (size_t)L.npts);
    if (!duplicate_blocks) {                                                // (pass 2 filters these lists; with the re-added blocks the caller's arrays are used as they are)
      w.oc.assign(L.obs_cam, L.obs_cam + L.nobs); w.op.assign(L.obs_pt, L.obs_pt + L.nobs);
      w.uv.assign(L.obs_uv, L.obs_uv + 2 * (size_t)L.nobs);
    }
    w.w.resize(L.nobs);
    for (int i = 0; i < L.nobs; i++) {
      ORBHIP_REQUIRE(L.obs_cam[i] >= 0 && L.obs_cam[i] < L.ncam && L.obs_pt[i] >= 0 && L.obs_pt[i] < L.npts, ORBHIP_EINVAL, "observation index out of range");
      w.w[i] = (double)L.obs_inv_sigma2[i];                                // F7: weight = invSigma2
    }
    w.rob.assign(L.nobs, 1); w.erase.assign(L.nobs, 0);
  }
  auto classify = [&](int q) {                                              // :529-567
    const ba_local_problem& L = problems[q]; Work& w = W[q];
    for (int i = 0; i < L.nobs; i++) {
      w.erase[i] = 0;
      const int c = L.obs_cam[i];
      if (!L.cam_local[c]) continue;
      double depth;
      int out = check_outlier(L.K4 + 4 * c, w.P0.data() + 7 * c, w.X0.data() + 3 * (size_t)L.obs_pt[i], L.obs_uv[2 * (size_t)i], L.obs_uv[2 * (size_t)i + 1],
                              (double)L.obs_inv_sigma2[i], 5.991, &depth);
      if (out || depth <= 0) w.erase[i] = 1;
    }
  };
  auto bind = [&]() {
    for (int q = 0; q < nproblems; q++) {
      const ba_local_problem& L = problems[q]; Work& w = W[q];
      if (duplicate_blocks) {
        in[q] = BaInputs{L.K4, w.P0.data(), L.cam_fixed, L.ncam, w.X0.data(), L.npts, L.obs_cam, L.obs_pt, L.obs_uv, w.w.data(), w.rob.data(), L.nobs};
        in[q].cam_local = L.cam_local;                         // both passes see every observation: the outlier test runs on the device
      } else {
        in[q] = BaInputs{L.K4, w.P0.data(), L.cam_fixed, L.ncam, w.X0.data(), L.npts, w.oc.data(), w.op.data(), w.uv.data(), w.w.data(), w.rob.data(), (int)w.oc.size()};
      }
    }
  };
  std::vector<uint8_t*> erase_ptrs(nproblems);
  for (int q = 0; q < nproblems; q++) erase_ptrs[q] = W[q].erase.data();
  uint8_t* const* erase_dev = duplicate_blocks ? erase_ptrs.data() : nullptr;
  if (stop_flag && *stop_flag) { *aborted = 1; return 0; }                  // :509-512
  ba_options o1; o1.max_iterations = 5; o1.huber_delta = sqrt(5.991); o1.fix_points = 0; o1.stop_flag = stop_flag;
  bind();
  const double tt1 = ba_now_ms();
  int rc = ba_solve_batch_impl(in.data(), nproblems, &o1, pass1, false, erase_dev);
  if (rc) return rc;
  const double tt2 = ba_now_ms();
  for (int q = 0; q < nproblems; q++) {
    const ba_local_problem& L = problems[q]; Work& w = W[q];
    if (!duplicate_blocks) classify(q);
    if (duplicate_blocks) {
      // F6: the reference adds every kept observation AGAIN without loss on the same problem; a kept observation and its
      // twin are folded into one block (obs_robust = 2, see reproj_eval) instead of being listed twice
      for (int i = 0; i < L.nobs; i++) w.rob[i] = w.erase[i] ? 1 : 2;
    } else {
      w.oc.clear(); w.op.clear(); w.uv.clear(); w.w.clear(); w.rob.clear();
      for (int i = 0; i < L.nobs; i++) {
        if (w.erase[i]) continue;
        w.oc.push_back(L.obs_cam[i]); w.op.push_back(L.obs_pt[i]); w.uv.push_back(L.obs_uv[2 * (size_t)i]); w.uv.push_back(L.obs_uv[2 * (size_t)i + 1]);
        w.w.push_back((double)L.obs_inv_sigma2[i]); w.rob.push_back(0);
      }
    }
  }
  if (stop_flag && *stop_flag) { *aborted = 1; return 0; }
  ba_options o2 = o1; o2.max_iterations = 10;
  bind();
  const double tt3 = ba_now_ms();
  rc = ba_solve_batch_impl(in.data(), nproblems, &o2, pass2, duplicate_blocks != 0, erase_dev);      // same observation set: structure reused
  if (rc) return rc;
  const double tt4 = ba_now_ms();
  for (int q = 0; q < nproblems; q++) {
    const ba_local_problem& L = problems[q]; Work& w = W[q];
    if (!duplicate_blocks) classify(q);
    if (L.nobs) std::memcpy(L.obs_erase, w.erase.data(), L.nobs);
    for (int c = 0; c < L.ncam; c++) {                                      // Matrix_7_1_ToMatrix4d normalises (:80-81)
      double* qd = w.P0.data() + 7 * c + 3;
      const double nq = std::sqrt(qd[0] * qd[0] + qd[1] * qd[1] + qd[2] * qd[2] + qd[3] * qd[3]);
      for (int k = 0; k < 4; k++) qd[k] /= nq;
    }
    std::memcpy(L.poses7, w.P0.data(), sizeof(double) * 7 * L.ncam);
    if (L.npts) std::memcpy(L.pts3, w.X0.data(), sizeof(double) * 3 * L.npts);
  }
  if (timing) fprintf(stderr, "[ba_local_batch] setup %.2f ms, pass 1 %.2f, between %.2f, pass 2 %.2f, write-back %.2f\n", tt1 - tt0, tt2 - tt1, tt3 - tt2, tt4 - tt3, ba_now_ms() - tt4);
  return 0;
}

int ba_local_bundle_adjustment(const double* K4, double* poses7, const uint8_t* cam_fixed, const uint8_t* cam_local, int ncam,
                               double* pts3, int npts, const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv,
                               const float* obs_inv_sigma2, int nobs, const volatile uint8_t* stop_flag, int duplicate_blocks,
                               uint8_t* obs_erase, int* aborted, ba_summary* pass1, ba_summary* pass2) {
  ORBHIP_REQUIRE(K4 && poses7 && cam_fixed && cam_local && obs_erase && aborted && ncam > 0 && nobs >= 0, ORBHIP_EINVAL, "NULL argument");
  ba_local_problem L{K4, poses7, cam_fixed, cam_local, ncam, pts3, npts, obs_cam, obs_pt, obs_uv, obs_inv_sigma2, nobs, obs_erase};
  return ba_local_bundle_adjustment_batch(&L, 1, stop_flag, duplicate_blocks, aborted, pass1, pass2);
}

int ba_optimize_essential_graph(double* lie7, const uint8_t* kf_fixed, int n_kf, const int32_t* edge_j, const int32_t* edge_i,
                                const double* edge_Sji, int n_edges, int max_iterations, const volatile uint8_t* stop_flag, ba_summary* summary) {
  return pg_solve_impl(lie7, kf_fixed, n_kf, edge_j, edge_i, edge_Sji, n_edges, max_iterations, stop_flag, summary);
}

int ba_essential_graph_correct(const double* lie7_orig, const double* lie7_opt, int n_kf, double* Tiw, const int32_t* pt_ref_kf, double* pts3, int npts) {
  ORBHIP_REQUIRE(lie7_orig && lie7_opt && n_kf > 0 && Tiw && npts >= 0 && (npts == 0 || (pt_ref_kf && pts3)), ORBHIP_EINVAL, "NULL argument");
  for (int p = 0; p < npts; p++) ORBHIP_REQUIRE(pt_ref_kf[p] >= 0 && pt_ref_kf[p] < n_kf, ORBHIP_EINVAL, "reference keyframe out of range");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (the HIP path has no CPU fallback)"); return ORBHIP_ENODEV; }
  if (int rcd = use_default_device()) return rcd;
  hipStream_t s = thread_stream();
  HostBA H; int rc = 0;
  const double* d0 = H.upload(lie7_orig, 7 * (size_t)n_kf, &rc, s); const double* d1 = H.upload(lie7_opt, 7 * (size_t)n_kf, &rc, s);
  double* dT = H.alloc<double>(12 * (size_t)n_kf, &rc);
  const int* dref = H.upload(pt_ref_kf, npts, &rc, s); double* dp = H.upload(pts3, 3 * (size_t)npts, &rc, s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_pg_poses, dim3((n_kf + 127) / 128), dim3(128), 0, s, d1, n_kf, dT);
  if (npts) hipLaunchKernelGGL(k_pg_points, dim3((npts + 255) / 256), dim3(256), 0, s, d0, d1, dref, dp, npts);
  ORBHIP_CHECK_HIP(hipGetLastError());
  ORBHIP_CHECK_HIP(hipMemcpyAsync(Tiw, dT, 12 * (size_t)n_kf * sizeof(double), hipMemcpyDeviceToHost, s));
  if (npts) ORBHIP_CHECK_HIP(hipMemcpyAsync(pts3, dp, 3 * (size_t)npts * sizeof(double), hipMemcpyDeviceToHost, s));
  ORBHIP_CHECK_HIP(hipStreamSynchronize(s));
  return 0;
}

// ---- measurement hook: device time of this host thread's ba_solve / ba_local_bundle_adjustment calls ---------------------
int ba_set_profiling(int enable) { g_ba_profiling.store(enable ? 1 : 0); return 0; }
int ba_set_wait_limit_ms(double ms) {
  if (int rc = use_default_device()) return rc;
  const unsigned long long v = (ms > 0.0 && ms < 1e9) ? std::max<unsigned long long>(1ull, (unsigned long long)(ms * 1e5)) : 500000000ull;      // 10-ns ticks
  ORBHIP_CHECK_HIP(hipDeviceSynchronize());
  ORBHIP_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_cp_wait_ticks), &v, sizeof(v)));
  return 0;
}
int ba_get_last_plan(int32_t* out12) {
  ORBHIP_REQUIRE(out12, ORBHIP_EINVAL, "NULL argument");
  const BaPlan& p = g_last_plan;
  const int32_t v[12] = {p.ny, p.la_form, p.tl_form, p.bsolve_form, p.npad_la, p.npad_2l, p.band, p.persist_nwg, p.persist_mode, p.la_large, p.wg_ok, 0};
  std::memcpy(out12, v, sizeof(v));
  return 0;
}
int ba_get_profile(double* device_ms, int* nsolves, int* lm_iterations) {
  if (device_ms) *device_ms = g_prof_ms;
  if (nsolves) *nsolves = g_prof_solves;
  if (lm_iterations) *lm_iterations = g_prof_iters;
  g_prof_ms = 0.0; g_prof_solves = 0; g_prof_iters = 0;
  return 0;
}

#ifdef ORBHIP_CHOL_PROF
int ba_debug_pose_ticks(unsigned long long* out, int reset) {
  ORBHIP_CHECK_HIP(hipDeviceSynchronize());
  if (out) ORBHIP_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pose_ticks), 64));
  if (reset) { static unsigned long long z[8]; ORBHIP_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_pose_ticks), z, 64)); }
  return 0;
}
int ba_debug_df_stamps(unsigned long long* out16, int reset) {      // g_df_stamp[4][4]
  ORBHIP_CHECK_HIP(hipDeviceSynchronize());
  if (out16) ORBHIP_CHECK_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_df_stamp), 128));
  if (reset) { static unsigned long long z[16]; ORBHIP_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_df_stamp), z, 128)); }
  return 0;
}
int ba_debug_factor_nw(const double* A, double* X, int n, int nw, unsigned long long* ticks2) {     // host pointers, 32 x 32 row-major; ticks2: [10-ns ticks of n factors, bad flag]
  double *dA = nullptr, *dX = nullptr; unsigned long long* dt = nullptr;
  ORBHIP_CHECK_HIP(hipMalloc(&dA, 8192)); ORBHIP_CHECK_HIP(hipMalloc(&dX, 8192)); ORBHIP_CHECK_HIP(hipMalloc(&dt, 16));
  ORBHIP_CHECK_HIP(hipMemcpy(dA, A, 8192, hipMemcpyHostToDevice));
  if (nw == 1) hipLaunchKernelGGL(k_factor_nw<1>, dim3(1), dim3(256), 0, 0, dA, dX, n, dt);
  else if (nw == 2) hipLaunchKernelGGL(k_factor_nw<2>, dim3(1), dim3(256), 0, 0, dA, dX, n, dt);
  else hipLaunchKernelGGL(k_factor_nw<4>, dim3(1), dim3(256), 0, 0, dA, dX, n, dt);
  ORBHIP_CHECK_HIP(hipDeviceSynchronize());
  ORBHIP_CHECK_HIP(hipMemcpy(X, dX, 8192, hipMemcpyDeviceToHost));
  ORBHIP_CHECK_HIP(hipMemcpy(ticks2, dt, 16, hipMemcpyDeviceToHost));
  (void)hipFree(dA); (void)hipFree(dX); (void)hipFree(dt);
  return 0;
}
int ba_debug_p2_prof(unsigned long long* out, int reset) {       // [8][128] absolute ticks (100 MHz)
  ORBHIP_CHECK_HIP(hipDeviceSynchronize());
  if (out) ORBHIP_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_p2_prof), sizeof(unsigned long long) * 8 * 128));
  if (reset) { static unsigned long long z[8 * 128]; ORBHIP_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_p2_prof), z, sizeof(z))); }
  return 0;
}
int ba_debug_chol_prof(unsigned long long* out, int reset) {     // [128][10]: ticks per phase summed over launches, [9] = launches
  ORBHIP_CHECK_HIP(hipDeviceSynchronize());
  if (out) ORBHIP_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chol_prof), sizeof(unsigned long long) * 128 * 10));
  if (reset) { static unsigned long long z[128 * 10]; ORBHIP_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_chol_prof), z, sizeof(z))); }
  return 0;
}
#endif

#if defined(ORBHIP_SCHUR_PROF) && !defined(ORBHIP_CHOL_PROF)
int ba_debug_chol_prof(unsigned long long* out, int reset) {     // k_ba_schur phase stamps of problem 0: [row a][column], [9] = launches
  ORBHIP_CHECK_HIP(hipDeviceSynchronize());
  if (out) ORBHIP_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chol_prof), sizeof(unsigned long long) * 128 * 10));
  if (reset) { static unsigned long long z[128 * 10]; ORBHIP_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_chol_prof), z, sizeof(z))); }
  return 0;
}
#endif

// ---- 7-vector pose codec (src/MatEigenConverter.cc:66-85); T = row-major 4x4 ------------------------------------------
// Matrix4dToMatrix_7_1: [t, Eigen::Quaterniond(R).coeffs()] -- Eigen's matrix -> quaternion conversion branches on the
// trace and, when it is not positive, on the largest diagonal element (no normalisation, no sign convention on w).
int ba_matrix4d_to_pose7(const double* T, double* pose7) {
  ORBHIP_REQUIRE(T && pose7, ORBHIP_EINVAL, "NULL argument");
  const double m[3][3] = {{T[0], T[1], T[2]}, {T[4], T[5], T[6]}, {T[8], T[9], T[10]}};
  double q[4];                                                  // x y z w
  double t = m[0][0] + m[1][1] + m[2][2];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[2][1] - m[1][2]) * t; q[1] = (m[0][2] - m[2][0]) * t; q[2] = (m[1][0] - m[0][1]) * t;
  } else {
    int i = 0;
    if (m[1][1] > m[0][0]) i = 1;
    if (m[2][2] > m[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m[k][j] - m[j][k]) * t; q[j] = (m[j][i] + m[i][j]) * t; q[k] = (m[k][i] + m[i][k]) * t;
  }
  pose7[0] = T[3]; pose7[1] = T[7]; pose7[2] = T[11];
  pose7[3] = q[0]; pose7[4] = q[1]; pose7[5] = q[2]; pose7[6] = q[3];
  return 0;
}
// Matrix_7_1_ToMatrix4d: q.normalized().toRotationMatrix() (coefficients divided by the norm, then Eigen's product form)
int ba_pose7_to_matrix4d(const double* pose7, double* T) {
  ORBHIP_REQUIRE(T && pose7, ORBHIP_EINVAL, "NULL argument");
  const double n = std::sqrt(pose7[3] * pose7[3] + pose7[4] * pose7[4] + pose7[5] * pose7[5] + pose7[6] * pose7[6]);
  const double x = pose7[3] / n, y = pose7[4] / n, z = pose7[5] / n, w = pose7[6] / n;
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x,
               tyy = ty * y, tyz = tz * y, tzz = tz * z;
  T[0] = 1 - (tyy + tzz); T[1] = txy - twz; T[2] = txz + twy; T[3] = pose7[0];
  T[4] = txy + twz; T[5] = 1 - (txx + tzz); T[6] = tyz - twx; T[7] = pose7[1];
  T[8] = txz - twy; T[9] = tyz + twx; T[10] = 1 - (txx + tyy); T[11] = pose7[2];
  T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
  return 0;
}

int ba_solve_batch(const ba_problem* problems, int nproblems, const ba_options* opts, ba_summary* summaries) {
  ORBHIP_REQUIRE(problems && opts && nproblems > 0, ORBHIP_EINVAL, "NULL argument");
  std::vector<BaInputs> in(nproblems);
  for (int q = 0; q < nproblems; q++) {
    const ba_problem& B = problems[q];
    in[q] = BaInputs{B.K4, B.poses7, B.cam_fixed, B.ncam, B.pts3, B.npts, B.obs_cam, B.obs_pt, B.obs_uv, B.obs_weight, B.obs_robust, B.nobs};
  }
  return ba_solve_batch_impl(in.data(), nproblems, opts, summaries);
}

}  // extern "C"
