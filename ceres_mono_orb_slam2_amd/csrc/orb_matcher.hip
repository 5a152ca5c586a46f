// ============================================================================
// orb_matcher.hip -- MI355X (gfx950) 256-bit Hamming matching behind the C ABI of
// include/orbslam_hip.h.  Distance core of ORB_SLAM2::ORBmatcher
// (reference src/ORBmatcher.cc): DescriptorDistance (:1422-1437), the
// best/second-best inner loop of every Search* method (e.g. :192-208), the ratio /
// threshold acceptance (:210-212) and the rotation-histogram filter
// (ComputeThreeMaxima :1386-1418, idiom :217-252).
//
// Bound by integer VALU issue (xor + v_bcnt_u32_b32), not by HBM: targets are staged
// once per workgroup in LDS and read with wave-uniform (broadcast) ds_read_b128; each
// lane keeps its query in 8 VGPRs.  Order-dependent greedy bookkeeping of the guided
// entry points stays on the host in reference loop order.
// ============================================================================
#include <hip/hip_runtime.h>
#include <climits>
#include <cmath>
#include <cstdint>
#include <vector>
#include <mutex>
#include <algorithm>

#include "common.h"
#include "orb_frame.h"

namespace orbhip {

static const int TH_LOW = 50, HISTO_LENGTH = 30;

// v_bcnt_u32_b32 D = popcount(S0) + S1: written out so that the eight popcounts accumulate in ONE chain.  The compiler,
// given __popc() + __popc() + ..., counts every word against 0 and folds the eight results with v_add3_u32 - 11.5
// instructions per distance instead of 8 (the match kernel runs at 85 % of the VALU issue rate of its CU, so
// instructions are time).
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
  uint32_t d;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc));
  return d;
}
__device__ __forceinline__ int hamming256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
  uint32_t d = (uint32_t)__popc(a0.x ^ b0.x);
  d = bcnt_acc(a0.y ^ b0.y, d); d = bcnt_acc(a0.z ^ b0.z, d); d = bcnt_acc(a0.w ^ b0.w, d);
  d = bcnt_acc(a1.x ^ b1.x, d); d = bcnt_acc(a1.y ^ b1.y, d); d = bcnt_acc(a1.z ^ b1.z, d); d = bcnt_acc(a1.w ^ b1.w, d);
  return (int)d;
}

// ---- brute force: every query against every target, first minimum wins -----------------------
#define BF_CHUNK 1024
__global__ __launch_bounds__(256) void k_best2_brute(const uint8_t* __restrict__ q, int nq,
                                                     const uint8_t* __restrict__ t, int nt,
                                                     int* __restrict__ best_idx, int* __restrict__ best_d,
                                                     int* __restrict__ second_d) {
  __shared__ uint4 s_t[BF_CHUNK * 2];
  const int i = blockIdx.x * 256 + threadIdx.x;
  uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
  if (i < nq) { q0 = ((const uint4*)q)[2 * (size_t)i]; q1 = ((const uint4*)q)[2 * (size_t)i + 1]; }
  int b1 = 256, b2 = 256, bi = -1;
  for (int base = 0; base < nt; base += BF_CHUNK) {
    const int m = min(BF_CHUNK, nt - base);
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * m; k += 256) s_t[k] = ((const uint4*)t)[2 * (size_t)base + k];
    __syncthreads();
    for (int j = 0; j < m; j++) {
      int d = hamming256(q0, q1, s_t[2 * j], s_t[2 * j + 1]);
      if (d < b1) { b2 = b1; b1 = d; bi = base + j; }
      else if (d < b2) { b2 = d; }
    }
  }
  if (i < nq) { best_idx[i] = bi; best_d[i] = b1; second_d[i] = b2; }
}

// ---- CSR candidate lists: best/second over each query's candidates in list order ---------------
__global__ __launch_bounds__(256) void k_best2_csr(const uint8_t* __restrict__ q, int nq,
                                                   const uint8_t* __restrict__ t, const uint32_t* __restrict__ off,
                                                   const uint32_t* __restrict__ idx, int* __restrict__ best_idx,
                                                   int* __restrict__ best_d, int* __restrict__ second_d) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nq) return;
  const uint4 q0 = ((const uint4*)q)[2 * (size_t)i], q1 = ((const uint4*)q)[2 * (size_t)i + 1];
  int b1 = 256, b2 = 256, bi = -1;
  for (uint32_t c = off[i]; c < off[i + 1]; c++) {
    const uint32_t j = idx[c];
    int d = hamming256(q0, q1, ((const uint4*)t)[2 * (size_t)j], ((const uint4*)t)[2 * (size_t)j + 1]);
    if (d < b1) { b2 = b1; b1 = d; bi = (int)j; }
    else if (d < b2) { b2 = d; }
  }
  best_idx[i] = bi; best_d[i] = b1; second_d[i] = b2;
}

// ---- CSR candidate lists: every (query, candidate) distance, for host-side greedy passes --------
__global__ __launch_bounds__(256) void k_dist_csr(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t,
                                                  const uint32_t* __restrict__ off, const uint32_t* __restrict__ idx,
                                                  uint32_t total, int* __restrict__ dist) {
  const uint32_t c = blockIdx.x * 256u + threadIdx.x;
  if (c >= total) return;
  int lo = 0, hi = nq;                 // largest i with off[i] <= c
  while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (off[mid] <= c) lo = mid; else hi = mid; }
  const uint32_t j = idx[c];
  dist[c] = hamming256(((const uint4*)q)[2 * (size_t)lo], ((const uint4*)q)[2 * (size_t)lo + 1],
                       ((const uint4*)t)[2 * (size_t)j], ((const uint4*)t)[2 * (size_t)j + 1]);
}

// ---- fused frame-pair matcher: brute force + acceptance + rotation consistency ------------------
// one 1024-thread workgroup per pair; targets (<= MP_MAXT) live in LDS.
#define MP_THREADS 1024
#define MP_Q 2
__global__ __launch_bounds__(MP_THREADS) void k_match_pairs(const orbx_keypoint* __restrict__ kps,
                                                            const uint8_t* __restrict__ desc,
                                                            const int* __restrict__ counts, int cap,
                                                            const int* __restrict__ pair_a, const int* __restrict__ pair_b,
                                                            float ratio, int th, int check_ori,
                                                            int* __restrict__ match12, int* __restrict__ nmatch,
                                                            int max_targets) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint4* s_t = (uint4*)smem;                                  // [max_targets * 2]
  __shared__ int s_hist[HISTO_LENGTH], s_keep[3], s_n;
  const int p = blockIdx.x, tid = threadIdx.x;
  const int fa = pair_a[p], fb = pair_b[p];
  int n1 = counts[fa], n2 = counts[fb];
  n1 = n1 < 0 ? 0 : min(n1, cap);
  n2 = n2 < 0 ? 0 : min(n2, min(cap, max_targets));
  const uint4* Q = (const uint4*)(desc + (size_t)fa * cap * 32);
  const uint4* T = (const uint4*)(desc + (size_t)fb * cap * 32);
  const orbx_keypoint* KA = kps + (size_t)fa * cap;
  const orbx_keypoint* KB = kps + (size_t)fb * cap;
  int* M = match12 + (size_t)p * cap;
  if (tid < HISTO_LENGTH) s_hist[tid] = 0;
  if (tid == 0) s_n = 0;
  for (int k = tid; k < 2 * n2; k += MP_THREADS) s_t[k] = T[k];
  __syncthreads();
  const float factor = 1.0f / HISTO_LENGTH;
  // each thread keeps MP_Q queries in registers and walks the LDS-resident targets ONCE (every
  // broadcast ds_read_b128 of a target is reused MP_Q times)
  for (int i0 = 0; i0 < cap; i0 += MP_THREADS * MP_Q) {
    uint4 q0[MP_Q], q1[MP_Q];
    // best / second best as packed keys (distance << 16 | target index): "first minimum wins" is the lexicographic minimum
    // of (d, j), and the reference's second-best update (d < second, ties of the best included) is min(k2, max(k1, k)) -
    // one shift-or, two minima and one maximum per distance instead of two compares and four selects
    uint32_t k1[MP_Q], k2[MP_Q];
#pragma unroll
    for (int s = 0; s < MP_Q; s++) {
      const int i = i0 + s * MP_THREADS + tid;
      k1[s] = (256u << 16) | 0xFFFFu; k2[s] = (256u << 16) | 0xFFFFu;
      if (i < n1) { q0[s] = Q[2 * i]; q1[s] = Q[2 * i + 1]; } else { q0[s] = make_uint4(0, 0, 0, 0); q1[s] = q0[s]; }
    }
    if (i0 + (tid & ~63) < n1) {                         // whole wave beyond n1: nothing to do
      for (int j = 0; j < n2; j++) {
        const uint4 t0 = s_t[2 * j], t1 = s_t[2 * j + 1];
#pragma unroll
        for (int s = 0; s < MP_Q; s++) {
          const uint32_t k = ((uint32_t)hamming256(q0[s], q1[s], t0, t1) << 16) | (uint32_t)j;
          k2[s] = min(k2[s], max(k1[s], k));
          k1[s] = min(k1[s], k);
        }
      }
    }
    int b1[MP_Q], b2[MP_Q], bi[MP_Q];
#pragma unroll
    for (int s = 0; s < MP_Q; s++) {
      b1[s] = (int)(k1[s] >> 16); b2[s] = (int)(k2[s] >> 16);
      bi[s] = (k1[s] & 0xFFFFu) == 0xFFFFu ? -1 : (int)(k1[s] & 0xFFFFu);       // (no target, or only targets at distance 256: rejected by the threshold either way)
    }
#pragma unroll
    for (int s = 0; s < MP_Q; s++) {
      const int i = i0 + s * MP_THREADS + tid;
      int res = -1;
      if (i < n1 && bi[s] >= 0 && b1[s] <= th && (float)b1[s] < __fmul_rn(ratio, (float)b2[s])) {
        res = bi[s];
        int bin = 0;
        if (check_ori) {
          float rot = __fsub_rn(KA[i].angle, KB[bi[s]].angle);
          if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
          bin = (int)roundf(__fmul_rn(rot, factor));
          if (bin == HISTO_LENGTH) bin = 0;
          atomicAdd(&s_hist[bin], 1);
        }
        res |= bin << 24;                                   // stash the bin (indices < 2^24)
      }
      if (i < cap) M[i] = res;
    }
  }
  __syncthreads();
  if (tid == 0) {
    int i1 = -1, i2 = -1, i3 = -1;
    if (check_ori) {                                          // ComputeThreeMaxima (:1386-1418)
      int max1 = 0, max2 = 0, max3 = 0;
      for (int b = 0; b < HISTO_LENGTH; b++) {
        const int s = s_hist[b];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = b; }
        else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = b; }
        else if (s > max3) { max3 = s; i3 = b; }
      }
      if ((float)max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
      else if ((float)max3 < 0.1f * (float)max1) { i3 = -1; }
    }
    s_keep[0] = i1; s_keep[1] = i2; s_keep[2] = i3;
  }
  __syncthreads();
  int mine = 0;
  for (int i = tid; i < cap; i += MP_THREADS) {
    int r = M[i];
    if (r >= 0) {
      int bin = r >> 24, j = r & 0xFFFFFF;
      if (check_ori && bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) j = -1;
      M[i] = j;
      mine += j >= 0;
    }
  }
  if (mine) atomicAdd(&s_n, mine);
  __syncthreads();
  if (tid == 0) nmatch[p] = s_n;
}

// ---- host mirror of Frame's 64x48 grid (reference src/Frame.cc:158-173, :243-320) -----------------
struct FrameGrid {
  static const int COLS = 64, ROWS = 48;
  float min_x, min_y, winv, hinv;
  std::vector<int> cell[COLS][ROWS];
  const float* kps; int n;
  void build(const float* k, int n_, const float* b) {
    kps = k; n = n_; min_x = b[0]; min_y = b[2];
    winv = static_cast<float>(COLS) / (b[1] - b[0]);
    hinv = static_cast<float>(ROWS) / (b[3] - b[2]);
    for (int i = 0; i < n; i++) {
      int px = (int)std::round((k[4 * i] - min_x) * winv), py = (int)std::round((k[4 * i + 1] - min_y) * hinv);
      if (px < 0 || px >= COLS || py < 0 || py >= ROWS) continue;
      cell[px][py].push_back(i);
    }
  }
  void query(float x, float y, float r, int minLevel, int maxLevel, std::vector<uint32_t>& out) const {
    const int min_cx = std::max(0, (int)std::floor((x - min_x - r) * winv));
    if (min_cx >= COLS) return;
    const int max_cx = std::min(COLS - 1, (int)std::ceil((x - min_x + r) * winv));
    if (max_cx < 0) return;
    const int min_cy = std::max(0, (int)std::floor((y - min_y - r) * hinv));
    if (min_cy >= ROWS) return;
    const int max_cy = std::min(ROWS - 1, (int)std::ceil((y - min_y + r) * hinv));
    if (max_cy < 0) return;
    const bool check = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = min_cx; ix <= max_cx; ix++)
      for (int iy = min_cy; iy <= max_cy; iy++)
        for (int j : cell[ix][iy]) {
          int oct = (int)kps[4 * j + 2];
          if (check) {
            if (oct < minLevel) continue;
            if (maxLevel >= 0 && oct > maxLevel) continue;
          }
          if (std::fabs(kps[4 * j] - x) < r && std::fabs(kps[4 * j + 1] - y) < r) out.push_back((uint32_t)j);
        }
  }
};

}  // namespace orbhip

using namespace orbhip;

extern "C" {

int orbm_descriptor_distance(const uint8_t* a, const uint8_t* b) {
  int d = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t x, y;
    std::memcpy(&x, a + 4 * i, 4); std::memcpy(&y, b + 4 * i, 4);
    d += __builtin_popcount(x ^ y);
  }
  return d;
}

int orbm_hamming_best2_device(const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, const uint32_t* d_off,
                              const uint32_t* d_idx, int32_t* d_best_idx, int32_t* d_best_d, int32_t* d_second_d,
                              void* stream) {
  ORBHIP_REQUIRE(nq >= 0 && nt >= 0, ORBHIP_EINVAL, "negative size");
  if (nq == 0) return 0;
  ORBHIP_REQUIRE(d_q && d_best_idx && d_best_d && d_second_d, ORBHIP_EINVAL, "NULL argument");
  ORBHIP_REQUIRE((d_off == nullptr) == (d_idx == nullptr), ORBHIP_EINVAL, "cand_offsets/cand_idx must both be given or both be NULL");
  ORBHIP_REQUIRE(d_off || nt == 0 || d_t, ORBHIP_EINVAL, "NULL targets");
  hipStream_t st = (hipStream_t)stream;
  if (d_off)
    hipLaunchKernelGGL(k_best2_csr, dim3((nq + 255) / 256), dim3(256), 0, st, d_q, nq, d_t, d_off, d_idx, d_best_idx, d_best_d, d_second_d);
  else
    hipLaunchKernelGGL(k_best2_brute, dim3((nq + 255) / 256), dim3(256), 0, st, d_q, nq, d_t, nt, d_best_idx, d_best_d, d_second_d);
  ORBHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

int orbm_hamming_best2(const uint8_t* q, int nq, const uint8_t* t, int nt, const uint32_t* off, const uint32_t* idx,
                       int32_t* best_idx, int32_t* best_d, int32_t* second_d) {
  ORBHIP_REQUIRE(nq >= 0 && nt >= 0, ORBHIP_EINVAL, "negative size");
  if (nq == 0) return 0;
  ORBHIP_REQUIRE(q && best_idx && best_d && second_d && (nt == 0 || t), ORBHIP_EINVAL, "NULL argument");
  ORBHIP_REQUIRE((off == nullptr) == (idx == nullptr), ORBHIP_EINVAL, "cand_offsets/cand_idx must both be given or both be NULL");
  if (int rcd = use_default_device()) return rcd;
  DevBuf dq, dt, doff, didx, dout;
  const size_t total = off ? off[nq] : 0;
  int rc = 0;
  if ((rc = dq.ensure((size_t)nq * 32)) || (rc = dt.ensure(std::max<size_t>((size_t)nt * 32, 32))) ||
      (rc = dout.ensure((size_t)nq * 12)) || (off && ((rc = doff.ensure((size_t)(nq + 1) * 4)) || (rc = didx.ensure(std::max<size_t>(total * 4, 4)))))) {
    dq.release(); dt.release(); doff.release(); didx.release(); dout.release();
    return rc;
  }
  auto cleanup = [&]() { dq.release(); dt.release(); doff.release(); didx.release(); dout.release(); };
#define MCHK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { set_error("%s: %s", #e, hipGetErrorString(_e)); cleanup(); return ORBHIP_ENODEV; } } while (0)
  MCHK(hipMemcpy(dq.p, q, (size_t)nq * 32, hipMemcpyHostToDevice));
  if (nt) MCHK(hipMemcpy(dt.p, t, (size_t)nt * 32, hipMemcpyHostToDevice));
  if (off) {
    MCHK(hipMemcpy(doff.p, off, (size_t)(nq + 1) * 4, hipMemcpyHostToDevice));
    if (total) MCHK(hipMemcpy(didx.p, idx, total * 4, hipMemcpyHostToDevice));
  }
  int32_t* o = dout.as<int32_t>();
  rc = orbm_hamming_best2_device(dq.as<uint8_t>(), nq, dt.as<uint8_t>(), nt, off ? doff.as<uint32_t>() : nullptr,
                                 off ? didx.as<uint32_t>() : nullptr, o, o + nq, o + 2 * (size_t)nq, nullptr);
  if (rc) { cleanup(); return rc; }
  MCHK(hipMemcpy(best_idx, o, (size_t)nq * 4, hipMemcpyDeviceToHost));
  MCHK(hipMemcpy(best_d, o + nq, (size_t)nq * 4, hipMemcpyDeviceToHost));
  MCHK(hipMemcpy(second_d, o + 2 * (size_t)nq, (size_t)nq * 4, hipMemcpyDeviceToHost));
  cleanup();
  return 0;
}

int orbm_match_frames_batch_device(const orbx_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_counts, int cap,
                                   const int32_t* d_pair_a, const int32_t* d_pair_b, int npairs, float ratio, int th,
                                   int check_ori, int32_t* d_match12, int32_t* d_nmatch, void* stream) {
  ORBHIP_REQUIRE(npairs >= 0 && cap > 0, ORBHIP_EINVAL, "bad size");
  if (npairs == 0) return 0;
  ORBHIP_REQUIRE(d_kps && d_desc && d_counts && d_pair_a && d_pair_b && d_match12 && d_nmatch, ORBHIP_EINVAL, "NULL argument");
  const size_t lds = (size_t)cap * 32;
  ORBHIP_REQUIRE(lds <= 150 * 1024, ORBHIP_EINVAL, "per-frame capacity too large for the LDS-resident matcher (cap <= 4800)");
  ORBHIP_REQUIRE(cap < (1 << 24), ORBHIP_EINVAL, "cap too large");
  if (lds > 64 * 1024) {                                   // the dynamic-LDS opt-in is per device: cache it per device, under a lock
    static std::mutex mu; static size_t attr_set[64] = {0};
    int dev = 0;
    ORBHIP_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(mu);
    if (dev < 0 || dev >= 64 || lds > attr_set[dev]) {
      ORBHIP_CHECK_HIP(hipFuncSetAttribute((const void*)k_match_pairs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      if (dev >= 0 && dev < 64) attr_set[dev] = lds;
    }
  }
  hipLaunchKernelGGL(k_match_pairs, dim3(npairs), dim3(MP_THREADS), lds, (hipStream_t)stream, d_kps, d_desc, d_counts,
                     cap, d_pair_a, d_pair_b, ratio, th, check_ori, d_match12, d_nmatch, cap);
  ORBHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- shared helper: all (query, candidate) distances of a CSR list on the GPU (host pointers in/out) -----------
static int csr_distances_gpu(const uint8_t* q, int nq, const uint8_t* t, int nt, const std::vector<uint32_t>& off,
                             const std::vector<uint32_t>& idx, std::vector<int>& dist) {
  const uint32_t total = off[nq];
  dist.resize(total);
  if (total == 0) return 0;
  if (int rcd = use_default_device()) return rcd;
  DevBuf dq, dt, doff, didx, dd;
  auto cleanup = [&]() { dq.release(); dt.release(); doff.release(); didx.release(); dd.release(); };
  int rc = 0;
  if ((rc = dq.ensure((size_t)nq * 32)) || (rc = dt.ensure((size_t)nt * 32)) || (rc = doff.ensure((size_t)(nq + 1) * 4)) ||
      (rc = didx.ensure((size_t)total * 4)) || (rc = dd.ensure((size_t)total * 4))) { cleanup(); return rc; }
  MCHK(hipMemcpy(dq.p, q, (size_t)nq * 32, hipMemcpyHostToDevice));
  MCHK(hipMemcpy(dt.p, t, (size_t)nt * 32, hipMemcpyHostToDevice));
  MCHK(hipMemcpy(doff.p, off.data(), (size_t)(nq + 1) * 4, hipMemcpyHostToDevice));
  MCHK(hipMemcpy(didx.p, idx.data(), (size_t)total * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_dist_csr, dim3((total + 255) / 256), dim3(256), 0, 0, dq.as<uint8_t>(), nq, dt.as<uint8_t>(),
                     doff.as<uint32_t>(), didx.as<uint32_t>(), total, dd.as<int>());
  MCHK(hipGetLastError());
  MCHK(hipMemcpy(dist.data(), dd.p, (size_t)total * 4, hipMemcpyDeviceToHost));
  cleanup();
  return 0;
}

static void three_maxima_host(const int* cnt, int& i1, int& i2, int& i3) {          // src/ORBmatcher.cc:1386-1418
  int max1 = 0, max2 = 0, max3 = 0; i1 = i2 = i3 = -1;
  for (int b = 0; b < HISTO_LENGTH; b++) {
    const int s = cnt[b];
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = b; }
    else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = b; }
    else if (s > max3) { max3 = s; i3 = b; }
  }
  if (max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
  else if (max3 < 0.1f * (float)max1) { i3 = -1; }
}
static inline int rot_bin_host(float a1, float a2) {
  const float factor = 1.0f / HISTO_LENGTH;
  float rot = a1 - a2;
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)std::round(rot * factor);
  if (bin == HISTO_LENGTH) bin = 0;
  return bin;
}

int orbm_search_by_projection(const float* kps4, const uint8_t* desc, int n, const float* bounds, const float* q_uv,
                              const float* q_radius, const int32_t* q_min_level, const int32_t* q_max_level,
                              const int32_t* q_pred_level, const uint8_t* q_desc, const uint8_t* q_valid,
                              const float* q_angle, int nq, const float* inv_level_sigma2, float chi2_gate,
                              uint8_t* taken, int mode_best2, float ratio, int th, int check_ori, int32_t* q_match,
                              int32_t* q_best_dist, int* nmatches) {
  ORBHIP_REQUIRE(n >= 0 && nq >= 0 && nmatches && q_match, ORBHIP_EINVAL, "bad size");
  *nmatches = 0;
  for (int i = 0; i < nq; i++) { q_match[i] = -1; if (q_best_dist) q_best_dist[i] = 256; }
  if (nq == 0 || n == 0) return 0;
  ORBHIP_REQUIRE(kps4 && desc && bounds && q_uv && q_radius && q_desc, ORBHIP_EINVAL, "NULL argument");
  ORBHIP_REQUIRE(!check_ori || q_angle, ORBHIP_EINVAL, "rotation check needs q_angle");
  ORBHIP_REQUIRE(chi2_gate <= 0.f || inv_level_sigma2, ORBHIP_EINVAL, "chi2 gate needs inv_level_sigma2");
  // ---- grid, window candidates and every (query, candidate) distance on the device: the host only keeps the
  //      order-dependent greedy pass below, fed with the CSR lists in the reference's candidate order
  std::vector<uint32_t> off(nq + 1, 0), idx;
  std::vector<int> dist;
  {
    if (int rcd = use_default_device()) return rcd;
    FrameGridDev G; DevBuf dk, dd, dq, dr, dmn, dmx, dv, dqd, dcnt, doff, didx, ddist;
    DevBuf* all[] = {&dk, &dd, &dq, &dr, &dmn, &dmx, &dv, &dqd, &dcnt, &doff, &didx, &ddist};
    auto cleanup = [&]() { G.release(); for (DevBuf* b : all) b->release(); };
    int rc = 0;
    if ((rc = dk.ensure((size_t)n * 16)) || (rc = dd.ensure((size_t)n * 32)) || (rc = dq.ensure((size_t)nq * 8)) || (rc = dr.ensure((size_t)nq * 4)) ||
        (rc = dqd.ensure((size_t)nq * 32)) || (q_min_level && (rc = dmn.ensure((size_t)nq * 4))) || (q_max_level && (rc = dmx.ensure((size_t)nq * 4))) ||
        (q_valid && (rc = dv.ensure((size_t)nq)))) { cleanup(); return rc; }
#define SBP_CHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { set_error("%s failed: %s", #expr, hipGetErrorString(_e)); cleanup(); return ORBHIP_ENODEV; } } while (0)
    SBP_CHK(hipMemcpy(dk.p, kps4, (size_t)n * 16, hipMemcpyHostToDevice)); SBP_CHK(hipMemcpy(dd.p, desc, (size_t)n * 32, hipMemcpyHostToDevice));
    SBP_CHK(hipMemcpy(dq.p, q_uv, (size_t)nq * 8, hipMemcpyHostToDevice)); SBP_CHK(hipMemcpy(dr.p, q_radius, (size_t)nq * 4, hipMemcpyHostToDevice));
    SBP_CHK(hipMemcpy(dqd.p, q_desc, (size_t)nq * 32, hipMemcpyHostToDevice));
    if (q_min_level) SBP_CHK(hipMemcpy(dmn.p, q_min_level, (size_t)nq * 4, hipMemcpyHostToDevice));
    if (q_max_level) SBP_CHK(hipMemcpy(dmx.p, q_max_level, (size_t)nq * 4, hipMemcpyHostToDevice));
    if (q_valid) SBP_CHK(hipMemcpy(dv.p, q_valid, (size_t)nq, hipMemcpyHostToDevice));
    const float gb[4] = {bounds[0], bounds[1], bounds[2], bounds[3]};
    if ((rc = frame_grid_build(G, dk.as<float>(), n, gb, nullptr))) { cleanup(); return rc; }
    uint32_t total = 0;
    if ((rc = frame_area_candidates(G, dk.as<float>(), dq.as<float>(), dr.as<float>(), q_min_level ? dmn.as<int>() : nullptr,
                                    q_max_level ? dmx.as<int>() : nullptr, q_valid ? dv.as<uint8_t>() : nullptr, nq, dcnt, doff, didx, &total, nullptr))) { cleanup(); return rc; }
    SBP_CHK(hipMemcpy(off.data(), doff.p, (size_t)(nq + 1) * 4, hipMemcpyDeviceToHost));
    idx.resize(total); dist.resize(total);
    if (total) {
      if ((rc = ddist.ensure((size_t)total * 4))) { cleanup(); return rc; }
      hipLaunchKernelGGL(k_dist_csr, dim3((total + 255) / 256), dim3(256), 0, 0, dqd.as<uint8_t>(), nq, dd.as<uint8_t>(), doff.as<uint32_t>(),
                         didx.as<uint32_t>(), total, ddist.as<int>());
      SBP_CHK(hipGetLastError());
      SBP_CHK(hipMemcpy(idx.data(), didx.p, (size_t)total * 4, hipMemcpyDeviceToHost));
      SBP_CHK(hipMemcpy(dist.data(), ddist.p, (size_t)total * 4, hipMemcpyDeviceToHost));
    }
#undef SBP_CHK
    cleanup();
  }
  // ---- greedy pass in query order (reference loop order) ---------------------------------------
  int nm = 0;
  std::vector<int> hist_bin(nq, -1);
  int cnt[HISTO_LENGTH] = {0};
  for (int i = 0; i < nq; i++) {
    int bestDist = 256, bestDist2 = 256, bestLevel = -1, bestLevel2 = -1, bestIdx = -1;
    for (uint32_t c = off[i]; c < off[i + 1]; c++) {
      const int t = (int)idx[c];
      if (taken && taken[t]) continue;
      const int lvl = (int)kps4[4 * t + 2];
      if (q_pred_level && q_pred_level[i] >= 0 && (lvl < q_pred_level[i] - 1 || lvl > q_pred_level[i])) continue;
      if (chi2_gate > 0.f) {
        const float ex = q_uv[2 * i] - kps4[4 * t], ey = q_uv[2 * i + 1] - kps4[4 * t + 1];
        const float e2 = ex * ex + ey * ey;
        if (e2 * inv_level_sigma2[lvl] > chi2_gate) continue;
      }
      const int d = dist[c];
      if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestLevel2 = bestLevel; bestLevel = lvl; bestIdx = t; }
      else if (mode_best2 && d < bestDist2) { bestLevel2 = lvl; bestDist2 = d; }
    }
    if (q_best_dist) q_best_dist[i] = bestDist;
    if (bestIdx >= 0 && bestDist <= th) {
      if (mode_best2 && bestLevel == bestLevel2 && bestDist > ratio * bestDist2) continue;
      q_match[i] = bestIdx;
      if (taken) taken[bestIdx] = 1;
      nm++;
      if (check_ori) { hist_bin[i] = rot_bin_host(q_angle[i], kps4[4 * bestIdx + 3]); cnt[hist_bin[i]]++; }
    }
  }
  if (check_ori) {
    int i1, i2, i3;
    three_maxima_host(cnt, i1, i2, i3);
    for (int i = 0; i < nq; i++)
      if (q_match[i] >= 0 && hist_bin[i] != i1 && hist_bin[i] != i2 && hist_bin[i] != i3) {
        if (taken) taken[q_match[i]] = 0;
        q_match[i] = -1; nm--;
      }
  }
  *nmatches = nm;
  return 0;
}

int orbm_search_by_bow(const uint8_t* desc1, int n1, const uint8_t* valid1, const float* angle1, const uint8_t* desc2, int n2,
                       const uint8_t* valid2, const float* angle2, const uint32_t* fv1_node, const uint32_t* fv1_off,
                       const uint32_t* fv1_idx, int fv1_n, const uint32_t* fv2_node, const uint32_t* fv2_off,
                       const uint32_t* fv2_idx, int fv2_n, float ratio, int th, int strict, int check_ori,
                       int32_t* match12, int* nmatches) {
  ORBHIP_REQUIRE(n1 >= 0 && n2 >= 0 && fv1_n >= 0 && fv2_n >= 0 && nmatches && (n1 == 0 || match12), ORBHIP_EINVAL, "bad size");
  *nmatches = 0;
  for (int i = 0; i < n1; i++) match12[i] = -1;
  if (n1 == 0 || n2 == 0 || fv1_n == 0 || fv2_n == 0) return 0;
  ORBHIP_REQUIRE(desc1 && desc2 && fv1_node && fv1_off && fv1_idx && fv2_node && fv2_off && fv2_idx, ORBHIP_EINVAL, "NULL argument");
  ORBHIP_REQUIRE(!check_ori || (angle1 && angle2), ORBHIP_EINVAL, "rotation check needs angles");
  // merge-walk of the two feature vectors (std::map order = ascending node id): queries in (node, list) order
  std::vector<int> qidx;                       // query -> idx1
  std::vector<uint32_t> off(1, 0), idx;
  int a = 0, b = 0;
  while (a < fv1_n && b < fv2_n) {
    if (fv1_node[a] == fv2_node[b]) {
      for (uint32_t e1 = fv1_off[a]; e1 < fv1_off[a + 1]; e1++) {
        const int i1 = (int)fv1_idx[e1];
        if (valid1 && !valid1[i1]) continue;
        qidx.push_back(i1);
        for (uint32_t e2 = fv2_off[b]; e2 < fv2_off[b + 1]; e2++) idx.push_back(fv2_idx[e2]);
        off.push_back((uint32_t)idx.size());
      }
      a++; b++;
    } else if (fv1_node[a] < fv2_node[b]) a++;
    else b++;
  }
  const int nq = (int)qidx.size();
  if (nq == 0) return 0;
  std::vector<uint8_t> qd((size_t)nq * 32);
  for (int i = 0; i < nq; i++) std::memcpy(&qd[(size_t)32 * i], desc1 + (size_t)32 * qidx[i], 32);
  std::vector<int> dist;
  if (int rc = csr_distances_gpu(qd.data(), nq, desc2, n2, off, idx, dist)) return rc;
  std::vector<uint8_t> matched2(n2, 0);
  std::vector<int> bins(n1, -1);
  int cnt[HISTO_LENGTH] = {0}, nm = 0;
  for (int i = 0; i < nq; i++) {
    const int i1 = qidx[i];
    int bestDist1 = 256, bestDist2 = 256, bestIdx2 = -1;
    for (uint32_t c = off[i]; c < off[i + 1]; c++) {
      const int i2 = (int)idx[c];
      if (matched2[i2] || (valid2 && !valid2[i2])) continue;
      const int d = dist[c];
      if (d < bestDist1) { bestDist2 = bestDist1; bestDist1 = d; bestIdx2 = i2; }
      else if (d < bestDist2) { bestDist2 = d; }
    }
    const bool under = strict ? (bestDist1 < th) : (bestDist1 <= th);
    if (under && static_cast<float>(bestDist1) < ratio * static_cast<float>(bestDist2)) {
      match12[i1] = bestIdx2; matched2[bestIdx2] = 1; nm++;
      if (check_ori) { bins[i1] = rot_bin_host(angle1[i1], angle2[bestIdx2]); cnt[bins[i1]]++; }
    }
  }
  if (check_ori) {
    int i1, i2, i3;
    three_maxima_host(cnt, i1, i2, i3);
    for (int i = 0; i < n1; i++)
      if (match12[i] >= 0 && bins[i] != i1 && bins[i] != i2 && bins[i] != i3) { match12[i] = -1; nm--; }
  }
  *nmatches = nm;
  return 0;
}

int orbm_search_for_triangulation(const float* kps1, const uint8_t* desc1, const uint8_t* unmapped1, int n1, const float* kps2,
                                   const uint8_t* desc2, const uint8_t* unmapped2, int n2, const uint32_t* fv1_node,
                                   const uint32_t* fv1_off, const uint32_t* fv1_idx, int fv1_n, const uint32_t* fv2_node,
                                   const uint32_t* fv2_off, const uint32_t* fv2_idx, int fv2_n, const double* F12, float ex, float ey,
                                   const float* scale_factors, const float* level_sigma2, int check_ori, int32_t* match12,
                                   int* nmatches) {
  ORBHIP_REQUIRE(n1 >= 0 && n2 >= 0 && fv1_n >= 0 && fv2_n >= 0 && nmatches && (n1 == 0 || match12), ORBHIP_EINVAL, "bad size");
  *nmatches = 0;
  for (int i = 0; i < n1; i++) match12[i] = -1;
  if (n1 == 0 || n2 == 0 || fv1_n == 0 || fv2_n == 0) return 0;
  ORBHIP_REQUIRE(kps1 && kps2 && desc1 && desc2 && F12 && scale_factors && level_sigma2 && fv1_node && fv2_node, ORBHIP_EINVAL, "NULL argument");
  std::vector<int> qidx;
  std::vector<uint32_t> off(1, 0), idx;
  int a = 0, b = 0;
  while (a < fv1_n && b < fv2_n) {
    if (fv1_node[a] == fv2_node[b]) {
      for (uint32_t e1 = fv1_off[a]; e1 < fv1_off[a + 1]; e1++) {
        const int i1 = (int)fv1_idx[e1];
        if (unmapped1 && !unmapped1[i1]) continue;                  // already holds a MapPoint (:620-623)
        qidx.push_back(i1);
        for (uint32_t e2 = fv2_off[b]; e2 < fv2_off[b + 1]; e2++) idx.push_back(fv2_idx[e2]);
        off.push_back((uint32_t)idx.size());
      }
      a++; b++;
    } else if (fv1_node[a] < fv2_node[b]) a++;
    else b++;
  }
  const int nq = (int)qidx.size();
  if (nq == 0) return 0;
  std::vector<uint8_t> qd((size_t)nq * 32);
  for (int i = 0; i < nq; i++) std::memcpy(&qd[(size_t)32 * i], desc1 + (size_t)32 * qidx[i], 32);
  std::vector<int> dist;
  if (int rc = csr_distances_gpu(qd.data(), nq, desc2, n2, off, idx, dist)) return rc;
  std::vector<int> bins(n1, -1);
  int cnt[HISTO_LENGTH] = {0}, nm = 0;
  for (int i = 0; i < nq; i++) {
    const int i1 = qidx[i];
    const float x1 = kps1[4 * i1], y1 = kps1[4 * i1 + 1];
    // epipolar line in image 2: l = x1' F12 (CheckDistEpipolarLine, :128-149)
    const float la = x1 * F12[0] + y1 * F12[3] + F12[6];
    const float lb = x1 * F12[1] + y1 * F12[4] + F12[7];
    const float lc = x1 * F12[2] + y1 * F12[5] + F12[8];
    int bestDist = TH_LOW, bestIdx2 = -1;
    for (uint32_t c = off[i]; c < off[i + 1]; c++) {
      const int i2 = (int)idx[c];
      if (unmapped2 && !unmapped2[i2]) continue;                     // (vbMatched2 is never set in this fork)
      const int d = dist[c];
      if (d > TH_LOW || d > bestDist) continue;
      const float x2 = kps2[4 * i2], y2 = kps2[4 * i2 + 1];
      const int oct2 = (int)kps2[4 * i2 + 2];
      const float distex = ex - x2, distey = ey - y2;
      if (distex * distex + distey * distey < 100 * scale_factors[oct2]) continue;
      const float num = la * x2 + lb * y2 + lc;
      const float den = la * la + lb * lb;
      if (den == 0) continue;
      const float dsqr = num * num / den;
      if (dsqr < 3.84 * level_sigma2[oct2]) { bestIdx2 = i2; bestDist = d; }
    }
    if (bestIdx2 >= 0) {
      match12[i1] = bestIdx2; nm++;
      if (check_ori) { bins[i1] = rot_bin_host(kps1[4 * i1 + 3], kps2[4 * bestIdx2 + 3]); cnt[bins[i1]]++; }
    }
  }
  if (check_ori) {
    int i1, i2, i3;
    three_maxima_host(cnt, i1, i2, i3);
    for (int i = 0; i < n1; i++)
      if (match12[i] >= 0 && bins[i] != i1 && bins[i] != i2 && bins[i] != i3) { match12[i] = -1; nm--; }
  }
  *nmatches = nm;
  return 0;
}

int orbm_search_for_initialization(const float* kps1, const uint8_t* desc1, int n1, const float* kps2,
                                   const uint8_t* desc2, int n2, const float* bounds2, float* prev_matched, int window,
                                   float nnratio, int check_ori, int32_t* matches12, int* nmatches) {
  ORBHIP_REQUIRE(n1 >= 0 && n2 >= 0 && nmatches, ORBHIP_EINVAL, "bad size");
  *nmatches = 0;
  if (n1 == 0) return 0;
  ORBHIP_REQUIRE(kps1 && desc1 && prev_matched && matches12 && bounds2 && (n2 == 0 || (kps2 && desc2)), ORBHIP_EINVAL, "NULL argument");
  // ---- candidate generation on the host: Frame::GetFeaturesInArea (src/Frame.cc:243-307) ------
  FrameGrid* G = new FrameGrid();
  G->build(kps2, n2, bounds2);
  std::vector<uint32_t> off(n1 + 1, 0), idx;
  for (int i1 = 0; i1 < n1; i1++) {
    off[i1] = (uint32_t)idx.size();
    int level1 = (int)kps1[4 * i1 + 2];
    if (level1 > 0) continue;                                    // (:383-385)
    G->query(prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)window, level1, level1, idx);
  }
  off[n1] = (uint32_t)idx.size();
  delete G;
  const uint32_t total = off[n1];
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  if (total == 0) return 0;
  // ---- all candidate distances on the GPU -----------------------------------------------------
  std::vector<int> dist(total);
  {
    if (int rcd = use_default_device()) return rcd;
    DevBuf dq, dt, doff, didx, dd;
    auto cleanup = [&]() { dq.release(); dt.release(); doff.release(); didx.release(); dd.release(); };
    int rc = 0;
    if ((rc = dq.ensure((size_t)n1 * 32)) || (rc = dt.ensure((size_t)n2 * 32)) || (rc = doff.ensure((size_t)(n1 + 1) * 4)) ||
        (rc = didx.ensure((size_t)total * 4)) || (rc = dd.ensure((size_t)total * 4))) { cleanup(); return rc; }
    MCHK(hipMemcpy(dq.p, desc1, (size_t)n1 * 32, hipMemcpyHostToDevice));
    MCHK(hipMemcpy(dt.p, desc2, (size_t)n2 * 32, hipMemcpyHostToDevice));
    MCHK(hipMemcpy(doff.p, off.data(), (size_t)(n1 + 1) * 4, hipMemcpyHostToDevice));
    MCHK(hipMemcpy(didx.p, idx.data(), (size_t)total * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_dist_csr, dim3((total + 255) / 256), dim3(256), 0, 0, dq.as<uint8_t>(), n1, dt.as<uint8_t>(),
                       doff.as<uint32_t>(), didx.as<uint32_t>(), total, dd.as<int>());
    MCHK(hipGetLastError());
    MCHK(hipMemcpy(dist.data(), dd.p, (size_t)total * 4, hipMemcpyDeviceToHost));
    cleanup();
  }
  // ---- order-dependent greedy pass in reference loop order (src/ORBmatcher.cc:379-441) -----------
  int nm = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  std::vector<int> matchedDist(n2, INT_MAX), matches21(n2, -1);
  const float factor = 1.0f / HISTO_LENGTH;
  for (int i1 = 0; i1 < n1; i1++) {
    if (off[i1] == off[i1 + 1]) continue;
    int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
    for (uint32_t c = off[i1]; c < off[i1 + 1]; c++) {
      const int i2 = (int)idx[c], d = dist[c];
      if (matchedDist[i2] <= d) continue;                        // (:408)
      if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestIdx2 = i2; }
      else if (d < bestDist2) { bestDist2 = d; }
    }
    if (bestDist <= TH_LOW) {
      if (bestDist < (float)bestDist2 * nnratio) {
        if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; nm--; }   // (:421-424)
        matches12[i1] = bestIdx2; matches21[bestIdx2] = i1; matchedDist[bestIdx2] = bestDist;
        nm++;
        if (check_ori) {
          float rot = kps1[4 * i1 + 3] - kps2[4 * bestIdx2 + 3];
          if (rot < 0.0) rot += 360.0f;
          int bin = (int)std::round(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rotHist[bin].push_back(i1);
        }
      }
    }
  }
  if (check_ori) {
    int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
    for (int b = 0; b < HISTO_LENGTH; b++) {
      const int s = (int)rotHist[b].size();
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = b; }
      else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = b; }
      else if (s > max3) { max3 = s; i3 = b; }
    }
    if (max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
    else if (max3 < 0.1f * (float)max1) { i3 = -1; }
    for (int b = 0; b < HISTO_LENGTH; b++) {
      if (b == i1 || b == i2 || b == i3) continue;
      for (int idx1 : rotHist[b]) if (matches12[idx1] >= 0) { matches12[idx1] = -1; nm--; }
    }
  }
  for (int i1 = 0; i1 < n1; i1++)
    if (matches12[i1] >= 0) { prev_matched[2 * i1] = kps2[4 * matches12[i1]]; prev_matched[2 * i1 + 1] = kps2[4 * matches12[i1] + 1]; }
  *nmatches = nm;
  return 0;
}

}  // extern "C"
