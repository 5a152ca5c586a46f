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
#include <map>
#include <mutex>
#include <algorithm>
#include <cstring>
#include <cstdlib>

#include "common.h"
#include "orb_frame.h"
#include "handoff.h"

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
// median of three unsigned keys in ONE instruction.  With k1 <= k2 (best and second-best key) the second-best update of the
// reference, min(k2, max(k1, k)), is exactly med3(k1, k2, k): k <= k1 -> k1, k1 < k < k2 -> k, k >= k2 -> k2.
__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

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
// `nsplit` 1024-thread workgroups per pair; targets (<= max_targets) live in LDS of every one of them, the queries are dealt
// out in 64-wide chunks (chunk g belongs to workgroup g % nsplit), so that two workgroups of one pair share a CU (2 x 76 kB of
// LDS, 8 waves per SIMD) and few pairs still fill the chip.  With nsplit > 1 the rotation histogram is merged through global
// memory: every workgroup adds its 30 bins to the pair's scratch row, and the LAST one to arrive (atomic ticket after a
// device-scope fence) runs ComputeThreeMaxima and the filter pass over the whole pair, then clears the scratch for the next call.
#define MP_THREADS 1024
template <int MP_Q>       // queries per thread: 2 when a pair is one workgroup (16 waves x 2 chunks = 2048 query slots), 1 when it is split
__global__ __launch_bounds__(MP_THREADS) void k_match_pairs(const orbx_keypoint* __restrict__ kps,
                                                            const uint8_t* __restrict__ desc,
                                                            const int* __restrict__ counts, int cap,
                                                            const int* __restrict__ pair_a, const int* __restrict__ pair_b,
                                                            float ratio, int th, int check_ori,
                                                            int* __restrict__ match12, int* __restrict__ nmatch,
                                                            int max_targets, int nsplit, int* __restrict__ scratch /*[npairs][32]*/) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint4* s_t = (uint4*)smem;                                  // [max_targets * 2]
  __shared__ int s_hist[HISTO_LENGTH], s_keep[3], s_n, s_last;
  const int p = (int)blockIdx.x / nsplit, sp = (int)blockIdx.x % nsplit, tid = threadIdx.x;
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
  const int lane = tid & 63, wv = tid >> 6;
  const int nchunk = (cap + 63) >> 6;                         // 64-wide query chunks of the pair
  // each thread keeps MP_Q queries in registers and walks the LDS-resident targets ONCE (every
  // broadcast ds_read_b128 of a target is reused MP_Q times)
  for (int c0 = 0; (c0 * nsplit + sp) < nchunk; c0 += (MP_THREADS / 64) * MP_Q) {
    uint4 q0[MP_Q], q1[MP_Q];
    int qi[MP_Q];
    // best / second best as packed keys (distance << 16 | target index): "first minimum wins" is the lexicographic minimum
    // of (d, j), and the reference's second-best update (d < second, ties of the best included) is min(k2, max(k1, k)) -
    // one shift-or, two minima and one maximum per distance instead of two compares and four selects
    uint32_t k1[MP_Q], k2[MP_Q];
    bool any_q = false;
#pragma unroll
    for (int s = 0; s < MP_Q; s++) {
      const int g = (c0 + wv * MP_Q + s) * nsplit + sp;                    // this wave's chunk (wave-uniform)
      const int i = g * 64 + lane;
      qi[s] = i;
      k1[s] = (256u << 16) | 0xFFFFu; k2[s] = (256u << 16) | 0xFFFFu;
      if (i < n1) { q0[s] = Q[2 * i]; q1[s] = Q[2 * i + 1]; } else { q0[s] = make_uint4(0, 0, 0, 0); q1[s] = q0[s]; }
      any_q = any_q || (g * 64 < n1);
    }
    if (any_q) {                                          // (whole wave beyond n1: nothing to do)
      // Software pipeline over the LDS-resident targets, two per trip on alternating register sets: the NEXT target's two
      // broadcast reads are requested before the current one's 2 x 19 VALU instructions (the plain loop waited for its own reads
      // at the top of every iteration: one exposed LDS round trip per target and wave).  sched_barrier keeps the reads where
      // they are written (the scheduler sinks them to the middle of the body otherwise), the empty asm pins the values to this
      // trip (the compiler otherwise re-issues the reads at the top of the next one, in front of their first use).
      auto score = [&](const uint4& t0, const uint4& t1, int j) {
#pragma unroll
        for (int s = 0; s < MP_Q; s++) {
          const uint32_t k = ((uint32_t)hamming256(q0[s], q1[s], t0, t1) << 16) | (uint32_t)j;
          k2[s] = med3_u32(k1[s], k2[s], k);              // = min(k2, max(k1, k)) because k1 <= k2: the median of the three
          k1[s] = min(k1[s], k);
        }
      };
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));            // (one 128-bit register tuple per asm operand)
      const u32x4* s_v = (const u32x4*)s_t;
      auto U4 = [](const u32x4& v) { return make_uint4(v.x, v.y, v.z, v.w); };
      u32x4 a0 = s_v[0], a1 = s_v[1];
      for (int j = 0; j < n2; j += 2) {
        const int jb = min(j + 1, n2 - 1), ja = min(j + 2, n2 - 1);
        u32x4 b0 = s_v[2 * jb], b1 = s_v[2 * jb + 1];
        __builtin_amdgcn_sched_barrier(0);
        score(U4(a0), U4(a1), j);
        asm volatile("" : "+v"(b0), "+v"(b1));
        a0 = s_v[2 * ja]; a1 = s_v[2 * ja + 1];
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < n2) score(U4(b0), U4(b1), j + 1);
        asm volatile("" : "+v"(a0), "+v"(a1));
      }
    }
#pragma unroll
    for (int s = 0; s < MP_Q; s++) {
      const int i = qi[s];
      const int b1 = (int)(k1[s] >> 16), b2 = (int)(k2[s] >> 16);
      const int bi = (k1[s] & 0xFFFFu) == 0xFFFFu ? -1 : (int)(k1[s] & 0xFFFFu);       // (no target, or only targets at distance 256: rejected by the threshold either way)
      int res = -1;
      if (i < n1 && bi >= 0 && b1 <= th && (float)b1 < __fmul_rn(ratio, (float)b2)) {
        res = bi;
        int bin = 0;
        if (check_ori) {
          float rot = __fsub_rn(KA[i].angle, KB[bi].angle);
          if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
          bin = (int)roundf(__fmul_rn(rot, factor));
          if (bin == HISTO_LENGTH) bin = 0;
          atomicAdd(&s_hist[bin], 1);
        }
        res |= bin << 24;                                   // stash the bin (indices < 2^24)
      }
      // (split pairs: agent-scope stores, read back with agent-scope loads by the last workgroup - no __threadfence(), which writes
      // back the XCD's whole L2; see k_match_pairs_mfma)
      if (i < cap) { if (nsplit > 1) __hip_atomic_store(&M[i], res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else M[i] = res; }
    }
  }
  __syncthreads();
  if (nsplit > 1) {
    // merge: partial bins -> the pair's scratch row; the last workgroup of the pair to arrive finishes it
    int* sc = scratch + (size_t)p * 32;
    if (tid < HISTO_LENGTH && s_hist[tid]) __hip_atomic_fetch_add(&sc[tid], s_hist[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    HANDOFF_DRAIN();                                          // this thread's stores and atomics have reached the memory side (handoff.h)
    __syncthreads();
    if (tid == 0) s_last = (__hip_atomic_fetch_add(&sc[31], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsplit - 1);
    __syncthreads();
    if (!s_last) return;
    HANDOFF_ACQUIRE();                                        // (the loads below stay behind the ticket)
    if (tid < HISTO_LENGTH) { s_hist[tid] = __hip_atomic_load(&sc[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(&sc[tid], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    if (tid == 31) __hip_atomic_store(&sc[31], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }
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
    int r = nsplit > 1 ? __hip_atomic_load(&M[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : M[i];       // (other workgroups wrote part of M)
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

// ---- the same frame-pair matcher with the distances on the MATRIX cores -------------------------------------------------
// All-pairs Hamming distance is a matrix product: d(a, b) = |a| + |b| - 2 |a & b|, and |a & b| is the dot product of the two
// descriptors written out as 256 bytes of 0 / 1.  v_mfma_i32_16x16x64_i8 takes 16 queries x 16 targets x 64 bits per instruction
// (~18 cycles of a SIMD, tools/ubench/mfma_i8.hip: 4.7 POPS): 4 instructions per 256 distances, against 19 VALU instructions per
// distance and lane (xor + popcount chain + key + min / med3) that held k_match_pairs at ~73 % of the chip's VALU issue rate.
// The packed key of the best / second-best update comes out of the product itself: query bits are bytes of {0, -128}, target
// bits bytes of {0, 4} (a common bit contributes -512 = -2 * 256), and the accumulator of a (query, target tile) product STARTS
// at |b| * 256 + tile index - so the result is (d - |a|) * 256 + tile: ordered like (d, target index) for the lane that owns the
// column (its targets are 16 tile + lane & 15), |a| is a row constant that is added at the end.  What is left for the VALU is
// one v_med3_i32 + one v_min_i32 per distance, the same update as above (signed: d - |a| may be negative).
// Workgroup = 512 threads, 8 waves x 64 queries (four 16-row tiles, their expanded bits live in 64 VGPRs); the targets stream
// through LDS in chunks of MM_CHUNK, expanded bit -> byte by all threads one chunk ahead of the products (two buffers, one
// barrier per chunk); a wave reads a 16-target tile (4 x ds_read_b128 per lane + the start values) once for its four query
// tiles.  C/D layout (col = lane & 15 = target, row = 4 (lane >> 4) + r = query): a lane sees one target column per tile and
// keeps (k1, k2) for its 16 rows; the 16 lanes of a DPP row are merged once at the end.  The k index inside an instruction is
// the same function of (lane >> 4, byte) for A and B, so any consistent bit -> byte order gives the full sum.
// 256 pairs of ~1816 x 1816: 0.205 ms against 0.506 ms (tools/match_ab.py, identical match lists); the products alone would
// take ~0.11 ms.  The VALU does NOT co-issue with this MFMA form (tools/ubench/mfma_valu_mix.hip: 19.6 cycles per product alone,
// +4 per VALU instruction beside it), so the key updates and the expansion add to the products' time: PMC 42 % + 42 % busy.
// Measured and not adopted: the 32x32x32 form, with which independent VALU work does co-issue in the microbenchmark - 32-query
// tiles need 64 key registers per lane, the target operands then have to be re-read per half tile, and the kernel took 0.249 ms
// (0.117 ms with the key updates removed: the products and the staging fit, the keys did not hide under them).
#ifndef MM_THREADS
#define MM_THREADS 512
#endif
#define MM_IT (MM_CHUNK * 16 / MM_THREADS)      /* 16-bit pieces a thread expands per chunk */
#define MM_QT 4                      // 16-query tiles per wave
#ifndef MM_CHUNK
#define MM_CHUNK 128                 // targets per LDS chunk (one barrier per chunk: 64 -> 128 targets 0.215 -> 0.205 ms per 256 pairs)
#endif
#define MM_TSTRIDE 272               // bytes per expanded target in LDS (256 + 16: the 16 lanes of a tile read 16 different bank groups)
typedef int mm_v4i __attribute__((ext_vector_type(4)));
// 16 bits -> 16 bytes of {0, 4}: a nibble times (1 + 2^7 + 2^14 + 2^21) << 2 puts bit i at bit 8 i + 2 (the partial products do not
// overlap; both factors fit v_mul_u32_u24)
__device__ __forceinline__ mm_v4i mm_expand16(uint32_t h) {
  mm_v4i r;
  r.x = (int)(__umul24(h & 0xFu, 0x810204u) & 0x04040404u);
  r.y = (int)(__umul24((h >> 4) & 0xFu, 0x810204u) & 0x04040404u);
  r.z = (int)(__umul24((h >> 8) & 0xFu, 0x810204u) & 0x04040404u);
  r.w = (int)(__umul24((h >> 12) & 0xFu, 0x810204u) & 0x04040404u);
  return r;
}
__device__ __forceinline__ int med3_i32(int a, int b, int c) {
  int r;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
#define MM_NONE 0x3FFFFFFF            /* no candidate: tile index 255 (at most 255 target tiles: cap <= 4080) */
__global__ __launch_bounds__(MM_THREADS) void k_match_pairs_mfma(const orbx_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                                const int* __restrict__ counts, int cap,
                                                                const int* __restrict__ pair_a, const int* __restrict__ pair_b,
                                                                float ratio, int th, int check_ori,
                                                                int* __restrict__ match12, int* __restrict__ nmatch,
                                                                int nsplit, int* __restrict__ scratch /*[npairs][32]*/) {
  __shared__ __attribute__((aligned(16))) uint8_t s_b[2][MM_CHUNK * MM_TSTRIDE];
  __shared__ mm_v4i s_c0[2][MM_CHUNK];
  __shared__ int s_key1[MM_THREADS], s_key2[MM_THREADS];
  __shared__ int s_hist[HISTO_LENGTH], s_keep[3], s_n, s_last;
  const int p = (int)blockIdx.x / nsplit, sp = (int)blockIdx.x % nsplit, tid = threadIdx.x;
  const int fa = pair_a[p], fb = pair_b[p];
  int n1 = counts[fa], n2 = counts[fb];
  n1 = n1 < 0 ? 0 : min(n1, cap);
  n2 = n2 < 0 ? 0 : min(n2, cap);
  const uint8_t* Qb = desc + (size_t)fa * cap * 32;
  const uint8_t* Tb = desc + (size_t)fb * cap * 32;
  const unsigned short* T16 = (const unsigned short*)Tb;
  const orbx_keypoint* KA = kps + (size_t)fa * cap;
  const orbx_keypoint* KB = kps + (size_t)fb * cap;
  int* M = match12 + (size_t)p * cap;
  if (tid < HISTO_LENGTH) s_hist[tid] = 0;
  if (tid == 0) s_n = 0;
  const int lane = tid & 63, wv = tid >> 6, lj = lane & 15, lg = lane >> 4;
  const int nchunks = (n2 + MM_CHUNK - 1) / MM_CHUNK;
  const float factor = 1.0f / HISTO_LENGTH;
  // a thread's share of a chunk: MM_IT 16-bit pieces to expand, and (threads 0 .. MM_CHUNK - 1) one target's weight
  auto fetch = [&](int c, uint32_t (&h)[MM_IT], uint4& w0, uint4& w1) {
#pragma unroll
    for (int u = 0; u < MM_IT; u++) {
      const int item = tid + MM_THREADS * u, tg = c * MM_CHUNK + (item >> 4);
      h[u] = tg < n2 ? (uint32_t)T16[(size_t)tg * 16 + (item & 15)] : 0u;
    }
    w0 = make_uint4(0, 0, 0, 0); w1 = w0;
    if (tid < MM_CHUNK && c * MM_CHUNK + tid < n2) { const uint4* t4 = (const uint4*)(Tb + (size_t)(c * MM_CHUNK + tid) * 32); w0 = t4[0]; w1 = t4[1]; }
  };
  auto stage = [&](int c, const uint32_t (&h)[MM_IT], const uint4& w0, const uint4& w1) {
    uint8_t* B = s_b[c & 1];
#pragma unroll
    for (int u = 0; u < MM_IT; u++) {
      const int item = tid + MM_THREADS * u;
      *(mm_v4i*)(B + (item >> 4) * MM_TSTRIDE + (item & 15) * 16) = mm_expand16(h[u]);
    }
    if (tid < MM_CHUNK) {
      const int j = c * MM_CHUNK + tid;
      const int nb = __popc(w0.x) + __popc(w0.y) + __popc(w0.z) + __popc(w0.w) + __popc(w1.x) + __popc(w1.y) + __popc(w1.z) + __popc(w1.w);
      const int c0 = j < n2 ? ((nb << 8) | (c * (MM_CHUNK / 16) + (tid >> 4))) : MM_NONE;      // |b| * 256 + the tile's index
      s_c0[c & 1][tid] = (mm_v4i){c0, c0, c0, c0};
    }
  };
  // The workgroup takes the query blocks sp, sp + nsplit, ... of MM_THREADS queries each (nsplit = 1 when the launch has a pair
  // per CU: no merge through global memory then, and the per-workgroup latencies - counts, first operands, the final pass - are
  // paid once per pair instead of once per block; the targets are expanded again for every block, ~15 % of a block's time).
  const int nblocks = (n1 + MM_THREADS - 1) / MM_THREADS;
  for (int blk = sp; blk < max(nblocks, sp + 1); blk += nsplit) {
    const int qbase = blk * MM_THREADS + wv * 64;               // this wave's 64 queries
    const bool wave_live = qbase < n1;
    // the queries of this wave as {0, -128} bytes: tile t, instruction s: bits [64 s + 16 lg, + 16) of query qbase + 16 t + lj
    mm_v4i A[MM_QT][4];
#pragma unroll
    for (int t = 0; t < MM_QT; t++) {
      const int qi = qbase + 16 * t + lj;
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) {
        const uint32_t h = qi < n1 ? (uint32_t)((const unsigned short*)(Qb + (size_t)qi * 32))[4 * s4 + lg] : 0u;
        const mm_v4i e = mm_expand16(h);
        A[t][s4].x = e.x << 5; A[t][s4].y = e.y << 5; A[t][s4].z = e.z << 5; A[t][s4].w = e.w << 5;          // 0x04 -> 0x80 = -128
      }
    }
    int k1[MM_QT][4], k2[MM_QT][4];
#pragma unroll
    for (int t = 0; t < MM_QT; t++) {
#pragma unroll
      for (int r = 0; r < 4; r++) { k1[t][r] = MM_NONE; k2[t][r] = MM_NONE; }
    }
    uint32_t hb[MM_IT]; uint4 w0, w1;
    if (nchunks > 0) { fetch(0, hb, w0, w1); stage(0, hb, w0, w1); }
    __syncthreads();
    for (int c = 0; c < nchunks; c++) {
      if (c + 1 < nchunks) fetch(c + 1, hb, w0, w1);
      if (wave_live) {
        // Software pipeline over the chunk's 16-target tiles: the NEXT tile's operands (4 x ds_read_b128 + the accumulator start
        // values) are requested before the current tile's 16 products are issued, and the key updates of the PREVIOUS tile's
        // results (32 VALU instructions) are interleaved with them, two per product - the matrix pipe (~18 cycles per
        // instruction) is not waited for.
        const uint8_t* B = s_b[c & 1];
        auto loadB = [&](int tt, mm_v4i (&Bv)[4], mm_v4i& c0) {
#pragma unroll
          for (int s4 = 0; s4 < 4; s4++) Bv[s4] = *(const mm_v4i*)(B + (tt * 16 + lj) * MM_TSTRIDE + (4 * s4 + lg) * 16);
          c0 = s_c0[c & 1][tt * 16 + lj];
        };
        auto products = [&](const mm_v4i (&Bv)[4], const mm_v4i& c0, mm_v4i (&C)[MM_QT]) {
#pragma unroll
          for (int t = 0; t < MM_QT; t++) C[t] = c0;
#pragma unroll
          for (int s4 = 0; s4 < 4; s4++) {
#pragma unroll
            for (int t = 0; t < MM_QT; t++) C[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t][s4], Bv[s4], C[t], 0, 0, 0);
          }
        };
        auto keys = [&](const mm_v4i (&C)[MM_QT]) {
#pragma unroll
          for (int t = 0; t < MM_QT; t++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int k = C[t][r];
              k2[t][r] = med3_i32(k1[t][r], k2[t][r], k);
              k1[t][r] = min(k1[t][r], k);
            }
          }
        };
        auto interleave = [&]() {
          __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
#pragma unroll
          for (int i = 0; i < 16; i++) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 2, 0); }
          __builtin_amdgcn_sched_barrier(0);
        };
        constexpr int NT = MM_CHUNK / 16;                        // tiles per chunk (even)
        mm_v4i Ba[4], Bb[4], ca, cb, Ca[MM_QT], Cb[MM_QT];
        loadB(0, Ba, ca);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tp = 0; tp < NT; tp += 2) {
          loadB(tp + 1, Bb, cb);
          products(Ba, ca, Ca);
          if (tp > 0) keys(Cb);
          interleave();
          if (tp + 2 < NT) loadB(tp + 2, Ba, ca);
          products(Bb, cb, Cb);
          keys(Ca);
          interleave();
        }
        keys(Cb);
      }
      if (c + 1 < nchunks) stage(c + 1, hb, w0, w1);
      __syncthreads();
    }
    // merge the 16 lanes that share a row (same lane >> 4): k1 = min, k2 = min(k2a, k2b, max(k1a, k1b))
#pragma unroll
    for (int t = 0; t < MM_QT; t++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        // (d - |a|) * 256 + tile  ->  (d - |a|) * 4096 + target index (tile * 16 + this lane's column)
        auto widen = [&](int k) { return (k & 0xFF) == 0xFF ? MM_NONE : (((k >> 8) << 12) | ((k & 0xFF) << 4) | lj); };
        int a1 = widen(k1[t][r]), a2 = widen(k2[t][r]);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          const int b1 = __shfl_xor(a1, o), b2 = __shfl_xor(a2, o);
          a2 = min(min(a2, b2), max(a1, b1));
          a1 = min(a1, b1);
        }
        if (lj == 0) { s_key1[wv * 64 + 16 * t + 4 * lg + r] = a1; s_key2[wv * 64 + 16 * t + 4 * lg + r] = a2; }
      }
    }
    __syncthreads();
    {
      const int i = blk * MM_THREADS + tid;
      int res = -1;
      if (i < n1) {
        const uint4* q4 = (const uint4*)(Qb + (size_t)i * 32);
        const uint4 a0 = q4[0], a1 = q4[1];
        const int na = __popc(a0.x) + __popc(a0.y) + __popc(a0.z) + __popc(a0.w) + __popc(a1.x) + __popc(a1.y) + __popc(a1.z) + __popc(a1.w);
        const int e1 = s_key1[tid], e2 = s_key2[tid];
        const int bi = (e1 & 0xFFF) == 0xFFF ? -1 : (e1 & 0xFFF);
        const int b1 = (e1 >> 12) + na;
        const int b2 = (e2 & 0xFFF) == 0xFFF ? 256 : (e2 >> 12) + na;
        if (bi >= 0 && b1 <= th && (float)b1 < __fmul_rn(ratio, (float)b2)) {
          res = bi;
          int bin = 0;
          if (check_ori) {
            float rot = __fsub_rn(KA[i].angle, KB[bi].angle);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            bin = (int)roundf(__fmul_rn(rot, factor));
            if (bin == HISTO_LENGTH) bin = 0;
            atomicAdd(&s_hist[bin], 1);
          }
          res |= bin << 24;
        }
      }
      // (agent-scope stores and loads - written through to / read from the memory side - instead of __threadfence(): the fence
      // writes back the XCD's whole L2, 0.19 ms of a 0.54 ms launch when 1024 workgroups do it twice each)
      if (i < cap) { if (nsplit > 1) __hip_atomic_store(&M[i], res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else M[i] = res; }
    }
  }
  // (entries beyond the last block of queries)
  if (sp == 0) for (int i = max(nblocks, 1) * MM_THREADS + tid; i < cap; i += MM_THREADS) { if (nsplit > 1) __hip_atomic_store(&M[i], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else M[i] = -1; }
  __syncthreads();
  if (nsplit > 1) {
    int* sc = scratch + (size_t)p * 32;
    if (tid < HISTO_LENGTH && s_hist[tid]) __hip_atomic_fetch_add(&sc[tid], s_hist[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    HANDOFF_DRAIN();                                          // this thread's stores and atomics have reached the memory side (handoff.h)
    __syncthreads();
    if (tid == 0) s_last = (__hip_atomic_fetch_add(&sc[31], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsplit - 1);
    __syncthreads();
    if (!s_last) return;
    HANDOFF_ACQUIRE();                                        // (the loads below stay behind the ticket)
    if (tid < HISTO_LENGTH) { s_hist[tid] = __hip_atomic_load(&sc[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(&sc[tid], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    if (tid == 31) __hip_atomic_store(&sc[31], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }
  if (tid == 0) {
    int i1 = -1, i2 = -1, i3 = -1;
    if (check_ori) {
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
  for (int i = tid; i < cap; i += MM_THREADS) {
    int r = nsplit > 1 ? __hip_atomic_load(&M[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : M[i];
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

// ---- brute force for ONE query set, split so that a single 2000 x 2000 match fills the chip ------------------------
// 64 queries per workgroup (lane = query, descriptor in 8 VGPRs), its 16 waves take the targets j = w, w + 16, ... : the
// target index is wave-uniform, so the descriptors arrive through the scalar unit (s_load, no LDS staging), and every wave
// keeps (best, second best) as packed keys (distance << 16 | index) - "first minimum wins" is the minimum key, whatever
// order the slices are merged in.  The 16 partial pairs are merged through LDS: k1 = min(k1a, k1b),
// k2 = min(k2a, k2b, max(k1a, k1b)).  2000 queries -> 32 workgroups x 16 waves instead of 8 x 4.
#define BS_WAVES 16
__global__ __launch_bounds__(64 * BS_WAVES) void k_best2_split(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t, int nt,
                                                               int* __restrict__ best_idx, int* __restrict__ best_d, int* __restrict__ second_d) {
  __shared__ uint32_t s_k1[BS_WAVES][64], s_k2[BS_WAVES][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
  if (i < nq) { q0 = ((const uint4*)q)[2 * (size_t)i]; q1 = ((const uint4*)q)[2 * (size_t)i + 1]; }
  uint32_t k1 = (256u << 16) | 0xFFFFu, k2 = k1;
  const uint4* T = (const uint4*)t;
  for (int j = w; j < nt; j += BS_WAVES) {
    const uint4 t0 = T[2 * (size_t)j], t1 = T[2 * (size_t)j + 1];
    const uint32_t k = ((uint32_t)hamming256(q0, q1, t0, t1) << 16) | (uint32_t)j;
    k2 = med3_u32(k1, k2, k);
    k1 = min(k1, k);
  }
  s_k1[w][lane] = k1; s_k2[w][lane] = k2;
  __syncthreads();
  if (w == 0 && i < nq) {
#pragma unroll
    for (int u = 1; u < BS_WAVES; u++) {
      const uint32_t a1 = s_k1[u][lane], a2 = s_k2[u][lane];
      k2 = min(min(k2, a2), max(k1, a1));
      k1 = min(k1, a1);
    }
    const int b1 = (int)(k1 >> 16);
    best_idx[i] = (k1 & 0xFFFFu) == 0xFFFFu ? -1 : (int)(k1 & 0xFFFFu);
    best_d[i] = b1 > 256 ? 256 : b1;
    second_d[i] = min((int)(k2 >> 16), 256);
  }
}

// every (query, candidate) distance of a CSR list whose total is only known on the device (off[nq]); grid sized for `cap`
// (the lists are {candidate index, distance} pairs: pair[c].x is given, pair[c].y is written here)
__global__ __launch_bounds__(256) void k_dist_csr_dev(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t,
                                                      const uint32_t* __restrict__ off, uint2* __restrict__ pair, uint32_t cap) {
  const uint32_t c = blockIdx.x * 256u + threadIdx.x;
  const uint32_t total = min(off[nq], cap);
  if (c >= total) return;
  int lo = 0, hi = nq;                 // largest i with off[i] <= c
  while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (off[mid] <= c) lo = mid; else hi = mid; }
  const uint32_t j = pair[c].x;
  pair[c].y = (uint32_t)hamming256(((const uint4*)q)[2 * (size_t)lo], ((const uint4*)q)[2 * (size_t)lo + 1],
                       ((const uint4*)t)[2 * (size_t)j], ((const uint4*)t)[2 * (size_t)j + 1]);
}

// A guided search only ever looks at candidates within a distance bound (the acceptance threshold, divided by the ratio where a
// second-best test exists: a farther candidate can be neither the best nor a second best that matters), and a window holds few
// of those - random descriptors are 128 +- 8 bits apart.  One wave per query: count the candidates within `keep`, and (after a
// scan of the counts) write them densely, in list order, so that the download and the host pass see a few thousand pairs
// instead of a few hundred thousand (SearchForInitialization, window 100: 600 k pairs, 4.8 MB -> ~50 kB).
template <bool FILL>
__global__ __launch_bounds__(256) void k_filter_lists(const uint32_t* __restrict__ off, const uint2* __restrict__ pair, uint32_t cap, int nq, uint32_t keep,
                                                      int* __restrict__ fcnt, const uint32_t* __restrict__ foff, uint2* __restrict__ fpair, uint32_t fcap) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= nq) return;
  const uint32_t b = min(off[q], cap), e = min(off[q + 1], cap);
  uint32_t found = 0;
  const uint32_t base_out = FILL ? foff[q] : 0u;
  for (uint32_t base = b; base < e; base += 64) {
    const uint32_t k = base + lane;
    uint2 v = make_uint2(0u, 0xFFFFFFFFu);
    if (k < e) v = pair[k];
    const bool ok = k < e && v.y <= keep;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
    if (FILL) { const uint32_t at = base_out + found + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); if (ok && at < fcap) fpair[at] = v; }
    found += (uint32_t)__popcll(m);
  }
  if (!FILL && lane == 0) fcnt[q] = (int)found;
}

// ---- rotation-consistency bookkeeping of the guided searches (host; tiny) ----------------------------------------------
// Bin of a match: rot = a1 - a2 (+360 if negative), bin = round(rot / 30), 30 -> 0 (e.g. src/ORBmatcher.cc:431-437).
static inline int rotation_bin(float a1, float a2) {
  float rot = a1 - a2;
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)std::round(rot * (1.0f / HISTO_LENGTH));
  return bin == HISTO_LENGTH ? 0 : bin;
}
// ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:1386-1418) on bin populations: which bins survive.
struct RotationFilter {
  int cnt[HISTO_LENGTH];
  RotationFilter() { for (int& c : cnt) c = 0; }
  void add(int bin) { cnt[bin]++; }
  // keep[b] = bin b is one of the (up to) three most populated ones after the 10 % rule
  void kept(bool keep[HISTO_LENGTH]) const {
    int top[3] = {-1, -1, -1}, pop[3] = {0, 0, 0};
    for (int b = 0; b < HISTO_LENGTH; b++) {                 // strict '>' : the earlier bin wins ties, as the reference's scan
      int r = 3;
      while (r > 0 && cnt[b] > pop[r - 1]) r--;
      if (r == 3) continue;
      for (int m = 2; m > r; m--) { top[m] = top[m - 1]; pop[m] = pop[m - 1]; }
      top[r] = b; pop[r] = cnt[b];
    }
    if ((float)pop[1] < 0.1f * (float)pop[0]) { top[1] = -1; top[2] = -1; }
    else if ((float)pop[2] < 0.1f * (float)pop[0]) { top[2] = -1; }
    for (int b = 0; b < HISTO_LENGTH; b++) keep[b] = (b == top[0] || b == top[1] || b == top[2]);
  }
};

// ---- window candidates + all their distances, device-side, one synchronisation -----------------------------------------
// Frame grid (Frame::AssignFeaturesToGrid), Frame::GetFeaturesInArea lists in the reference's order and the Hamming
// distance of every (query, candidate) pair.  Results land in pinned host memory owned by the thread workspace:
// off[nq + 1], idx[total], dist[total].
struct Candidate { uint32_t idx; int32_t dist; };
struct WindowLists { const uint32_t* off = nullptr; const Candidate* cand = nullptr; uint32_t total = 0; };
static thread_local FrameGridDev g_grid;                       // device grid of the calling thread (buffers grow, are reused)
static thread_local uint32_t g_cand_cap = 0;                   // candidate capacity that was enough so far
static thread_local int g_grid_device = -1;                    // device g_grid's buffers live on

static thread_local uint32_t g_fcand_cap = 0;                  // kept-candidate capacity that was enough so far
// keep < 0: every candidate comes back; keep >= 0: only the candidates within that distance (dense, list order preserved)
static int window_lists(ThreadWs& W, const float* kps4, const uint8_t* desc, int n, const float* bounds, const float* q_uv, const float* q_radius,
                        const int32_t* q_minl, const int32_t* q_maxl, const uint8_t* q_valid, const uint8_t* q_desc, int nq, WindowLists* out, int keep = -1) {
  for (int attempt = 0; attempt < 2; attempt++) {
    int rc = W.begin();
    if (rc) return rc;
    if (g_grid_device != W.device) {                             // default device changed: like ThreadWs, forget (leak) the old device's buffers
      g_grid = FrameGridDev(); g_cand_cap = 0; g_fcand_cap = 0; g_grid_device = W.device;
    }
    const uint32_t cap = std::max<uint32_t>(g_cand_cap, (uint32_t)nq * 64u);
    ThreadWs::Pack in;                                           // every input in one pinned block, one H2D copy
    const int pk = in.add(kps4, 16 * (size_t)n), pd = in.add(desc, 32 * (size_t)n), pq = in.add(q_uv, 8 * (size_t)nq), pr = in.add(q_radius, 4 * (size_t)nq),
              pqd = in.add(q_desc, 32 * (size_t)nq), pmn = q_minl ? in.add(q_minl, 4 * (size_t)nq) : -1, pmx = q_maxl ? in.add(q_maxl, 4 * (size_t)nq) : -1,
              pv = q_valid ? in.add(q_valid, (size_t)nq) : -1;
    if ((rc = W.commit(in))) return rc;
    // outputs in one device block [off (nq + 1) | {idx, dist} pairs (cap)]: its used prefix comes back with one D2H copy
    const size_t off_words = ((size_t)nq + 1 + 1) & ~(size_t)1;  // (pairs 8-byte aligned)
    int* dcnt = W.d<int>(nq, &rc);
    uint32_t* dout = W.d<uint32_t>(off_words + 2 * (size_t)cap, &rc);
    if (rc) return rc;
    uint32_t* doff = dout; uint2* dpair = (uint2*)(dout + off_words);
    const float* dk = in.dev<float>(pk);
    const float gb[4] = {bounds[0], bounds[1], bounds[2], bounds[3]};
    if ((rc = frame_grid_build(g_grid, dk, n, gb, W.s))) return rc;
    if ((rc = frame_area_candidates_enqueue(g_grid, dk, in.dev<float>(pq), in.dev<float>(pr), in.dev<int>(pmn), in.dev<int>(pmx), in.dev<uint8_t>(pv), nq,
                                            dcnt, doff, (uint32_t*)dpair, cap, 2, W.s))) return rc;
    hipLaunchKernelGGL(k_dist_csr_dev, dim3((cap + 255) / 256), dim3(256), 0, W.s, in.dev<uint8_t>(pqd), nq, in.dev<uint8_t>(pd), doff, dpair, cap);
    if (keep >= 0) {
      // filtered form: [foff (nq + 1) | kept pairs]; the full lists stay on the device
      const uint32_t fcap = std::max<uint32_t>(g_fcand_cap, (uint32_t)nq * 8u);
      int* dfc = W.d<int>(nq, &rc);
      uint32_t* dfo = W.d<uint32_t>(off_words + 2 * (size_t)fcap + 2, &rc);
      uint32_t* hf = W.h<uint32_t>(off_words + 2 * (size_t)fcap + 2, &rc);
      if (rc) return rc;
      uint2* dfp = (uint2*)(dfo + off_words);
      hipLaunchKernelGGL(k_filter_lists<false>, dim3((nq + 3) / 4), dim3(256), 0, W.s, doff, dpair, cap, nq, (uint32_t)keep, dfc, (const uint32_t*)nullptr, (uint2*)nullptr, 0u);
      if ((rc = frame_scan_enqueue(dfc, nq, dfo, W.s))) return rc;
      hipLaunchKernelGGL(k_filter_lists<true>, dim3((nq + 3) / 4), dim3(256), 0, W.s, doff, dpair, cap, nq, (uint32_t)keep, (int*)nullptr, dfo, dfp, fcap);
      // the true total of the unfiltered lists travels in the spare word behind the offsets (off_words >= nq + 1 ... nq + 2)
      ORBHIP_CHECK_HIP(hipMemcpyAsync(dfo + off_words + 2 * (size_t)fcap, doff + nq, 4, hipMemcpyDeviceToDevice, W.s));
      const uint32_t fguess = std::min<uint32_t>(fcap, std::max<uint32_t>(g_fcand_cap, (uint32_t)nq * 4u));
      ORBHIP_CHECK_HIP(hipMemcpyAsync(hf, dfo, (off_words + 2 * (size_t)fguess) * 4, hipMemcpyDeviceToHost, W.s));
      ORBHIP_CHECK_HIP(hipMemcpyAsync(hf + off_words + 2 * (size_t)fcap, dfo + off_words + 2 * (size_t)fcap, 4, hipMemcpyDeviceToHost, W.s));
      if ((rc = W.sync())) return rc;
      const uint32_t total_all = hf[off_words + 2 * (size_t)fcap], ftotal = hf[nq];
      if (total_all > cap) { g_cand_cap = total_all + total_all / 4; continue; }
      if (ftotal > fcap) { g_fcand_cap = ftotal + ftotal / 4; continue; }
      if (ftotal > fguess) {
        ORBHIP_CHECK_HIP(hipMemcpyAsync(hf + off_words + 2 * (size_t)fguess, dfo + off_words + 2 * (size_t)fguess, 2 * (size_t)(ftotal - fguess) * 4, hipMemcpyDeviceToHost, W.s));
        if ((rc = W.sync())) return rc;
      }
      g_cand_cap = std::max<uint32_t>(g_cand_cap, total_all + total_all / 8);
      g_fcand_cap = std::max<uint32_t>(g_fcand_cap, ftotal + ftotal / 8);
      out->off = hf; out->cand = (const Candidate*)(hf + off_words); out->total = ftotal;
      return 0;
    }
    // the lists are usually far shorter than the capacity: download what the previous call needed (+ slack), the rest only if used
    const uint32_t guess = std::min<uint32_t>(cap, std::max<uint32_t>(g_cand_cap, (uint32_t)nq * 16u));
    uint32_t* hout = W.h<uint32_t>(off_words + 2 * (size_t)cap, &rc);
    if (rc) return rc;
    ORBHIP_CHECK_HIP(hipMemcpyAsync(hout, dout, (off_words + 2 * (size_t)guess) * 4, hipMemcpyDeviceToHost, W.s));
    if ((rc = W.sync())) return rc;
    const uint32_t total = hout[nq];
    if (total > cap) { g_cand_cap = total + total / 4; continue; }          // (rare) lists longer than the buffer: once more, larger
    if (total > guess) {
      ORBHIP_CHECK_HIP(hipMemcpyAsync(hout + off_words + 2 * (size_t)guess, dout + off_words + 2 * (size_t)guess, 2 * (size_t)(total - guess) * 4, hipMemcpyDeviceToHost, W.s));
      if ((rc = W.sync())) return rc;
    }
    g_cand_cap = std::max<uint32_t>(g_cand_cap, total + total / 8);
    out->off = hout; out->cand = (const Candidate*)(hout + off_words); out->total = total;
    return 0;
  }
  set_error("window candidate lists did not fit after regrowing");
  return ORBHIP_ENOMEM;
}

// all distances of host-built CSR lists (BoW family): uploads, one kernel, one download, one synchronisation
static int csr_distances(ThreadWs& W, const uint8_t* q, int nq, const uint8_t* t, int nt, const std::vector<uint32_t>& off,
                         const std::vector<uint32_t>& idx, const int** dist_out) {
  const uint32_t total = off[nq];
  *dist_out = nullptr;
  if (total == 0) return 0;
  int rc = W.begin();
  if (rc) return rc;
  ThreadWs::Pack in;
  const int pq = in.add(q, 32 * (size_t)nq), pt = in.add(t, 32 * (size_t)nt), po = in.add(off.data(), 4 * ((size_t)nq + 1)), pi = in.add(idx.data(), 4 * (size_t)total);
  if ((rc = W.commit(in))) return rc;
  int* dd = W.d<int>(total, &rc);
  if (rc) return rc;
  hipLaunchKernelGGL(k_dist_csr, dim3((total + 255) / 256), dim3(256), 0, W.s, in.dev<uint8_t>(pq), nq, in.dev<uint8_t>(pt), in.dev<uint32_t>(po),
                     in.dev<uint32_t>(pi), total, dd);
  const int* h = W.down(dd, total, &rc);
  if (rc) return rc;
  if ((rc = W.sync())) return rc;
  *dist_out = h;
  return 0;
}

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
  else if (nt < 65535)          // (packed 16-bit target index)
    hipLaunchKernelGGL(k_best2_split, dim3((nq + 63) / 64), dim3(64 * BS_WAVES), 0, st, d_q, nq, d_t, nt, d_best_idx, d_best_d, d_second_d);
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
  ThreadWs& W = thread_ws();
  int rc = W.begin();
  if (rc) return rc;
  const size_t total = off ? off[nq] : 0;
  ThreadWs::Pack in;
  const int pq = in.add(q, 32 * (size_t)nq), pt = in.add(t, 32 * (size_t)nt), po = off ? in.add(off, 4 * ((size_t)nq + 1)) : -1, pi = off ? in.add(idx, 4 * total) : -1;
  if ((rc = W.commit(in))) return rc;
  int32_t* dout = W.d<int32_t>(3 * (size_t)nq, &rc);
  if (rc) return rc;
  if ((rc = orbm_hamming_best2_device(in.dev<uint8_t>(pq), nq, in.dev<uint8_t>(pt), nt, in.dev<uint32_t>(po), in.dev<uint32_t>(pi), dout, dout + nq,
                                      dout + 2 * (size_t)nq, W.s))) return rc;
  const int32_t* h = W.down(dout, 3 * (size_t)nq, &rc);
  if (rc) return rc;
  if ((rc = W.sync())) return rc;
  std::memcpy(best_idx, h, (size_t)nq * 4); std::memcpy(best_d, h + nq, (size_t)nq * 4); std::memcpy(second_d, h + 2 * (size_t)nq, (size_t)nq * 4);
  return 0;
}

int orbm_match_frames_batch_device(const orbx_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_counts, int cap,
                                   const int32_t* d_pair_a, const int32_t* d_pair_b, int npairs, float ratio, int th,
                                   int check_ori, int32_t* d_match12, int32_t* d_nmatch, void* stream) {
  ORBHIP_REQUIRE(npairs >= 0 && cap > 0, ORBHIP_EINVAL, "bad size");
  if (npairs == 0) return 0;
  ORBHIP_REQUIRE(d_kps && d_desc && d_counts && d_pair_a && d_pair_b && d_match12 && d_nmatch, ORBHIP_EINVAL, "NULL argument");
  const size_t lds = (size_t)cap * 32;
  // ORBHIP_MATCH_MFMA=0: the VALU kernel (xor + popcount), otherwise the distances come from the matrix cores (the target index
  // takes 12 bits of the packed key there)
  static const bool use_mfma = []() { const char* e = std::getenv("ORBHIP_MATCH_MFMA"); return !(e && e[0] == '0'); }();
  const bool mfma = use_mfma && cap <= 4080;               // (at most 255 target tiles of 16: the tile index takes 8 bits of the packed key)
  ORBHIP_REQUIRE(mfma || lds <= 150 * 1024, ORBHIP_EINVAL, "per-frame capacity too large for the LDS-resident matcher (cap <= 4800)");
  ORBHIP_REQUIRE(cap < (1 << 24), ORBHIP_EINVAL, "cap too large");
  if (!mfma && lds > 64 * 1024) {                                   // (the dynamic-LDS opt-in is per (function, device): only ever raised)
    int dev = 0;
    ORBHIP_CHECK_HIP(hipGetDevice(&dev));
    if (int rc = raise_dynamic_lds((const void*)k_match_pairs<1>, dev, lds)) return rc;
    if (int rc = raise_dynamic_lds((const void*)k_match_pairs<2>, dev, lds)) return rc;
  }
  // workgroups per pair: one workgroup saturates a CU's issue slots (two per CU gain nothing: 256 pairs take 0.64 ms as 256
  // workgroups, 0.67 ms as 512), so a launch is split until it has about one workgroup per CU - 64 pairs: 0.63 ms unsplit,
  // 0.38 / 0.28 / 0.35 ms with 2 / 4 / 8 workgroups per pair.  ORBHIP_MATCH_SPLIT forces a value.
  static const int force_split = []() { const char* e = ORBHIP_EXP_ENV("ORBHIP_MATCH_SPLIT"); return e ? atoi(e) : 0; }();
  const int nsplit = mfma ? std::min((cap + MM_THREADS - 1) / MM_THREADS, force_split > 0 ? force_split : std::max(1, 320 / npairs)) : force_split > 0 ? force_split : std::min(8, std::max(1, 256 / npairs));
  int* scratch = nullptr;
  if (nsplit > 1) {
    // scratch rows {30 bins, -, ticket} per (device, stream): zero when allocated and left zero by every launch (the last
    // workgroup of a pair clears its row).  Launches that share a row set are ordered by their stream, so tickets and bins of
    // two launches in flight (other streams, other host threads) never mix.  A row set that has to grow is REPLACED, the old
    // allocation stays alive until process end (a launch enqueued earlier may still use it; growth is geometric, so the
    // retired memory is bounded by the final size).
    struct Rows { int* p = nullptr; int rows = 0; };
    static std::mutex mu; static std::map<std::pair<int, void*>, Rows> pool;
    int dev = 0;
    ORBHIP_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(mu);
    Rows& R = pool[std::make_pair(dev, stream)];
    if (npairs > R.rows) {
      const int want = std::max(std::max(npairs, 2 * R.rows), 256);
      void* q = nullptr;
      hipError_t e = hipMalloc(&q, (size_t)want * 32 * sizeof(int));
      if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", (size_t)want * 32 * sizeof(int), hipGetErrorString(e)); return ORBHIP_ENOMEM; }
      ORBHIP_CHECK_HIP(hipMemsetAsync(q, 0, (size_t)want * 32 * sizeof(int), (hipStream_t)stream));   // ordered before the launch below
      R.p = (int*)q; R.rows = want;
    }
    scratch = R.p;
  }
  if (mfma)
    hipLaunchKernelGGL(k_match_pairs_mfma, dim3(npairs * nsplit), dim3(MM_THREADS), 0, (hipStream_t)stream, d_kps, d_desc, d_counts,
                       cap, d_pair_a, d_pair_b, ratio, th, check_ori, d_match12, d_nmatch, nsplit, scratch);
  else if (nsplit > 1)
    hipLaunchKernelGGL(k_match_pairs<1>, dim3(npairs * nsplit), dim3(MP_THREADS), lds, (hipStream_t)stream, d_kps, d_desc, d_counts,
                       cap, d_pair_a, d_pair_b, ratio, th, check_ori, d_match12, d_nmatch, cap, nsplit, scratch);
  else
    hipLaunchKernelGGL(k_match_pairs<2>, dim3(npairs), dim3(MP_THREADS), lds, (hipStream_t)stream, d_kps, d_desc, d_counts,
                       cap, d_pair_a, d_pair_b, ratio, th, check_ori, d_match12, d_nmatch, cap, 1, scratch);
  ORBHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

// The projection family.  Device: grid, window lists, distances.  Host: the order-dependent pass, in query order - a target
// that received a map point is closed for later queries (`taken`), best / second best with the level rule of :109-113,
// Fuse's chi-square gate, and the rotation-consistency filter which re-opens the targets it rejects.
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
  WindowLists L;
  // (the reported best distance needs every candidate; without it only candidates that can be accepted, or be a second best that
  // vetoes an acceptance, matter: distance <= th, or <= th / ratio with the second-best rule)
  int keep = -1;
  if (!q_best_dist) { keep = th; if (mode_best2 && ratio > 0.f) keep = (int)std::ceil((float)th / ratio) + 1; if (mode_best2 && !(ratio > 0.f)) keep = -1; if (keep > 256) keep = -1; }
  if (int rc = window_lists(thread_ws(), kps4, desc, n, bounds, q_uv, q_radius, q_min_level, q_max_level, q_valid, q_desc, nq, &L, keep)) return rc;
  struct Level2 { int d = 256, lvl = -1; };                     // (distance, octave) of a candidate
  RotationFilter rot;
  std::vector<signed char> bin_of(nq, -1);
  int found = 0;
  for (int qi = 0; qi < nq; qi++) {
    Level2 first, second; int target = -1;
    const bool level_gate = q_pred_level && q_pred_level[qi] >= 0;
    for (uint32_t c = L.off[qi]; c < L.off[qi + 1]; c++) {
      const int t = (int)L.cand[c].idx;
      if (taken && taken[t]) continue;
      const int lvl = (int)kps4[4 * t + 2];
      if (level_gate && (lvl < q_pred_level[qi] - 1 || lvl > q_pred_level[qi])) continue;
      if (chi2_gate > 0.f) {
        const float ex = q_uv[2 * qi] - kps4[4 * t], ey = q_uv[2 * qi + 1] - kps4[4 * t + 1];
        const float e2 = ex * ex + ey * ey;
        if (e2 * inv_level_sigma2[lvl] > chi2_gate) continue;
      }
      const int d = L.cand[c].dist;
      if (d < first.d) { second = first; first.d = d; first.lvl = lvl; target = t; }
      else if (mode_best2 && d < second.d) { second.d = d; second.lvl = lvl; }
    }
    if (q_best_dist) q_best_dist[qi] = first.d;
    if (target < 0 || first.d > th) continue;
    if (mode_best2 && first.lvl == second.lvl && first.d > ratio * second.d) continue;
    q_match[qi] = target;
    // (q_valid bit 1: the query's map point has no observations - the reference assigns it but the feature stays open for
    // later queries, ":83-84", ":1220-1221")
    if (taken && !(q_valid && (q_valid[qi] & 2))) taken[target] = 1;
    found++;
    if (check_ori) { bin_of[qi] = (signed char)rotation_bin(q_angle[qi], kps4[4 * target + 3]); rot.add(bin_of[qi]); }
  }
  if (check_ori) {
    bool keep[HISTO_LENGTH];
    rot.kept(keep);
    for (int qi = 0; qi < nq; qi++)
      if (q_match[qi] >= 0 && !keep[bin_of[qi]]) { if (taken) taken[q_match[qi]] = 0; q_match[qi] = -2 - q_match[qi]; found--; }
  }
  *nmatches = found;
  return 0;
}

// ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:956-1159) from the two window searches on: every valid keyframe-1 feature
// looks for its map point in keyframe 2 (q12_*: projected position, radius th * scale[predicted level], predicted level;
// descriptor = q12_desc row = the map point's representative descriptor) and vice versa; best <=
// TH_HIGH each way with the [pred - 1, pred] level gate, no `taken` state; a pair survives iff both directions agree
// (:1145-1157).  match12[n1] = index in keyframe 2 or -1.  bounds1 / bounds2 = the image bounds of keyframe 1 / keyframe 2
// (each keyframe has its own grid: KeyFrame::GetFeaturesInArea of pKF2 for the 1->2 search :1022, of pKF1 for 2->1 :1102).
int orbm_search_by_sim3(const float* kps1, const uint8_t* desc1, int n1, const float* kps2, const uint8_t* desc2, int n2,
                        const float* bounds1, const float* bounds2, const float* q12_uv, const float* q12_radius, const int32_t* q12_pred,
                        const uint8_t* q12_valid, const uint8_t* q12_desc, const float* q21_uv, const float* q21_radius, const int32_t* q21_pred, const uint8_t* q21_valid,
                        const uint8_t* q21_desc, int32_t* match12, int* nfound) {
  ORBHIP_REQUIRE(n1 >= 0 && n2 >= 0 && nfound && (n1 == 0 || match12), ORBHIP_EINVAL, "bad size");
  *nfound = 0;
  for (int i = 0; i < n1; i++) match12[i] = -1;
  if (n1 == 0 || n2 == 0) return 0;
  ORBHIP_REQUIRE(kps1 && kps2 && desc1 && desc2 && bounds1 && bounds2 && q12_uv && q12_radius && q12_pred && q21_uv && q21_radius && q21_pred, ORBHIP_EINVAL, "NULL argument");
  std::vector<int32_t> m1(n1), m2(n2);
  int k = 0;
  const int TH_HIGH = 100;
  if (int rc = orbm_search_by_projection(kps2, desc2, n2, bounds2, q12_uv, q12_radius, nullptr, nullptr, q12_pred, q12_desc ? q12_desc : desc1, q12_valid, nullptr, n1,
                                         nullptr, 0.f, nullptr, 0, 1.f, TH_HIGH, 0, m1.data(), nullptr, &k)) return rc;
  if (int rc = orbm_search_by_projection(kps1, desc1, n1, bounds1, q21_uv, q21_radius, nullptr, nullptr, q21_pred, q21_desc ? q21_desc : desc2, q21_valid, nullptr, n2,
                                         nullptr, 0.f, nullptr, 0, 1.f, TH_HIGH, 0, m2.data(), nullptr, &k)) return rc;
  int found = 0;
  for (int i1 = 0; i1 < n1; i1++) {
    const int i2 = m1[i1];
    if (i2 >= 0 && m2[i2] == i1) { match12[i1] = i2; found++; }
  }
  *nfound = found;
  return 0;
}

}  // extern "C"

// ---- the BoW family: queries = features of set 1 in (node, list) order of the merge-walk over the two feature vectors ----
namespace {
struct NodeQueries { std::vector<int> q_feature; std::vector<uint32_t> off{0}, idx; };
// DBoW2::FeatureVector is a std::map: both inputs are ascending node ids; equal ids pair their lists (:170-176, :496-500)
void pair_node_lists(const uint32_t* n1, const uint32_t* o1, const uint32_t* x1, int c1, const uint32_t* n2, const uint32_t* o2, const uint32_t* x2, int c2,
                     const uint8_t* usable1, NodeQueries* Q) {
  int a = 0, b = 0;
  while (a < c1 && b < c2) {
    if (n1[a] < n2[b]) { a++; continue; }
    if (n2[b] < n1[a]) { b++; continue; }
    for (uint32_t e1 = o1[a]; e1 < o1[a + 1]; e1++) {
      const int f = (int)x1[e1];
      if (usable1 && !usable1[f]) continue;
      Q->q_feature.push_back(f);
      Q->idx.insert(Q->idx.end(), x2 + o2[b], x2 + o2[b + 1]);
      Q->off.push_back((uint32_t)Q->idx.size());
    }
    a++; b++;
  }
}
std::vector<uint8_t> gather_descriptors(const uint8_t* desc, const std::vector<int>& rows) {
  std::vector<uint8_t> out(rows.size() * 32);
  for (size_t i = 0; i < rows.size(); i++) std::memcpy(&out[32 * i], desc + 32 * (size_t)rows[i], 32);
  return out;
}
}  // namespace

extern "C" {

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
  NodeQueries Q;
  pair_node_lists(fv1_node, fv1_off, fv1_idx, fv1_n, fv2_node, fv2_off, fv2_idx, fv2_n, valid1, &Q);
  const int nq = (int)Q.q_feature.size();
  if (nq == 0) return 0;
  const std::vector<uint8_t> qd = gather_descriptors(desc1, Q.q_feature);
  const int* dist = nullptr;
  if (int rc = csr_distances(thread_ws(), qd.data(), nq, desc2, n2, Q.off, Q.idx, &dist)) return rc;
  std::vector<uint8_t> used2(n2, 0);                            // vpMapPointMatches[realIdxF] / vbMatched2
  std::vector<signed char> bin_of(n1, -1);
  RotationFilter rot;
  int found = 0;
  for (int k = 0; k < nq; k++) {
    const int f1 = Q.q_feature[k];
    int d1 = 256, d2 = 256, f2 = -1;
    for (uint32_t c = Q.off[k]; c < Q.off[k + 1]; c++) {
      const int cand = (int)Q.idx[c];
      if (used2[cand] || (valid2 && !valid2[cand])) continue;
      const int d = dist[c];
      if (d < d1) { d2 = d1; d1 = d; f2 = cand; }
      else if (d < d2) d2 = d;
    }
    const bool close_enough = strict ? d1 < th : d1 <= th;     // (:535 '<' between keyframes, :210 '<=' keyframe-frame)
    if (!close_enough || !(static_cast<float>(d1) < ratio * static_cast<float>(d2))) continue;
    match12[f1] = f2; used2[f2] = 1; found++;
    if (check_ori) { bin_of[f1] = (signed char)rotation_bin(angle1[f1], angle2[f2]); rot.add(bin_of[f1]); }
  }
  if (check_ori) {
    bool keep[HISTO_LENGTH];
    rot.kept(keep);
    for (int f1 = 0; f1 < n1; f1++)
      if (match12[f1] >= 0 && !keep[bin_of[f1]]) { match12[f1] = -1; found--; }
  }
  *nmatches = found;
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
  NodeQueries Q;
  pair_node_lists(fv1_node, fv1_off, fv1_idx, fv1_n, fv2_node, fv2_off, fv2_idx, fv2_n, unmapped1, &Q);     // features that already hold a MapPoint are skipped (:620-623)
  const int nq = (int)Q.q_feature.size();
  if (nq == 0) return 0;
  const std::vector<uint8_t> qd = gather_descriptors(desc1, Q.q_feature);
  const int* dist = nullptr;
  if (int rc = csr_distances(thread_ws(), qd.data(), nq, desc2, n2, Q.off, Q.idx, &dist)) return rc;
  std::vector<signed char> bin_of(n1, -1);
  RotationFilter rot;
  int found = 0;
  for (int k = 0; k < nq; k++) {
    const int f1 = Q.q_feature[k];
    const float x1 = kps1[4 * f1], y1 = kps1[4 * f1 + 1];
    // CheckDistEpipolarLine (:128-149): l = x1' F12, double products narrowed to float
    const float la = x1 * F12[0] + y1 * F12[3] + F12[6];
    const float lb = x1 * F12[1] + y1 * F12[4] + F12[7];
    const float lc = x1 * F12[2] + y1 * F12[5] + F12[8];
    const float den = la * la + lb * lb;
    int limit = TH_LOW, f2 = -1;                                // a later candidate at the SAME distance replaces the earlier one (:654)
    for (uint32_t c = Q.off[k]; c < Q.off[k + 1]; c++) {
      const int cand = (int)Q.idx[c];
      if (unmapped2 && !unmapped2[cand]) continue;               // (vbMatched2 is never set in this fork)
      const int d = dist[c];
      if (d > TH_LOW || d > limit) continue;
      const float x2 = kps2[4 * cand], y2 = kps2[4 * cand + 1];
      const int oct2 = (int)kps2[4 * cand + 2];
      const float dex = ex - x2, dey = ey - y2;
      if (dex * dex + dey * dey < 100 * scale_factors[oct2]) continue;      // too close to the epipole (:658-664)
      if (den == 0) continue;
      const float num = la * x2 + lb * y2 + lc;
      if (num * num / den < 3.84 * level_sigma2[oct2]) { f2 = cand; limit = d; }
    }
    if (f2 < 0) continue;
    match12[f1] = f2; found++;
    if (check_ori) { bin_of[f1] = (signed char)rotation_bin(kps1[4 * f1 + 3], kps2[4 * f2 + 3]); rot.add(bin_of[f1]); }
  }
  if (check_ori) {
    bool keep[HISTO_LENGTH];
    rot.kept(keep);
    for (int f1 = 0; f1 < n1; f1++)
      if (match12[f1] >= 0 && !keep[bin_of[f1]]) { match12[f1] = -1; found--; }
  }
  *nmatches = found;
  return 0;
}

// SearchForInitialization: the window lists come from the same device grid as every other guided search (queries = the
// level-0 features of frame 1 at their previously matched positions, window radius, level range [0, 0]); the host keeps the
// pass whose outcome depends on the order of frame 1: a frame-2 feature remembers the distance it was won with, a later
// query only takes it with a strictly smaller one and then displaces the earlier owner (:408, :421-427).
int orbm_search_for_initialization(const float* kps1, const uint8_t* desc1, int n1, const float* kps2,
                                   const uint8_t* desc2, int n2, const float* bounds2, float* prev_matched, int window,
                                   float nnratio, int check_ori, int32_t* matches12, int* nmatches) {
  ORBHIP_REQUIRE(n1 >= 0 && n2 >= 0 && nmatches, ORBHIP_EINVAL, "bad size");
  *nmatches = 0;
  if (n1 == 0) return 0;
  ORBHIP_REQUIRE(kps1 && desc1 && prev_matched && matches12 && bounds2 && (n2 == 0 || (kps2 && desc2)), ORBHIP_EINVAL, "NULL argument");
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  if (n2 == 0) return 0;
  std::vector<float> radius(n1, (float)window);
  std::vector<int32_t> level(n1);
  std::vector<uint8_t> level0(n1);
  for (int i = 0; i < n1; i++) { level[i] = (int32_t)kps1[4 * i + 2]; level0[i] = level[i] > 0 ? 0 : 1; }      // (:383-385)
  WindowLists L;
  // (a candidate farther than TH_LOW / nnratio can be neither the best nor a second best that fails the ratio test, :425-427)
  int keep = nnratio > 0.f ? (int)std::ceil((float)TH_LOW / nnratio) + 1 : -1;
  if (keep > 256) keep = -1;
  if (int rc = window_lists(thread_ws(), kps2, desc2, n2, bounds2, prev_matched, radius.data(), level.data(), level.data(), level0.data(), desc1, n1, &L, keep)) return rc;
  std::vector<int> owner(n2, -1), won_with(n2, INT_MAX);
  std::vector<signed char> bin_of(n1, -1);
  int found = 0;
  for (int i1 = 0; i1 < n1; i1++) {
    int d1 = INT_MAX, d2 = INT_MAX, f2 = -1;
    for (uint32_t c = L.off[i1]; c < L.off[i1 + 1]; c++) {
      const int cand = (int)L.cand[c].idx, d = L.cand[c].dist;
      if (won_with[cand] <= d) continue;
      if (d < d1) { d2 = d1; d1 = d; f2 = cand; }
      else if (d < d2) d2 = d;
    }
    if (f2 < 0 || d1 > TH_LOW || !(d1 < (float)d2 * nnratio)) continue;
    if (owner[f2] >= 0) { matches12[owner[f2]] = -1; found--; }
    matches12[i1] = f2; owner[f2] = i1; won_with[f2] = d1; found++;
    if (check_ori) bin_of[i1] = (signed char)rotation_bin(kps1[4 * i1 + 3], kps2[4 * f2 + 3]);
  }
  if (check_ori) {
    // the reference's histogram holds EVERY accepted i1, displaced ones included (they stay in rotHist, :433-437): the
    // populations below count them, the removal only touches matches that still exist
    RotationFilter rot;
    for (int i1 = 0; i1 < n1; i1++) if (bin_of[i1] >= 0) rot.add(bin_of[i1]);
    bool keep[HISTO_LENGTH];
    rot.kept(keep);
    for (int i1 = 0; i1 < n1; i1++)
      if (bin_of[i1] >= 0 && !keep[bin_of[i1]] && matches12[i1] >= 0) { matches12[i1] = -1; found--; }
  }
  for (int i1 = 0; i1 < n1; i1++)
    if (matches12[i1] >= 0) { prev_matched[2 * i1] = kps2[4 * matches12[i1]]; prev_matched[2 * i1 + 1] = kps2[4 * matches12[i1] + 1]; }
  *nmatches = found;
  return 0;
}

}  // extern "C"
