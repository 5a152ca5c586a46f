// ============================================================================
// orb_track.hip -- the per-frame Tracking step with the motion model, device-resident from the image to the optimised pose
// (reference src/Tracking.cc:616-646: Frame construction - ORBextractor::operator(), AssignFeaturesToGrid -, then
// ORBmatcher::SearchByProjection(current_frame_, last_frame_, th) src/ORBmatcher.cc:1161-1271 and
// CeresOptimizer::PoseOptimization src/CeresOptimizer.cc:275-342).
//
// The separate entry points cost a host round trip each (orbx_extract 0.19 ms, orbm_search_by_projection 0.21 ms,
// ba_pose_optimization 0.14 ms, profiles/r02_api_latency.json) because every one uploads its inputs, synchronises and downloads.
// Here the frame's records never leave the device between the stages:
//   upload   the image, and ONE packed block with the predicted pose and the last frame's map-point arrays
//   kernels  extractor (orb_extractor.hip, on this call's stream) -> k_trk_prepare: projection of the last frame's points with
//            the reference's float / double mix, the current frame's {x, y, octave, angle} records and its 64 x 48 grid in the
//            reference's push_back order (one workgroup, the keypoint COUNT is read on the device) -> k_trk_windows: window
//            candidates + descriptor distances, the acceptable ones compacted per query -> k_trk_greedy: the reference's
//            sequential, order-dependent pass as dataflow / a parallel fixpoint (below), rotation histogram, slot ownership, the
//            observation list of PoseOptimization in feature order -> k_pose_lm (ba_solver.hip)
//   download ONE block: keypoints, descriptors, matches, slot owners, outlier flags, pose, counters.
//
// The greedy pass on the device.  The reference walks the last frame's features in index order; query q takes the closest
// candidate that no EARLIER query with an observed map point has claimed (first minimum in candidate order), and claims it
// unless its own map point has no observations.  That is a serial dictatorship, evaluated here in rounds: every unsettled query
// proposes its best still-available candidate.  A claimer can be settled with its proposal when it is the first to want it AND
// no earlier claimer that stays unsettled has that target anywhere on its list (such a one could fall back to it later); the
// set of claimers that stay unsettled is found from the losers outwards (they sign every target they could still take, a
// winner whose target carries an earlier signature joins them, until nothing changes).  The earliest unsettled claimer is
// always settled, so the loop ends; 4 - 11 rounds on dense 2000-feature frames (signing ALL candidates of ALL unsettled claimers
// instead needed 15 - 25).  Availability is "claimed by a settled query with a SMALLER index": the result is the sequential one exactly.
// ============================================================================
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/orbslam_hip.h"
#include "common.h"
#include "orb_frame.h"

namespace orbhip {

#define TRK_TH_HIGH 100
#define TRK_HISTO 30
#define TRK_MAXKP 4096                                       // keypoints per frame the single-workgroup kernels hold in LDS
#define TRK_NCELL (FRAME_GRID_COLS * FRAME_GRID_ROWS)
#define TRK_LMAX 16                                          // acceptable candidates per query held in registers
#define TRK_GT 1024                                          // threads of the greedy workgroup
#define TRK_QPT (TRK_MAXKP / TRK_GT)                         // queries per thread

struct TrkIn {                                               // the packed constant part of the upload
  double R[9], t[3];                                         // predicted Tcw (velocity * last pose)
  double pose7[7];                                           // the same as [t, q] for the pose optimisation
  float K4[4], bounds[4], th;
  float scale[16], inv_sigma2[16];
  int nq, check_ori, nlevels, pad;
};

// ---- one workgroup: queries, the frame's float records, its grid ------------------------------------------------------
__global__ __launch_bounds__(1024) void k_trk_prepare(const TrkIn* __restrict__ in, const double* __restrict__ last_Xw,
                                                      const int32_t* __restrict__ last_octave, const uint8_t* __restrict__ last_valid,
                                                      const orbx_keypoint* __restrict__ kps, const int32_t* __restrict__ d_count, int cap,
                                                      float* __restrict__ q_uv, float* __restrict__ q_radius, int32_t* __restrict__ q_lo,
                                                      int32_t* __restrict__ q_hi, uint8_t* __restrict__ q_valid, float* __restrict__ kps4,
                                                      uint32_t* __restrict__ cell_off, uint32_t* __restrict__ cell_idx, const int nq, uint32_t* __restrict__ list_total) {
  __shared__ int s_cnt[TRK_NCELL];
  __shared__ int s_off[TRK_NCELL + 1];
  __shared__ unsigned short s_cell[TRK_MAXKP];
  __shared__ int s_w[16];
  __shared__ unsigned short s_idx[TRK_MAXKP];
  const int tid = threadIdx.x;
  if (tid == 0) *list_total = 0u;                               // (k_trk_windows' allocation counter)
  const TrkIn I = *in;
  const int n = min(max(*d_count, 0), min(cap, TRK_MAXKP));
  // every global value is requested up front (a lone workgroup pays each dependent round trip in full): the last frame's points
  // and the frame's keypoints, up to four of each per thread
  constexpr int PT = TRK_MAXKP / 1024;
  uint8_t lv[PT]; double lx[PT], ly[PT], lz[PT]; int lo[PT]; orbx_keypoint kp[PT];
#pragma unroll
  for (int k = 0; k < PT; k++) {
    const int i = tid + 1024 * k;
    const bool qi = i < nq;
    lv[k] = qi ? last_valid[i] : (uint8_t)0;
    lx[k] = qi ? last_Xw[3 * i] : 0.0; ly[k] = qi ? last_Xw[3 * i + 1] : 0.0; lz[k] = qi ? last_Xw[3 * i + 2] : 1.0;
    lo[k] = qi ? last_octave[i] : 0;
    if (i < n) kp[k] = kps[i];
  }
  for (int c = tid; c < TRK_NCELL; c += 1024) s_cnt[c] = 0;
  // projection of the last frame's map points (src/ORBmatcher.cc:1185-1212): double camera coordinates, float from there on
#pragma unroll
  for (int k = 0; k < PT; k++) {
    const int i = tid + 1024 * k;
    if (i >= nq) continue;
    uint8_t v = lv[k];
    float u = 0.f, vv = 0.f, rad = 0.f; int oc = 0;
    if (v) {
      const double X = lx[k], Y = ly[k], Z = lz[k];
      const double cx3 = (I.R[0] * X + I.R[1] * Y + I.R[2] * Z) + I.t[0], cy3 = (I.R[3] * X + I.R[4] * Y + I.R[5] * Z) + I.t[1],
                   cz3 = (I.R[6] * X + I.R[7] * Y + I.R[8] * Z) + I.t[2];
      const float xc = (float)cx3, yc = (float)cy3;
      const float invzc = (float)(1.0 / cz3);
      if (invzc < 0) v = 0;
      u = I.K4[0] * xc * invzc + I.K4[2];
      vv = I.K4[1] * yc * invzc + I.K4[3];
      if (u < I.bounds[0] || u > I.bounds[1] || vv < I.bounds[2] || vv > I.bounds[3]) v = 0;
      oc = lo[k];
      if (oc < 0 || oc >= I.nlevels) v = 0; else rad = I.th * in->scale[oc];      // (in->: a dynamic index into the private copy would put the whole struct into scratch)
    }
    q_uv[2 * i] = u; q_uv[2 * i + 1] = vv; q_radius[i] = rad; q_lo[i] = oc - 1; q_hi[i] = oc + 1; q_valid[i] = v;
  }
  // the frame: {x, y, octave, angle} of the (undistorted == raw: zero distortion) keypoints and their grid cells (src/Frame.cc:158-173, :309-320)
  const float winv = (float)FRAME_GRID_COLS / (I.bounds[1] - I.bounds[0]), hinv = (float)FRAME_GRID_ROWS / (I.bounds[3] - I.bounds[2]);
  __syncthreads();
#pragma unroll
  for (int kk = 0; kk < PT; kk++) {
    const int i = tid + 1024 * kk;
    if (i >= n) continue;
    const orbx_keypoint k = kp[kk];
    *(float4*)&kps4[4 * i] = make_float4(k.x, k.y, (float)k.octave, k.angle);
    const int px = (int)roundf((k.x - I.bounds[0]) * winv), py = (int)roundf((k.y - I.bounds[2]) * hinv);
    unsigned short id = 0xFFFF;
    if (!(px < 0 || px >= FRAME_GRID_COLS || py < 0 || py >= FRAME_GRID_ROWS)) { id = (unsigned short)(px * FRAME_GRID_ROWS + py); atomicAdd(&s_cnt[id], 1); }
    s_cell[i] = id;
  }
  __syncthreads();
  // exclusive scan of the 3072 cell counts: three per thread, wave scan, wave totals through LDS
  {
    const int c0 = 3 * tid;
    const int a0 = s_cnt[c0], a1 = s_cnt[c0 + 1], a2 = s_cnt[c0 + 2], mine = a0 + a1 + a2;
    int inc = mine;
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < w; k++) base += s_w[k];
    const int ex = base + inc - mine;
    s_off[c0] = ex; s_off[c0 + 1] = ex + a0; s_off[c0 + 2] = ex + a0 + a1;
    if (tid == 1023) s_off[TRK_NCELL] = ex + mine;
  }
  __syncthreads();
  for (int c = tid; c <= TRK_NCELL; c += 1024) cell_off[c] = (uint32_t)s_off[c];
  for (int c = tid; c < TRK_NCELL; c += 1024) s_cnt[c] = 0;
  __syncthreads();
  // fill: slots by atomics, then every (tiny) list sorted by keypoint index = the reference's push_back order - in LDS, written
  // out in one coalesced pass
  for (int i = tid; i < n; i += 1024) {
    const unsigned short id = s_cell[i];
    if (id != 0xFFFF) s_idx[s_off[id] + atomicAdd(&s_cnt[id], 1)] = (unsigned short)i;
  }
  __syncthreads();
  for (int c = tid; c < TRK_NCELL; c += 1024) {
    const int b = s_off[c], e = s_off[c + 1];
    for (int i = b + 1; i < e; i++) {
      const unsigned short v = s_idx[i];
      int j = i - 1;
      while (j >= b && s_idx[j] > v) { s_idx[j + 1] = s_idx[j]; j--; }
      s_idx[j + 1] = v;
    }
  }
  __syncthreads();
  const int total = s_off[TRK_NCELL];
  for (int i = tid; i < total; i += 1024) cell_idx[i] = (uint32_t)s_idx[i];
}

// ---- the window search and the distances in ONE launch (three + a scan before), one WAVE per query: Frame::GetFeaturesInArea's
// cells in the reference's loop order (ix outer, iy inner), 64 cells at a time, every lane walks the (tiny) list of its cell;
// pass 1 counts the hits and the ACCEPTABLE ones (distance <= TH_HIGH: the only ones that can ever be chosen or block anybody; a
// window holds few of them - the true match and the odd look-alike, random descriptors are 128 +- 8 bits apart), pass 2 writes
// the acceptable ones compacted in list order: acc[q][r] = target | distance << 16 for the first TRK_LMAX of them, acc_n[q] = how
// many there are.  The full list {target, distance} is written only for the rare query with more than TRK_LMAX acceptable
// candidates (the greedy pass walks it then): its place in `pairs` comes from an atomic counter - the lists of different queries
// need no order among each other.
__global__ __launch_bounds__(256) void k_trk_windows(const float* __restrict__ kps4, const uint32_t* __restrict__ cell_off, const uint32_t* __restrict__ cell_idx,
                                                     float min_x, float min_y, float winv, float hinv, const float* __restrict__ q_xy, const float* __restrict__ q_r,
                                                     const int* __restrict__ q_minl, const int* __restrict__ q_maxl, const uint8_t* __restrict__ q_valid, int nq,
                                                     const uint8_t* __restrict__ q_desc, const uint8_t* __restrict__ t_desc, uint32_t* __restrict__ total,
                                                     uint32_t* __restrict__ off, uint2* __restrict__ pairs, uint32_t cap, uint32_t* __restrict__ acc,
                                                     int32_t* __restrict__ acc_n, const int dmax) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= nq) return;
  uint32_t base = 0, nbig = 0;
  int found = 0;
  if (q_valid[q]) {
    const float x = q_xy[2 * q], y = q_xy[2 * q + 1], r = q_r[q];
    const int minLevel = q_minl[q], maxLevel = q_maxl[q];
    const uint4* a = (const uint4*)(q_desc + 32 * (size_t)q);
    const uint4 a0 = a[0], a1 = a[1];
    const int min_cx = max(0, (int)floorf((x - min_x - r) * winv));
    const int max_cx = min(FRAME_GRID_COLS - 1, (int)ceilf((x - min_x + r) * winv));
    const int min_cy = max(0, (int)floorf((y - min_y - r) * hinv));
    const int max_cy = min(FRAME_GRID_ROWS - 1, (int)ceilf((y - min_y + r) * hinv));
    if (!(min_cx >= FRAME_GRID_COLS || max_cx < 0 || min_cy >= FRAME_GRID_ROWS || max_cy < 0) && max_cx >= min_cx && max_cy >= min_cy) {
      const bool check = (minLevel > 0) || (maxLevel >= 0);
      const int ny = max_cy - min_cy + 1, ncell = (max_cx - min_cx + 1) * ny;
      auto hit = [&](uint32_t j) {
        const float4 k = ((const float4*)kps4)[j];
        const int oct = (int)k.z;
        if (check) {
          if (oct < minLevel) return false;
          if (maxLevel >= 0 && oct > maxLevel) return false;
        }
        return fabsf(k.x - x) < r && fabsf(k.y - y) < r;
      };
      auto dist = [&](uint32_t j) {
        const uint4* tb = (const uint4*)(t_desc + 32 * (size_t)j);
        const uint4 b0 = tb[0], b1 = tb[1];
        return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) +
               __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
      };
      auto cell_range = [&](int w, uint32_t& lo, uint32_t& hi) {
        lo = 0; hi = 0;
        if (w < ncell) { const int ix = min_cx + w / ny, iy = min_cy + w % ny; const int c = ix * FRAME_GRID_ROWS + iy; lo = cell_off[c]; hi = cell_off[c + 1]; }
      };
      int nhit = 0, nacc = 0;
      for (int w0 = 0; w0 < ncell; w0 += 64) {
        uint32_t lo, hi; cell_range(w0 + lane, lo, hi);
        for (uint32_t e = lo; e < hi; e++) { const uint32_t j = cell_idx[e]; if (hit(j)) { nhit++; nacc += dist(j) <= dmax ? 1 : 0; } }
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) { nhit += __shfl_xor(nhit, o); nacc += __shfl_xor(nacc, o); }
      const bool big = nacc > TRK_LMAX;
      if (big) {
        if (lane == 0) base = atomicAdd(total, (uint32_t)nhit);
        base = __shfl(base, 0); nbig = (uint32_t)nhit;
      }
      int tot = 0;
      for (int w0 = 0; w0 < ncell; w0 += 64) {
        uint32_t lo, hi; cell_range(w0 + lane, lo, hi);
        int mine = 0, amine = 0;
        for (uint32_t e = lo; e < hi; e++) { const uint32_t j = cell_idx[e]; if (hit(j)) { mine++; amine += dist(j) <= dmax ? 1 : 0; } }
        int incl = mine, aincl = amine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o), ta = __shfl_up(aincl, o); if (lane >= o) { incl += t; aincl += ta; } }
        if (amine > 0 || (big && mine > 0)) {
          uint32_t pos = base + (uint32_t)(tot + incl - mine);
          int ar = found + aincl - amine;
          for (uint32_t e = lo; e < hi; e++) {
            const uint32_t j = cell_idx[e];
            if (!hit(j)) continue;
            const int d = dist(j);
            if (big) { if (pos < cap) pairs[pos] = make_uint2(j, (uint32_t)d); pos++; }
            if (d <= dmax) { if (ar < TRK_LMAX) acc[(size_t)q * TRK_LMAX + ar] = j | ((uint32_t)d << 16); ar++; }
          }
        }
        tot += __shfl(incl, 63); found += __shfl(aincl, 63);
      }
    }
  }
  if (lane == 0) { acc_n[q] = found; off[2 * q] = base; off[2 * q + 1] = base + nbig; }
}

struct TrkOut { int32_t n_keypoints, nmatches, nobs, rounds, cand_total, n_inliers, pad0, pad1; int32_t ticks[8]; };

#ifdef ORBHIP_TRK_PROF
// settles per microsecond since the dataflow loop started [0..63], polls of the slowest thread [64], queries left after pass one [65],
// longest chain of waits in links [66], sum of the chain lengths [67]
__device__ int g_trk_prof[72];
#endif
// ---- one workgroup: the order-dependent pass, rotation consistency, slot owners, PoseOptimization's observation list ----
__global__ __launch_bounds__(TRK_GT) void k_trk_greedy(const TrkIn* __restrict__ in, const uint8_t* __restrict__ q_valid, const float* __restrict__ q_angle,
                                                     const uint32_t* __restrict__ acc, const int32_t* __restrict__ acc_n,
                                                     const double* __restrict__ last_Xw, const uint32_t* __restrict__ off, const uint32_t* __restrict__ off_total_p, const uint2* __restrict__ pairs,
                                                     uint32_t cand_cap, const float* __restrict__ kps4, const int32_t* __restrict__ d_count, int cap,
                                                     int32_t* __restrict__ match, int32_t* __restrict__ owner, int32_t* __restrict__ obs_feat,
                                                     double* __restrict__ obs_Xw, double* __restrict__ obs_uv, float* __restrict__ obs_w,
                                                     int32_t* __restrict__ obs_off, double* __restrict__ pose7, double* __restrict__ K4d, TrkOut* __restrict__ out, int ecap,
                                                     int tcap, int force_rounds, const int nq, const int check_ori) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
  __shared__ int s_taken[TRK_MAXKP];                          // index of the settled claiming query that holds the target (INT_MAX: free)
  __shared__ int s_mark[TRK_MAXKP];                           // smallest index of a claimer that stays unsettled and could still take the target
  __shared__ int s_sign[TRK_MAXKP];                           // smallest index of an unsettled claimer that PROPOSES the target
  __shared__ int s_flag[3];
  __shared__ int s_hist[TRK_HISTO], s_keep[TRK_HISTO];
  __shared__ int s_left, s_nm, s_w[16], s_big;
  const unsigned long long tkb = __builtin_amdgcn_s_memrealtime();
  const int tid = threadIdx.x;
  // (nq and check_ori are arguments: nothing the setup loads depends on another load - a lone workgroup right behind a launch pays
  // ~3 us per dependent global round trip)
  const uint32_t off_total = *off_total_p;                      // (entries of the full lists: only queries with more than TRK_LMAX acceptable candidates have one)
  const int n_dev = *d_count;
  const bool overflow = off_total > cand_cap;                   // (the caller re-runs with a larger candidate buffer; first USED behind the batch of loads below)
  const int n = min(max(n_dev, 0), min(cap, TRK_MAXKP));
  for (int t = tid; t < TRK_MAXKP; t += TRK_GT) { s_taken[t] = INT_MAX; s_mark[t] = INT_MAX; s_sign[t] = INT_MAX; }
  if (tid < 3) s_flag[tid] = 0;
  if (tid < TRK_HISTO) s_hist[tid] = 0;
  if (tid == 0) { s_left = 0; s_nm = 0; s_big = 0; }
  __syncthreads();
  if (tid == 0) out->ticks[6] = (int)(__builtin_amdgcn_s_memrealtime() - tkb);
  // Only candidates within TH_HIGH can ever be chosen or block anybody, and a window holds few of those (the true match and
  // the odd look-alike; random descriptors are 128 +- 8 bits apart; k_trk_windows compacted them in list order): they go into LDS
  // as CSR lists (a query whose list does not fit any more reads its <= TRK_LMAX entries from global memory; one with more
  // than TRK_LMAX acceptable candidates walks its full list), and the rounds below run over a WORK LIST of the unsettled
  // queries that is compacted after every round - after the first round a few hundred of the ~2000 queries are left, so a
  // phase is a handful of instructions for a fraction of the threads instead of fully unrolled, predicated per-thread lists
  // for all of them (160 -> ~45 us on a dense frame).
  int* s_coff = (int*)s_dyn;                                   // [nq + 1]
  int* s_st = s_coff + (nq + 1);                               // [nq] -3 unsettled, -1 settled without a match, >= 0 the matched target
  int* s_prop = s_st + nq;                                     // [nq] proposal | nonfinal << 16
  unsigned short* s_wl[2];
  s_wl[0] = (unsigned short*)(s_prop + nq); s_wl[1] = s_wl[0] + ((nq + 1) & ~1);
  uint32_t* s_ent = (uint32_t*)(s_wl[1] + ((nq + 1) & ~1));    // [ecap]
  // per-query counts -> offsets (block scan over <= 4 queries per thread), work list of everything that has a candidate
  {
    // (every global value this block needs is requested HERE, in one go: the counts, the flags and the first four entries of each
    // list - the loads used to be issued one by one behind the scan, a dozen dependent round trips of a lone workgroup)
    int c4[TRK_QPT], an[TRK_QPT], qv[TRK_QPT], mine = 0;
    uint4 e4[TRK_QPT];
#pragma unroll
    for (int k = 0; k < TRK_QPT; k++) {
      const int q = TRK_QPT * tid + k;
      an[k] = q < nq ? acc_n[q] : 0;
      qv[k] = q < nq ? (int)q_valid[q] : 0;
      e4[k] = q < nq ? *(const uint4*)(acc + (size_t)q * TRK_LMAX) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < TRK_QPT; k++) {
      c4[k] = (qv[k] && !overflow) ? min(an[k], TRK_LMAX) : 0;
      mine += c4[k];
    }
    int inc = mine;
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    if (tid == 0) out->ticks[7] = (int)(__builtin_amdgcn_s_memrealtime() - tkb);
    int base = 0;
    for (int k = 0; k < w; k++) base += s_w[k];
    int at = base + inc - mine;
#pragma unroll
    for (int k = 0; k < TRK_QPT; k++) {
      const int q = TRK_QPT * tid + k;
      const bool live = q < nq && qv[k] && !overflow && an[k] > 0;
      {                                                         // work list slot: one LDS atomic per wave (a same-address atomic per LANE serialises)
        const unsigned long long m = __builtin_amdgcn_ballot_w64(live);
        int wbase = 0;
        if (lane == 0 && m) wbase = atomicAdd(&s_left, __popcll(m));
        wbase = __shfl(wbase, 0);
        if (live) s_wl[0][wbase + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)q;
      }
      if (q < nq) {
        // (bit 30: the query has more than TRK_LMAX acceptable candidates; bit 29: its map point has observations, it CLAIMS its
        // target - everything a phase needs to know about a query then comes from LDS)
        s_coff[q] = at | ((an[k] > TRK_LMAX) ? (1 << 30) : 0) | ((qv[k] == 1) ? (1 << 29) : 0);
        if (live && an[k] > TRK_LMAX) s_big = 1;
        s_st[q] = live ? -3 : -1;
        if (at + c4[k] <= ecap) {
          if (c4[k] > 0) s_ent[at] = e4[k].x;
          if (c4[k] > 1) s_ent[at + 1] = e4[k].y;
          if (c4[k] > 2) s_ent[at + 2] = e4[k].z;
          if (c4[k] > 3) s_ent[at + 3] = e4[k].w;
          for (int e = 4; e < c4[k]; e++) s_ent[at + e] = acc[(size_t)q * TRK_LMAX + e];
        }
        at += c4[k];
      }
      if (q == nq - 1) s_coff[nq] = at;                         // (s_coff[q + 1] - s_coff[q], masked, is the list length)
    }
    if (nq == 0 && tid == 0) s_coff[0] = 0;
  }
  __syncthreads();
  int U = s_left;
  __syncthreads();
  if (tid == 0) s_left = 0;
  // visit the acceptable, still available candidates of query q in list order: fn(target, distance)
  constexpr int OFFM = (1 << 29) - 1;
  auto claims = [&](int q) { return (s_coff[q] >> 29) & 1; };
  auto for_candidates = [&](int q, auto fn) {
    const int cq = s_coff[q];
    if (cq & (1 << 30)) {                                        // (rare) the full window list
      for (uint32_t c = off[2 * q]; c < off[2 * q + 1]; c++) { const int t = (int)pairs[c].x, d = (int)pairs[c].y; if (d <= TRK_TH_HIGH && s_taken[t] >= q) fn(t, d); }
      return;
    }
    const int b = cq & OFFM, cnt = (s_coff[q + 1] & OFFM) - b;
    const bool in_lds = b + cnt <= ecap;
    for (int e = 0; e < cnt; e++) {
      const uint32_t v = in_lds ? s_ent[b + e] : acc[(size_t)q * TRK_LMAX + e];
      const int t = (int)(v & 0xFFFFu);
      if (s_taken[t] >= q) fn(t, (int)(v >> 16));               // (:1220-1221: a feature that holds an observed point is skipped)
    }
  };
  const unsigned long long tk0 = __builtin_amdgcn_s_memrealtime();
  int rounds = 0, inner_total = 0, phase = 0, cur = 0;         // phase: counter of the rotating "anything changed?" flags
  // ---- the common case as DATAFLOW, no rounds: what query q gets depends only on EARLIER claimers (a later query cannot change it,
  // and a query without observations closes nothing), and only on those that list q's best candidate among the ones nobody holds
  // yet.  With the lists inverted (target -> the claimers that list it, in LDS) every thread polls exactly those for its queries
  // and decides the moment they are all settled - the smallest unsettled query is always ready, so this terminates; the depth is
  // the rounds' (~10 on a dense frame) times a few LDS round trips instead of ~10 workgroup barriers per round.
  // Needs every list in LDS; otherwise (or ORBHIP_TRACK_ROUNDS=1) the rounds below - both are the exact sequential result.
  const int total_ent = s_coff[nq] & OFFM;
  const bool fast = !force_rounds && !overflow && !s_big && total_ent <= ecap && total_ent <= tcap && U > 0;
  if (fast) {
    unsigned short* s_tl = (unsigned short*)(s_ent + ecap);     // [tcap] claimers per target
    int* s_tcnt = s_sign; int* s_toff = s_mark;                 // (free on this path) per target: number of claimers, first entry
    for (int t = tid; t < TRK_MAXKP; t += TRK_GT) s_tcnt[t] = 0;
    __syncthreads();
    for (int k = 0; k < TRK_QPT; k++) {
      const int q = tid + TRK_GT * k;
      if (q >= nq || s_st[q] != -3 || !claims(q)) continue;
      const int b = s_coff[q] & OFFM, cnt = (s_coff[q + 1] & OFFM) - b;
      for (int e = 0; e < cnt; e++) atomicAdd(&s_tcnt[s_ent[b + e] & 0xFFFFu], 1);
    }
    __syncthreads();
    {                                                           // exclusive scan of the 4096 counts
      const int c0 = TRK_QPT * tid;
      int v[TRK_QPT], mine = 0;
#pragma unroll
      for (int k = 0; k < TRK_QPT; k++) { v[k] = s_tcnt[c0 + k]; mine += v[k]; }
      int inc = mine;
      const int lane = tid & 63, w = tid >> 6;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
      if (lane == 63) s_w[w] = inc;
      __syncthreads();
      int at = inc - mine;
      for (int k = 0; k < w; k++) at += s_w[k];
#pragma unroll
      for (int k = 0; k < TRK_QPT; k++) { s_toff[c0 + k] = at; at += v[k]; s_tcnt[c0 + k] = 0; }
    }
    __syncthreads();
    for (int k = 0; k < TRK_QPT; k++) {
      const int q = tid + TRK_GT * k;
      if (q >= nq || s_st[q] != -3 || !claims(q)) continue;
      const int b = s_coff[q] & OFFM, cnt = (s_coff[q + 1] & OFFM) - b;
      for (int e = 0; e < cnt; e++) { const int t = (int)(s_ent[b + e] & 0xFFFFu); s_tl[s_toff[t] + atomicAdd(&s_tcnt[t], 1)] = (unsigned short)q; }
    }
    __syncthreads();
    auto ld = [](const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    // a query's state is published with RELEASE behind its claim and read with ACQUIRE in front of the re-read of s_taken
    // (LDS, workgroup scope: an lgkmcnt wait, no cache maintenance)
    auto lda = [](const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); };
    if (tid == 0) out->ticks[4] = (int)(__builtin_amdgcn_s_memrealtime() - tk0);      // inverted lists built
    bool done[TRK_QPT];
    int left = 0, blk[TRK_QPT];                                  // blk: the unsettled predecessor the query was last seen waiting for
#pragma unroll
    for (int k = 0; k < TRK_QPT; k++) { const int q = tid + TRK_GT * k; done[k] = !(q < nq && s_st[q] == -3); left += done[k] ? 0 : 1; blk[k] = -1; }
#ifdef ORBHIP_TRK_PROF
    int lastblk[TRK_QPT] = {-1, -1, -1, -1};
#endif
    for (int it = 0; it < (1 << 20) && left > 0; it++) {
#pragma unroll
      for (int k = 0; k < TRK_QPT; k++) {
        if (done[k]) continue;
        if (blk[k] >= 0 && lda(&s_st[blk[k]]) == -3) continue;     // still waiting for the same query: one read per poll
#ifdef ORBHIP_TRK_PROF
        if (blk[k] >= 0) lastblk[k] = blk[k];
#endif
        const int q = tid + TRK_GT * k;
        const int b = s_coff[q] & OFFM, cnt = (s_coff[q + 1] & OFFM) - b;
        // the best candidate no earlier claimer HOLDS (held = by a settled query: final); q takes it as soon as no earlier claimer
        // that lists it is still unsettled - what happens to q's other candidates cannot change that choice
        // (the first four entries / claimers are read side by side - independent LDS reads instead of a dependent chain per entry:
        // the latency of this block is the latency of a link of the dependency chain)
        int best = 256, bi = -1;
        {
          uint32_t v4[4]; int tk[4];
#pragma unroll
          for (int e = 0; e < 4; e++) v4[e] = e < cnt ? s_ent[b + e] : 0u;
#pragma unroll
          for (int e = 0; e < 4; e++) tk[e] = e < cnt ? ld(&s_taken[v4[e] & 0xFFFFu]) : -1;
#pragma unroll
          for (int e = 0; e < 4; e++) { const int d = (int)(v4[e] >> 16); if (e < cnt && tk[e] >= q && d < best) { best = d; bi = (int)(v4[e] & 0xFFFFu); } }
        }
        for (int e = 4; e < cnt; e++) {
          const uint32_t v = s_ent[b + e];
          const int t = (int)(v & 0xFFFFu), d = (int)(v >> 16);
          if (ld(&s_taken[t]) >= q && d < best) { best = d; bi = t; }      // (taken by a LATER query: still free for q, :1220-1221)
        }
        bool ready = true;
        if (bi >= 0) {
          const int tb = s_toff[bi], tn = s_tcnt[bi];
          {
            int qc4[4], st4[4];
#pragma unroll
            for (int c = 0; c < 4; c++) qc4[c] = c < tn ? (int)s_tl[tb + c] : INT_MAX;
#pragma unroll
            for (int c = 0; c < 4; c++) st4[c] = qc4[c] < q ? lda(&s_st[qc4[c]]) : 0;
#pragma unroll
            for (int c = 3; c >= 0; c--) if (qc4[c] < q && st4[c] == -3) { ready = false; blk[k] = qc4[c]; }
          }
          for (int c = 4; c < tn && ready; c++) { const int qc = s_tl[tb + c]; if (qc < q && lda(&s_st[qc]) == -3) { ready = false; blk[k] = qc; } }
          // (a claimer that settled between the two loops may have taken bi: look again)
          if (ready && ld(&s_taken[bi]) < q) { ready = false; blk[k] = -1; }
        }
        if (!ready) continue;
        if (bi >= 0 && claims(q)) {
          // an earlier claimer cannot hold bi (tested); a later one that took it while q, without observations ... cannot be: q claims
          __hip_atomic_fetch_min(&s_taken[bi], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __hip_atomic_store(&s_st[q], bi >= 0 ? bi : -1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        done[k] = true; left--;
#ifdef ORBHIP_TRK_PROF
        { const int dep = (lastblk[k] >= 0 ? s_prop[lastblk[k]] : 0) + 1; s_prop[q] = dep; atomicMax(&g_trk_prof[66], dep); atomicAdd(&g_trk_prof[67], dep); }
        atomicAdd(&g_trk_prof[min(63, (int)((__builtin_amdgcn_s_memrealtime() - tk0) / 100))], 1);
#endif
      }
      inner_total++;
#ifdef ORBHIP_TRK_PROF
      if (it == 0 && left > 0) atomicAdd(&g_trk_prof[65], left);
      if (left == 0) atomicMax(&g_trk_prof[64], it);
#endif
      if (it == 0 && tid == 0) out->ticks[5] = (int)(__builtin_amdgcn_s_memrealtime() - tk0);      // first pass over all queries
    }
    rounds = 1;
    __syncthreads();
    U = 0;
  }
  for (; rounds < 4096 && U > 0; rounds++) {
    const unsigned short* wl = s_wl[cur];
    // (1) proposals: the best still-available candidate of every unsettled query; claimers sign their PROPOSAL
    for (int i = tid; i < U; i += TRK_GT) {
      const int q = wl[i];
      int best = 256, bi = -1;
      for_candidates(q, [&](int t, int d) { if (d < best) { best = d; bi = t; } });
      if (bi < 0) { s_st[q] = -1; continue; }                  // can only get worse: settled, nothing matched
      s_prop[q] = bi;
      if (claims(q)) atomicMin(&s_sign[bi], q);
    }
    __syncthreads();
    // (2) + (3) the claimers that stay unsettled in this round: the ones that are not the first to want their target, and -
    // greatest fixpoint, from those losers outwards - every winner whose target an EARLIER unsettled claimer has on its list.
    // A claimer that joins the set signs, at once, every target it could still take (s_mark); ONE barrier per sweep, the
    // "anything changed?" flags rotate so that none has to be cleared between two barriers that read it.
    for (int sweep = 0; sweep < 8192; sweep++) {
      inner_total++;
      bool changed = false;
      for (int i = tid; i < U; i += TRK_GT) {
        const int q = wl[i];
        const int pr = s_prop[q];
        if (s_st[q] != -3 || (pr >> 16) || !claims(q)) continue;
        if (sweep == 0 ? (s_sign[pr] != q) : (s_mark[pr] < q)) {
          s_prop[q] = pr | (1 << 16); changed = true;
          for_candidates(q, [&](int t, int) { atomicMin(&s_mark[t], q); });
        }
      }
      if (changed) s_flag[phase % 3] = 1;
      if (tid == 0) s_flag[(phase + 1) % 3] = 0;
      __syncthreads();
      const int any = s_flag[phase % 3];
      phase++;
      if (!any) break;
    }
    // (4) settle: the final claimers take their targets ...
    for (int i = tid; i < U; i += TRK_GT) {
      const int q = wl[i];
      const int pr = s_prop[q];
      if (s_st[q] == -3 && claims(q) && !(pr >> 16)) { s_st[q] = pr; s_taken[pr] = q; }
    }
    __syncthreads();
    // ... a query without observations keeps its proposal when no earlier claimer took it in this round and none that stays
    // unsettled could; what is left goes onto the next work list
    unsigned short* wn = s_wl[cur ^ 1];
    for (int i = tid; i < U; i += TRK_GT) {
      const int q = wl[i];
      if (s_st[q] != -3) continue;
      const int pr = s_prop[q] & 0xFFFF;
      if (!claims(q) && s_mark[pr] > q && s_taken[pr] > q) s_st[q] = pr;
      else wn[atomicAdd(&s_left, 1)] = (unsigned short)q;
    }
    __syncthreads();
    U = s_left; cur ^= 1;
    for (int t = tid; t < TRK_MAXKP; t += TRK_GT) { s_mark[t] = INT_MAX; s_sign[t] = INT_MAX; }
    __syncthreads();
    if (tid == 0) s_left = 0;
  }
  __syncthreads();
  // back to the queries this thread owns for the passes below
  int st[TRK_QPT];
#pragma unroll
  for (int k = 0; k < TRK_QPT; k++) { const int q = tid + TRK_GT * k; st[k] = q < nq ? s_st[q] : -1; }
  const unsigned long long tk1 = __builtin_amdgcn_s_memrealtime();
  // rotation consistency (src/ORBmatcher.cc:1235-1264): histogram of the matches, the three fullest bins survive
  int bin[TRK_QPT];
#pragma unroll
  for (int k = 0; k < TRK_QPT; k++) bin[k] = -1;
  for (int k = 0; k < TRK_QPT; k++) {
    const int q = tid + TRK_GT * k;
    if (q >= nq || st[k] < 0) continue;
    atomicAdd(&s_nm, 1);
    if (check_ori) {
      float rot = q_angle[q] - kps4[4 * st[k] + 3];
      if (rot < 0.0) rot += 360.0f;
      int b = (int)roundf(rot * (1.0f / TRK_HISTO));
      if (b == TRK_HISTO) b = 0;
      bin[k] = b; atomicAdd(&s_hist[b], 1);
    }
  }
  __syncthreads();
  if (tid == 0) {
    int top[3] = {-1, -1, -1}, pop[3] = {0, 0, 0};
    for (int b = 0; b < TRK_HISTO; b++) {                    // strict '>': the earlier bin wins ties (ComputeThreeMaxima :1386-1418)
      int r = 3;
      while (r > 0 && s_hist[b] > pop[r - 1]) r--;
      if (r == 3) continue;
      for (int m = 2; m > r; m--) { top[m] = top[m - 1]; pop[m] = pop[m - 1]; }
      top[r] = b; pop[r] = s_hist[b];
    }
    if ((float)pop[1] < 0.1f * (float)pop[0]) { top[1] = -1; top[2] = -1; }
    else if ((float)pop[2] < 0.1f * (float)pop[0]) { top[2] = -1; }
    for (int b = 0; b < TRK_HISTO; b++) s_keep[b] = (b == top[0] || b == top[1] || b == top[2]) ? 1 : 0;
  }
  // slot owners: the LAST query assigned to a feature holds it (:1232), a removed match empties the slot whoever else shares it (:1260-1264)
  int* s_owner = s_taken; int* s_dead = s_mark;
  __syncthreads();
  for (int t = tid; t < TRK_MAXKP; t += TRK_GT) { s_owner[t] = -1; s_dead[t] = 0; }
  __syncthreads();
  for (int k = 0; k < TRK_QPT; k++) {
    const int q = tid + TRK_GT * k;
    if (q >= nq) continue;
    int mres = st[k] >= 0 ? st[k] : -1;
    if (st[k] >= 0) {
      atomicMax(&s_owner[st[k]], q);
      if (check_ori && !s_keep[bin[k]]) { s_dead[st[k]] = 1; mres = -2 - st[k]; atomicAdd(&s_nm, -1); }
    }
    match[q] = mres;
  }
  __syncthreads();
  const unsigned long long tk2 = __builtin_amdgcn_s_memrealtime();
  // PoseOptimization's observations: the features that hold a point, in feature order (src/CeresOptimizer.cc:297-327)
  constexpr int FPT = TRK_MAXKP / TRK_GT;                    // features per thread
  int cntv = 0;
  for (int k = 0; k < FPT; k++) { const int t = FPT * tid + k; cntv += (t < n && s_owner[t] >= 0 && !s_dead[t]) ? 1 : 0; }
  int inc = cntv;
  {
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < w; k++) base += s_w[k];
    inc += base;
  }
  int pos = inc - cntv;
  for (int k = 0; k < FPT; k++) {
    const int t = FPT * tid + k;
    const bool has = t < n && s_owner[t] >= 0 && !s_dead[t];
    if (t < cap) owner[t] = has ? s_owner[t] : -1;
    if (!has) continue;
    const int q = s_owner[t];
    obs_feat[pos] = t;
    obs_Xw[3 * pos] = last_Xw[3 * q]; obs_Xw[3 * pos + 1] = last_Xw[3 * q + 1]; obs_Xw[3 * pos + 2] = last_Xw[3 * q + 2];
    obs_uv[2 * pos] = (double)kps4[4 * t]; obs_uv[2 * pos + 1] = (double)kps4[4 * t + 1];
    const int oc = (int)kps4[4 * t + 2];
    obs_w[pos] = in->inv_sigma2[oc];
    pos++;
  }
  if (tid == TRK_GT - 1) { obs_off[0] = 0; obs_off[1] = inc; out->nobs = inc; }
  if (tid < 7) pose7[tid] = in->pose7[tid];
  if (tid < 4) K4d[tid] = (double)in->K4[tid];
  if (tid == 0) { const unsigned long long tk3 = __builtin_amdgcn_s_memrealtime(); out->ticks[0] = (int)(tk0 - tkb); out->ticks[1] = (int)(tk1 - tk0); out->ticks[2] = (int)(tk2 - tk1); out->ticks[3] = (int)(tk3 - tk2); }
  if (tid == 0) { out->n_keypoints = n_dev; out->nmatches = overflow ? -1 : s_nm; out->rounds = rounds + 1000 * inner_total; out->cand_total = (int32_t)off_total; }
}

// ======================================================================================================================
// Tracking::TrackLocalMap's data-parallel core (src/Tracking.cc:673-750): SearchLocalPoints (:793-842) = Frame::isInFrustum
// (src/Frame.cc:191-241) over the local map points + ORBmatcher::SearchByProjection(Frame&, vpMapPoints, th)
// (src/ORBmatcher.cc:42-119) + CeresOptimizer::PoseOptimization, on the frame the motion-model step left on the device.
// ======================================================================================================================
struct TlmIn {
  FrustumCam C;
  double pose7[7];
  float K4[4], scale[16], inv_sigma2[16];
  float th, ratio; int n_mp, n_kp_host, dmax, pad;
};
#define TLM_MAXMP 16384

// one thread per local map point: the frustum test and what SearchByProjection derives from it (:55-68): window radius
// RadiusByViewingCos * th * scale_factors_[level], level range [level - 1, level]
__global__ __launch_bounds__(256) void k_tlm_frustum(const TlmIn* __restrict__ in, const double* __restrict__ P, const double* __restrict__ Pn,
                                                     const float* __restrict__ mind, const float* __restrict__ maxd, const uint8_t* __restrict__ state,
                                                     uint8_t* __restrict__ in_view, float* __restrict__ q_uv, float* __restrict__ q_r, int32_t* __restrict__ q_lo,
                                                     int32_t* __restrict__ q_hi, uint8_t* __restrict__ q_valid, uint32_t* __restrict__ list_total) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) *list_total = 0u;
  const int n = in->n_mp;
  if (i >= n) return;
  const uint8_t stt = state[i];
  float u = 0.f, v = 0.f, vc = 0.f, dist = 0.f; int lvl = 0;
  bool ok = false;
  if (stt) ok = frustum_eval(in->C, P + 3 * (size_t)i, Pn + 3 * (size_t)i, mind[i], maxd[i], 0, u, v, lvl, vc, dist);
  float r = vc > 0.998f ? 2.5f : 4.0f;                                      // RadiusByViewingCos (:121-126)
  const float th = in->th;
  if (th != 1.0f) r *= th;
  in_view[i] = ok ? 1 : 0;
  q_uv[2 * i] = u; q_uv[2 * i + 1] = v; q_r[i] = ok ? r * in->scale[lvl] : 0.f; q_lo[i] = lvl - 1; q_hi[i] = lvl; q_valid[i] = ok ? stt : (uint8_t)0;
}

// One workgroup: the order-dependent pass of SearchByProjection(Frame&, vpMapPoints, th) and PoseOptimization's observation list.
// The reference walks the map points in vector order; point q looks at the features of its window whose slot does not hold a point
// with observations (:83-84: the slots filled before the call AND the ones earlier points of this loop filled), takes the best
// and second best distance in candidate order, applies the ratio test when both are on the same pyramid level and writes itself
// into the best feature's slot - a later point may overwrite a slot whose holder has no observations.  Exact parallel evaluation:
// what q decides depends only on EARLIER points that share a candidate with it (best AND second best count), so in every round
// each feature records the smallest unsettled point that lists it, and a point is settled when it is that smallest point for every
// feature on its list (the earliest unsettled point always is: the loop ends; windows of th = 1 hold a handful of features, so the
// chains are short).  Only candidates that can matter are listed (k_trk_windows, dmax): a distance d can be the winner if
// d <= TH_HIGH and can veto through the ratio test only if ratio * d < TH_HIGH.
__global__ __launch_bounds__(1024) void k_tlm_greedy(const TlmIn* __restrict__ in, const uint8_t* __restrict__ q_valid, const uint32_t* __restrict__ acc,
                                                     const int32_t* __restrict__ acc_n, const uint32_t* __restrict__ off, const uint32_t* __restrict__ off_total_p,
                                                     const uint2* __restrict__ pairs, uint32_t cand_cap, const float* __restrict__ kps4, const int32_t* __restrict__ d_count,
                                                     int cap, const double* __restrict__ mp_Xw, const double* __restrict__ slot_Xw, const uint8_t* __restrict__ slot_state,
                                                     int32_t* __restrict__ mp_match, int32_t* __restrict__ slot_owner, int32_t* __restrict__ obs_feat,
                                                     double* __restrict__ obs_Xw, double* __restrict__ obs_uv, float* __restrict__ obs_w, int32_t* __restrict__ obs_off,
                                                     double* __restrict__ pose7, double* __restrict__ K4d, TrkOut* __restrict__ out, const int nq) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
  unsigned short* wl0 = (unsigned short*)s_dyn;                 // work lists of the unsettled points
  unsigned short* wl1 = wl0 + TLM_MAXMP;
  __shared__ int s_taken[TRK_MAXKP];                            // the point with observations that holds the feature's slot (-1: filled before the call; INT_MAX: open)
  __shared__ int s_minu[TRK_MAXKP];                             // smallest unsettled point that lists the feature (this round)
  __shared__ int s_owner[TRK_MAXKP];                            // the LAST point written into the slot (:110), -1 none
  __shared__ unsigned char s_oct[TRK_MAXKP];
  __shared__ int s_n[2], s_nm, s_w[16];
  const int tid = threadIdx.x;
  const uint32_t off_total = *off_total_p;
  const int n_dev = *d_count;
  const bool overflow = off_total > cand_cap;
  const int n = min(max(n_dev, 0), min(cap, TRK_MAXKP));
  const float ratio = in->ratio; const int dmax = in->dmax;
  for (int t = tid; t < TRK_MAXKP; t += 1024) {
    s_taken[t] = (t < n && slot_state[t] == 1) ? -1 : INT_MAX;
    s_owner[t] = -1;
    s_oct[t] = t < n ? (unsigned char)(int)kps4[4 * t + 2] : (unsigned char)0;
  }
  if (tid == 0) { s_n[0] = 0; s_n[1] = 0; s_nm = 0; }
  __syncthreads();
  // the work list: every point in view, in index order (the order inside the list does not matter)
  for (int q0 = 0; q0 < nq; q0 += 1024) {
    const int q = q0 + tid;
    const bool act = q < nq && q_valid[q] != 0 && !overflow;
    if (q < nq) mp_match[q] = -1;
    const unsigned long long m = __ballot(act);
    int base = 0;
    if ((tid & 63) == 0 && m) base = atomicAdd(&s_n[0], __popcll(m));
    base = __shfl(base, 0);
    if (act) wl0[base + __popcll(m & ((1ull << (tid & 63)) - 1ull))] = (unsigned short)q;
  }
  __syncthreads();
  // a point's candidate list: <= TRK_LMAX packed entries, or - rarely - its full list in `pairs`, filtered
  auto for_each = [&](int q, auto&& f) {
    const int na = acc_n[q];
    if (na <= TRK_LMAX) { for (int e = 0; e < na; e++) { const uint32_t v = acc[(size_t)q * TRK_LMAX + e]; f((int)(v & 0xFFFFu), (int)(v >> 16)); } }
    else { for (uint32_t e = off[2 * q]; e < off[2 * q + 1]; e++) { const uint2 pr = pairs[e]; if ((int)pr.y <= dmax) f((int)pr.x, (int)pr.y); } }
  };
  int rounds = 0, cur = 0;
  unsigned short* wl[2] = {wl0, wl1};
  while (true) {
    const int nw = s_n[cur];
    if (nw == 0) break;
    rounds++;
    for (int t = tid; t < TRK_MAXKP; t += 1024) s_minu[t] = INT_MAX;
    if (tid == 0) s_n[cur ^ 1] = 0;
    __syncthreads();
    for (int i = tid; i < nw; i += 1024) { const int q = wl[cur][i]; for_each(q, [&](int t, int) { atomicMin(&s_minu[t], q); }); }
    __syncthreads();
    for (int i0 = 0; i0 < nw; i0 += 1024) {
      const int i = i0 + tid;
      bool keep = false; int q = 0;
      if (i < nw) {
        q = wl[cur][i];
        bool first = true;
        for_each(q, [&](int t, int) { first = first && s_minu[t] == q; });
        if (!first) keep = true;
        else {
          int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
          for_each(q, [&](int t, int d) {
            if (s_taken[t] < q) return;                            // (:83-84; a later point cannot have settled before q: it shares this feature)
            const int lv = s_oct[t];
            if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestLevel2 = bestLevel; bestLevel = lv; bestIdx = t; }
            else if (d < bestDist2) { bestLevel2 = lv; bestDist2 = d; }
          });
          if (bestIdx >= 0 && bestDist <= TRK_TH_HIGH && !(bestLevel == bestLevel2 && (float)bestDist > ratio * (float)bestDist2)) {
            mp_match[q] = bestIdx;
            atomicMax(&s_owner[bestIdx], q);
            if (q_valid[q] == 1) s_taken[bestIdx] = q;               // (a point without observations does not close the slot)
            atomicAdd(&s_nm, 1);
          }
        }
      }
      const unsigned long long m = __ballot(keep);
      int base = 0;
      if ((tid & 63) == 0 && m) base = atomicAdd(&s_n[cur ^ 1], __popcll(m));
      base = __shfl(base, 0);
      if (keep) wl[cur ^ 1][base + __popcll(m & ((1ull << (tid & 63)) - 1ull))] = (unsigned short)q;
    }
    __syncthreads();
    cur ^= 1;
  }
  // PoseOptimization's observations: every slot that holds a point now - the ones filled before the call and not overwritten
  // (slot_Xw) and the new ones (mp_Xw) - in feature order (src/CeresOptimizer.cc:297-327)
  constexpr int FPT = TRK_MAXKP / 1024;
  auto has_pt = [&](int t) { return t < n && (s_owner[t] >= 0 || slot_state[t] != 0); };
  int cntv = 0;
  for (int k = 0; k < FPT; k++) cntv += has_pt(FPT * tid + k) ? 1 : 0;
  int inc = cntv;
  {
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < w; k++) base += s_w[k];
    inc += base;
  }
  int pos = inc - cntv;
  for (int k = 0; k < FPT; k++) {
    const int t = FPT * tid + k;
    if (t < cap) slot_owner[t] = t < n ? s_owner[t] : -1;
    if (!has_pt(t)) continue;
    const double* X = s_owner[t] >= 0 ? mp_Xw + 3 * (size_t)s_owner[t] : slot_Xw + 3 * (size_t)t;
    obs_feat[pos] = t;
    obs_Xw[3 * pos] = X[0]; obs_Xw[3 * pos + 1] = X[1]; obs_Xw[3 * pos + 2] = X[2];
    obs_uv[2 * pos] = (double)kps4[4 * t]; obs_uv[2 * pos + 1] = (double)kps4[4 * t + 1];
    obs_w[pos] = in->inv_sigma2[(int)kps4[4 * t + 2]];
    pos++;
  }
  if (tid == 1023) { obs_off[0] = 0; obs_off[1] = inc; out->nobs = inc; }
  if (tid < 7) pose7[tid] = in->pose7[tid];
  if (tid < 4) K4d[tid] = (double)in->K4[tid];
  if (tid == 0) { out->n_keypoints = n_dev; out->nmatches = overflow ? -1 : s_nm; out->rounds = rounds; out->cand_total = (int32_t)off_total; }
}

// ======================================================================================================================
// Tracking::TrackReferenceKeyFrame's data-parallel core (src/Tracking.cc:566-615): Frame::ComputeBoW (the vocabulary descent,
// orb_vocab.hip) + ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (src/ORBmatcher.cc:151-256) + CeresOptimizer::PoseOptimization.
// SearchByBoW is sequential only INSIDE a vocabulary node: the keyframe's features of a node, in list order, each take the closest
// still unmatched frame feature of the same node (best < ratio * second, best <= TH_LOW).  A feature belongs to one node, so
// the nodes are independent: one WAVE (= one workgroup) per node walks the keyframe's list while its lanes hold the node's frame features.
// ======================================================================================================================
#define TRF_TH_LOW 50
struct TrfIn { double pose7[7]; float K4[4], inv_sigma2[16]; float ratio; int check_ori, nn, n_kf; };

__global__ __launch_bounds__(256) void k_trf_init(int32_t* __restrict__ match_kf, int n_kf, int32_t* __restrict__ f_owner, int32_t* __restrict__ f_bin, int cap, int* __restrict__ hist) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_kf) match_kf[i] = -1;
  if (i < cap) { f_owner[i] = -1; f_bin[i] = 0; }
  if (i < TRK_HISTO + 2) hist[i] = 0;
}

__device__ __forceinline__ void two_smallest_merge(unsigned& k1, unsigned& k2, unsigned o1, unsigned o2) {     // {k1 <= k2} and {o1 <= o2} -> the two smallest of the four
  const unsigned a = min(k1, o1), b = max(k1, o1);
  k2 = min(b, min(k2, o2)); k1 = a;
}

__global__ __launch_bounds__(64) void k_trf_bow(const TrfIn* __restrict__ in, const uint32_t* __restrict__ fv_node, const uint32_t* __restrict__ fv_off,
                                                 const uint32_t* __restrict__ fv_idx, const uint8_t* __restrict__ kf_desc, const uint8_t* __restrict__ kf_valid,
                                                 const float* __restrict__ kf_angle, const uint8_t* __restrict__ f_desc, const uint32_t* __restrict__ f_node,
                                                 const double* __restrict__ f_wt, const float* __restrict__ kps4, const int32_t* __restrict__ d_count, int cap,
                                                 int32_t* __restrict__ match_kf, int32_t* __restrict__ f_owner, int32_t* __restrict__ f_bin, int* __restrict__ hist) {
  __shared__ unsigned short s_list[1][TRK_MAXKP];               // the node's frame features, ascending index (FeatureVector order); one wave per workgroup: [w] = [0]
  __shared__ unsigned char s_taken[1][TRK_MAXKP];
  __shared__ unsigned s_dj[64][64];                             // [entry of the slice][lane]: four distances, one byte each
  __shared__ int s_mf[64], s_mb[64];                            // per entry of the slice: matched frame feature (or -1), its rotation bin
  const int w = 0, lane = threadIdx.x & 63;
  const int m = blockIdx.x;
  if (m >= in->nn) return;
  const uint32_t node = fv_node[m];
  const int n = min(max(*d_count, 0), min(cap, TRK_MAXKP));
  const float ratio = in->ratio; const int check_ori = in->check_ori;
  // The node's frame features, ascending index: four 64-feature slices per trip, their loads requested together (a lone wave per
  // node paid 32 dependent trips over the 2000 features one slice at a time).
  int cnt = 0;
  for (int base = 0; base < n; base += 256) {
    uint32_t nd[4]; double wt[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int i = base + 64 * u + lane; nd[u] = i < n ? f_node[i] : 0xFFFFFFFFu; wt[u] = i < n ? f_wt[i] : 0.0; }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = base + 64 * u + lane;
      const bool hit = i < n && nd[u] == node && wt[u] > 0.0;          // (a stopped word - weight 0 - is not in the FeatureVector, TemplatedVocabulary.h:1156)
      const unsigned long long mk = __ballot(hit);
      if (hit) { const int p = cnt + __popcll(mk & ((1ull << lane) - 1ull)); s_list[w][p] = (unsigned short)i; s_taken[w][p] = 0; }
      cnt += __popcll(mk);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
  if (cnt == 0) return;
  // The node's first 256 frame descriptors stay in registers, four per lane (list positions lane, lane + 64, ...), with their angles
  // and their "taken" bits: a node of the ORB vocabulary holds ~20 features of a frame, but the distribution has a tail (183 in the
  // synthetic vocabulary of tools/track_latency.py), and the walk below is sequential - 0.3 us per keyframe entry when it touches
  // registers only, 2.2 us when every entry re-reads descriptors and flags from memory.
  uint4 c0[4], c1[4];
  float fa_l[4]; int fi_l[4];
  unsigned taken = 0;                                          // bit t: list position lane + 64 t is matched (vpMapPointMatches[realIdxF], :200)
#pragma unroll
  for (int t = 0; t < 4; t++) {
    c0[t] = make_uint4(0, 0, 0, 0); c1[t] = c0[t]; fa_l[t] = 0.f; fi_l[t] = 0;
    const int p = lane + 64 * t;
    if (p < cnt) { const int fi = s_list[w][p]; const uint4* tb = (const uint4*)(f_desc + 32 * (size_t)fi); c0[t] = tb[0]; c1[t] = tb[1]; fa_l[t] = kps4[4 * fi + 3]; fi_l[t] = fi; }
  }
  const uint32_t e_lo = fv_off[m], e_hi = fv_off[m + 1];
  auto dppu = [](unsigned v, auto ctrl) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, decltype(ctrl)::value, 0xF, 0xF, false); };
  // The keyframe's list entries of the node, 64 at a time: lane l fetches entry l - index, validity, descriptor, angle: one chain of
  // dependent loads for the whole slice - and the walk takes entry j's values from lane j (v_readlane with the scalar loop counter).
  for (uint32_t e0 = e_lo; e0 < e_hi; e0 += 64) {
    const int ne = (int)min(64u, e_hi - e0);
    int q_l = -1;
    uint4 k0 = make_uint4(0, 0, 0, 0), k1v = k0;
    float ka_l = 0.f;
    if (lane < ne) {
      const int q = (int)fv_idx[e0 + lane];
      if (q >= 0 && q < in->n_kf && kf_valid[q]) { q_l = q; const uint4* a = (const uint4*)(kf_desc + 32 * (size_t)q); k0 = a[0]; k1v = a[1]; ka_l = kf_angle[q]; }   // no map point, or isBad() (:184-188): skipped
    }
    s_mf[lane] = -1;
    // (A) every distance of the slice first - entry j against this lane's (<= 4) register-resident features, one byte each
    // (saturated at 255: only distances <= TH_LOW can win, and 255 or 256 as the runner-up passes the same ratio test) -: 64
    // independent entries, throughput-bound.  (B) the walk, which is sequential by definition (:200 - an entry takes the closest
    // frame feature no earlier entry took), then has per entry only: mask, two wave minima, the decision.  (The first versions
    // computed the distances inside the walk: ~0.9 us of exposed instruction latency per entry on a lone wave, 175 us for the
    // 183-entry node of tools/track_latency.py's vocabulary.)
    for (int j = 0; j < ne; j++) {
      {
        uint4 a0, a1;
        a0.x = __builtin_amdgcn_readlane(k0.x, j); a0.y = __builtin_amdgcn_readlane(k0.y, j); a0.z = __builtin_amdgcn_readlane(k0.z, j); a0.w = __builtin_amdgcn_readlane(k0.w, j);
        a1.x = __builtin_amdgcn_readlane(k1v.x, j); a1.y = __builtin_amdgcn_readlane(k1v.y, j); a1.z = __builtin_amdgcn_readlane(k1v.z, j); a1.w = __builtin_amdgcn_readlane(k1v.w, j);
        unsigned pk = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const unsigned d = __popc(a0.x ^ c0[t].x) + __popc(a0.y ^ c0[t].y) + __popc(a0.z ^ c0[t].z) + __popc(a0.w ^ c0[t].w) + __popc(a1.x ^ c1[t].x) + __popc(a1.y ^ c1[t].y) +
                             __popc(a1.z ^ c1[t].z) + __popc(a1.w ^ c1[t].w);
          pk |= min(d, 255u) << (8 * t);
        }
        s_dj[j][lane] = pk;
      }
    }
    const unsigned have = (lane < cnt ? 1u : 0u) | (lane + 64 < cnt ? 2u : 0u) | (lane + 128 < cnt ? 4u : 0u) | (lane + 192 < cnt ? 8u : 0u);   // this lane's list positions that exist
    for (int j = 0; j < ne; j++) {
      const int q = __builtin_amdgcn_readlane(q_l, j);
      if (q < 0) continue;
      const unsigned dq = s_dj[j][lane];
      unsigned k1 = 256u << 16, k2 = 256u << 16;                        // (distance << 16 | list position): first minimum in list order
      auto offer = [&](unsigned d, int p) { const unsigned key = (d << 16) | (unsigned)p; if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key; };
      const unsigned avail = have & ~taken;
#pragma unroll
      for (int t = 0; t < 4; t++) {                                   // (selects, no branches: this is the sequential path)
        const unsigned key = ((avail >> t) & 1u) ? ((((dq >> (8 * t)) & 255u) << 16) | (unsigned)(lane + 64 * t)) : 0xFFFFFFFFu;
        const unsigned lo = min(key, k1), hi = max(key, k1);
        k1 = lo; k2 = min(k2, hi);
      }
      if (cnt > 256) {                                                // (a node with more than 256 frame features: the rest from memory)
        uint4 a0, a1;
        a0.x = __builtin_amdgcn_readlane(k0.x, j); a0.y = __builtin_amdgcn_readlane(k0.y, j); a0.z = __builtin_amdgcn_readlane(k0.z, j); a0.w = __builtin_amdgcn_readlane(k0.w, j);
        a1.x = __builtin_amdgcn_readlane(k1v.x, j); a1.y = __builtin_amdgcn_readlane(k1v.y, j); a1.z = __builtin_amdgcn_readlane(k1v.z, j); a1.w = __builtin_amdgcn_readlane(k1v.w, j);
        for (int p = lane + 256; p < cnt; p += 64) {
          if (s_taken[w][p]) continue;
          const uint4* tb = (const uint4*)(f_desc + 32 * (size_t)s_list[w][p]);
          const uint4 b0 = tb[0], b1 = tb[1];
          offer(min(255u, (unsigned)(__popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) +
                                     __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w))), p);
        }
      }
      // The two smallest keys of the wave as TWO wave minima on DPP only (no LDS crossbar on the sequential path): the smallest key,
      // then the smallest of what every lane has left without it (keys are unique: the list position is part of them).  Results
      // leave through lane 63.
      auto wave_min = [&](unsigned v) -> unsigned {
        v = min(v, dppu(v, std::integral_constant<int, 0xB1>()));       // quad_perm [1,0,3,2]
        v = min(v, dppu(v, std::integral_constant<int, 0x4E>()));       // quad_perm [2,3,0,1]
        v = min(v, dppu(v, std::integral_constant<int, 0x141>()));      // row_half_mirror
        v = min(v, dppu(v, std::integral_constant<int, 0x140>()));      // row_mirror: every lane of a row holds the row's minimum
        v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x142, 0xA, 0xF, false));     // row_bcast15 into rows 1 and 3
        v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x143, 0xC, 0xF, false));     // row_bcast31 into rows 2 and 3
        return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
      };
      const unsigned K1 = wave_min(k1);
      const unsigned K2 = wave_min(k1 == K1 ? k2 : k1);
      const int bestDist1 = (int)(K1 >> 16), bestDist2 = (int)(K2 >> 16), pos1 = (int)(K1 & 0xFFFFu);
      if (bestDist1 <= TRF_TH_LOW && (float)bestDist1 < ratio * (float)bestDist2) {
        // (feature index and both angles came with the slices: no load on the walk)
        const float ka = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ka_l), j));
        float fa; int f;
        if (pos1 < 256) {
          const int t = pos1 >> 6, l = pos1 & 63;
          const float v = t == 0 ? fa_l[0] : (t == 1 ? fa_l[1] : (t == 2 ? fa_l[2] : fa_l[3]));
          const int vi = t == 0 ? fi_l[0] : (t == 1 ? fi_l[1] : (t == 2 ? fi_l[2] : fi_l[3]));
          fa = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
          f = __builtin_amdgcn_readlane(vi, l);
          if (lane == l) taken |= 1u << t;
        } else {
          f = s_list[w][pos1];
          fa = kps4[4 * f + 3];
          if (lane == 0) s_taken[w][pos1] = 1;
        }
        if (lane == 0) {                                             // (recorded in LDS, written out behind the walk: a global store or atomic
          int b = 0;                                                  //  here puts a memory round trip on the sequential path - the compiler's
          if (check_ori) {                                            //  next vmcnt(0) wait - for every match)
            float rot = ka - fa;
            if (rot < 0.0) rot += 360.0f;
            b = (int)roundf(rot * (1.0f / TRK_HISTO));
            if (b == TRK_HISTO) b = 0;
          }
          s_mf[j] = f; s_mb[j] = b;
        }
        // (s_taken is touched by nodes beyond 256 features only: LDS operations of one wave execute in order, the compiler is told)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
    if (lane < ne && q_l >= 0) {                                      // the slice's matches, one lane each
      const int f = s_mf[lane];
      if (f >= 0) {
        match_kf[q_l] = f; f_owner[f] = q_l;
        if (check_ori) { const int b = s_mb[lane]; f_bin[f] = b; atomicAdd(&hist[b], 1); }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
  }
}

// one workgroup: the rotation check (ComputeThreeMaxima :1386-1418, the other bins' matches removed :240-251), PoseOptimization's
// observation list in feature order
__global__ __launch_bounds__(1024) void k_trf_finish(const TrfIn* __restrict__ in, const int* __restrict__ hist, const float* __restrict__ kps4, const int32_t* __restrict__ d_count,
                                                     int cap, const double* __restrict__ kf_Xw, int32_t* __restrict__ match_kf, int32_t* __restrict__ f_owner,
                                                     const int32_t* __restrict__ f_bin, int32_t* __restrict__ obs_feat, double* __restrict__ obs_Xw, double* __restrict__ obs_uv,
                                                     float* __restrict__ obs_w, int32_t* __restrict__ obs_off, double* __restrict__ pose7, double* __restrict__ K4d, TrkOut* __restrict__ out) {
  __shared__ int s_keep[TRK_HISTO], s_w[16], s_nm;
  const int tid = threadIdx.x;
  const int n_dev = *d_count;
  const int n = min(max(n_dev, 0), min(cap, TRK_MAXKP));
  if (tid == 0) {
    s_nm = 0;
    int top[3] = {-1, -1, -1}, pop[3] = {0, 0, 0};
    for (int b = 0; b < TRK_HISTO; b++) {                    // strict '>': the earlier bin wins ties
      const int h = hist[b];
      int r = 3;
      while (r > 0 && h > pop[r - 1]) r--;
      if (r == 3) continue;
      for (int mm = 2; mm > r; mm--) { top[mm] = top[mm - 1]; pop[mm] = pop[mm - 1]; }
      top[r] = b; pop[r] = h;
    }
    if ((float)pop[1] < 0.1f * (float)pop[0]) { top[1] = -1; top[2] = -1; }
    else if ((float)pop[2] < 0.1f * (float)pop[0]) { top[2] = -1; }
    for (int b = 0; b < TRK_HISTO; b++) s_keep[b] = (!in->check_ori || b == top[0] || b == top[1] || b == top[2]) ? 1 : 0;
  }
  __syncthreads();
  constexpr int FPT = TRK_MAXKP / 1024;
  int own[FPT]; int cntv = 0;
  for (int k = 0; k < FPT; k++) {
    const int t = FPT * tid + k;
    int o = t < n ? f_owner[t] : -1;
    if (o >= 0 && !s_keep[f_bin[t]]) { match_kf[o] = -1; o = -1; }
    if (t < cap) f_owner[t] = o;
    own[k] = o; cntv += o >= 0 ? 1 : 0;
  }
  int inc = cntv;
  {
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < w; k++) base += s_w[k];
    inc += base;
  }
  int pos = inc - cntv;
  for (int k = 0; k < FPT; k++) {
    const int t = FPT * tid + k;
    if (own[k] < 0) continue;
    const double* X = kf_Xw + 3 * (size_t)own[k];
    obs_feat[pos] = t;
    obs_Xw[3 * pos] = X[0]; obs_Xw[3 * pos + 1] = X[1]; obs_Xw[3 * pos + 2] = X[2];
    obs_uv[2 * pos] = (double)kps4[4 * t]; obs_uv[2 * pos + 1] = (double)kps4[4 * t + 1];
    obs_w[pos] = in->inv_sigma2[(int)kps4[4 * t + 2]];
    pos++;
  }
  if (tid == 1023) { obs_off[0] = 0; obs_off[1] = inc; out->nobs = inc; out->nmatches = inc; out->n_keypoints = n_dev; out->rounds = 0; out->cand_total = 0; }
  if (tid < 7) pose7[tid] = in->pose7[tid];
  if (tid < 4) K4d[tid] = (double)in->K4[tid];
}

}  // namespace orbhip

namespace orbhip {
void orbv_merge_host(const int32_t* word, const double* wt, const uint32_t* node, int n, uint32_t* bow_word, double* bow_value, int* n_words,
                     uint32_t* fv_node, uint32_t* fv_off, uint32_t* fv_idx, int* n_fv_nodes);      // orb_vocab.hip
int orbx_ctx_device(const orbx_ctx* c);
unsigned long long orbx_ctx_generation(const orbx_ctx* c);
int orbx_extract_chained(orbx_ctx* c, const uint8_t* d_img, int w, int h, int stride, orbx_keypoint* d_kps, uint8_t* d_desc, int cap,
                         int32_t* d_count, void* stream);      // orb_extractor.hip
}
using namespace orbhip;

namespace {
// The frame the motion-model step built stays on the device for the second stage of Tracking (orbt_track_local_map): its download
// block (keypoints, descriptors, count) and float records live in buffers of the calling thread that survive the call; the grid
// (FrameGridDev below) always did.
struct TrkFrame {
  DevBuf blk, kps4; FrameGridDev grid;
  size_t oKps = 0, oDesc = 0, oCnt = 0; int icap = 0, n_kp = 0, device = -1, nlevels = 0; float bounds[4] = {0, 0, 0, 0}; bool valid = false;
  const orbx_ctx* producer = nullptr;      // the extractor that produced the resident frame: its level count / scale tables / capacity are the frame's
  unsigned long long producer_gen = 0;     // ... and its generation: a context destroyed and re-created at the same address is NOT that extractor
  bool made_by(const orbx_ctx* c) const { return producer == c && producer_gen == orbhip::orbx_ctx_generation(c); }
};
thread_local TrkFrame g_trk_frame;
}  // namespace

// host wall time of the calling thread's most recent orbt_* call (orbt_last_call_ms): what a latency figure should be made of when
// the caller is an interpreter whose own locks add milliseconds around the call
static thread_local double g_last_call_ms = 0.0;
struct CallClock {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  ~CallClock() { g_last_call_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
extern "C" {
double orbt_last_call_ms(void) { return g_last_call_ms; }
#ifdef ORBHIP_TRK_PROF
int orbt_debug_prof(int* out, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trk_prof), sizeof(int) * 72) != hipSuccess) return -1;
  if (reset) { static int z[72]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_trk_prof), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif

int orbt_track_with_motion_model(orbx_ctx* ctx, const uint8_t* img, int w, int h, int stride, const float* K4, const float* bounds,
                                 const double* Tcw_pred, const double* last_Xw, const uint8_t* last_desc, const int32_t* last_octave,
                                 const float* last_angle, const uint8_t* last_valid, int n_last, float th, int check_ori,
                                 orbx_keypoint* kps_out, uint8_t* desc_out, int cap, int32_t* match_out, int32_t* owner_out,
                                 uint8_t* outlier_out, orbt_result* res) {
  CallClock call_clock;
  ORBHIP_REQUIRE(ctx && img && w > 0 && h > 0 && stride >= w && K4 && bounds && Tcw_pred && res && kps_out && desc_out && owner_out && outlier_out,
                 ORBHIP_EINVAL, "NULL argument");
  ORBHIP_REQUIRE(n_last >= 0 && n_last <= 4096 && (n_last == 0 || (last_Xw && last_desc && last_octave && last_angle && last_valid && match_out)), ORBHIP_EINVAL,
                 "bad last-frame arrays (at most 4096 features)");
  const int icap = orbx_max_keypoints(ctx);
  ORBHIP_REQUIRE(cap >= icap && icap <= TRK_MAXKP, ORBHIP_ECAP, "output capacity below orbx_max_keypoints(ctx) (or more than 4096 features per frame)");
  const int nlevels = orbx_get_levels(ctx);
  ORBHIP_REQUIRE(nlevels > 0 && nlevels <= 16, ORBHIP_EINVAL, "bad level count");
  ThreadWs& W = thread_ws();
  // ORBHIP_TRACK_TIMING=1: host wall time of the call's phases (mean over every 100 calls, on stderr)
  static const bool timing = []() { const char* e = std::getenv("ORBHIP_TRACK_TIMING"); return e && e[0] == '1'; }();
  static thread_local double t_acc[5] = {0, 0, 0, 0, 0}; static thread_local int t_n = 0;
  auto now_us = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = timing ? now_us() : 0.0;
  static thread_local uint32_t cand_cap_tl = 0;
  TrkFrame& TF = g_trk_frame;
  FrameGridDev& grid = TF.grid;                                // (off / idx buffers filled by k_trk_prepare)
  TF.valid = false;
  for (int attempt = 0; attempt < 2; attempt++) {
    int rc = W.begin();
    if (rc) return rc;
    // the workspace, the stream and every kernel of this call live on the default device; an extractor created on another one
    // would launch across devices
    ORBHIP_REQUIRE(orbhip::orbx_ctx_device(ctx) == W.device, ORBHIP_EINVAL, "the extractor context was created on another device than orbhip_set_default_device() selects");
    if (TF.device != W.device) { TF = TrkFrame(); TF.device = W.device; cand_cap_tl = 0; }
    const int nq = n_last;
    const uint32_t cand_cap = std::max<uint32_t>(cand_cap_tl, (uint32_t)std::max(nq, 1) * 64u);
    TrkIn I; std::memset(&I, 0, sizeof(I));
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) I.R[3 * r + c] = Tcw_pred[4 * r + c]; I.t[r] = Tcw_pred[4 * r + 3]; }
    { double T[16]; for (int k = 0; k < 12; k++) T[k] = Tcw_pred[k]; T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1; if (int r2 = ba_matrix4d_to_pose7(T, I.pose7)) return r2; }
    for (int k = 0; k < 4; k++) { I.K4[k] = K4[k]; I.bounds[k] = bounds[k]; }
    I.th = th; I.nq = nq; I.check_ori = check_ori ? 1 : 0; I.nlevels = nlevels;
    if (int r2 = orbx_get_tables(ctx, I.scale, nullptr, nullptr, I.inv_sigma2, nullptr)) return r2;
    // uploads: the image (straight from the caller's memory through pinned staging) and ONE packed block
    const size_t img_bytes = (size_t)stride * (h - 1) + w;
    uint8_t* d_img = W.up<uint8_t>(img, img_bytes, &rc);
    ThreadWs::Pack in;
    const int pI = in.add(&I, sizeof(I)), pX = in.add(last_Xw, 24 * (size_t)nq), pD = in.add(last_desc, 32 * (size_t)nq), pO = in.add(last_octave, 4 * (size_t)nq),
              pA = in.add(last_angle, 4 * (size_t)nq), pV = in.add(last_valid, (size_t)nq);
    if (rc || (rc = W.commit(in))) return rc;
    const double t1 = timing ? now_us() : 0.0;
    // ONE output block: [TrkOut | pose7 | summary | count | kps | desc | match | owner | obs_feat | outlier]
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~(size_t)255; return at; };
    const size_t oOut = take(sizeof(TrkOut)), oPose = take(56), oSum = take(sizeof(ba_summary)), oCnt = take(4), oNin = take(4), oKps = take((size_t)icap * sizeof(orbx_keypoint)),
                 oDesc = take((size_t)icap * 32), oMatch = take(4 * (size_t)std::max(nq, 1)), oOwner = take(4 * (size_t)icap), oFeat = take(4 * (size_t)icap),
                 oOutl = take((size_t)icap);
    if ((rc = TF.blk.ensure(o)) || (rc = TF.kps4.ensure(16 * (size_t)icap))) return rc;
    uint8_t* dblk = TF.blk.as<uint8_t>();
    float* d_quv = W.d<float>(2 * (size_t)std::max(nq, 1), &rc); float* d_qr = W.d<float>(std::max(nq, 1), &rc);
    int32_t* d_qlo = W.d<int32_t>(std::max(nq, 1), &rc); int32_t* d_qhi = W.d<int32_t>(std::max(nq, 1), &rc); uint8_t* d_qv = W.d<uint8_t>(std::max(nq, 1), &rc);
    float* d_kps4 = TF.kps4.as<float>();
    uint32_t* d_off = W.d<uint32_t>(2 * (size_t)nq + 4, &rc); uint2* d_pairs = W.d<uint2>(cand_cap, &rc);
    uint32_t* d_total = d_off + 2 * (size_t)nq + 2;
    double* d_oX = W.d<double>(3 * (size_t)icap, &rc); double* d_ouv = W.d<double>(2 * (size_t)icap, &rc); float* d_ow = W.d<float>(icap, &rc);
    int32_t* d_ooff = W.d<int32_t>(2, &rc); double* d_K4 = W.d<double>(4, &rc);
    uint32_t* d_acc = W.d<uint32_t>((size_t)std::max(nq, 1) * TRK_LMAX, &rc); int32_t* d_accn = W.d<int32_t>(std::max(nq, 1), &rc);
    if (rc) return rc;
    if ((rc = grid.off.ensure((size_t)(TRK_NCELL + 1) * 4)) || (rc = grid.idx.ensure((size_t)icap * 4))) return rc;
    grid.min_x = bounds[0]; grid.min_y = bounds[2];
    grid.winv = static_cast<float>(FRAME_GRID_COLS) / (bounds[1] - bounds[0]); grid.hinv = static_cast<float>(FRAME_GRID_ROWS) / (bounds[3] - bounds[2]);
    orbx_keypoint* d_kps = (orbx_keypoint*)(dblk + oKps); uint8_t* d_desc = dblk + oDesc; int32_t* d_count = (int32_t*)(dblk + oCnt);
    if ((rc = orbhip::orbx_extract_chained(ctx, d_img, w, h, stride, d_kps, d_desc, icap, d_count, (void*)W.s))) return rc;
    const TrkIn* dI = in.dev<TrkIn>(pI);
    hipLaunchKernelGGL(k_trk_prepare, dim3(1), dim3(1024), 0, W.s, dI, in.dev<double>(pX), in.dev<int32_t>(pO), in.dev<uint8_t>(pV), d_kps, d_count, icap, d_quv, d_qr,
                       d_qlo, d_qhi, d_qv, d_kps4, grid.off.as<uint32_t>(), grid.idx.as<uint32_t>(), nq, d_total);
    if (nq > 0) {
      hipLaunchKernelGGL(k_trk_windows, dim3((nq + 3) / 4), dim3(256), 0, W.s, d_kps4, grid.off.as<uint32_t>(), grid.idx.as<uint32_t>(), grid.min_x, grid.min_y, grid.winv,
                         grid.hinv, d_quv, d_qr, d_qlo, d_qhi, d_qv, nq, in.dev<uint8_t>(pD), d_desc, d_total, d_off, d_pairs, cand_cap, d_acc, d_accn, TRK_TH_HIGH);
    }
    // dynamic LDS of the greedy kernel: offsets, states, proposals (ints per query), two work lists, the candidate entries
    // (+ the inverted lists of the dataflow path: two bytes per entry behind the four of the entries themselves)
    const size_t lds_fixed = (size_t)(3 * nq + 1) * 4 + 2 * (size_t)((nq + 1) & ~1) * 2;
    const int ecap = (int)std::min<size_t>((size_t)std::max(nq, 1) * TRK_LMAX, (size_t)(100 * 1024 - lds_fixed) / 6);
    const int tcap = ecap;
    const size_t lds_greedy = lds_fixed + (size_t)ecap * 6 + 16;
    static const int force_rounds = []() { const char* e = ORBHIP_EXP_ENV("ORBHIP_TRACK_ROUNDS"); return (e && e[0] == '1') ? 1 : 0; }();
    if ((rc = raise_dynamic_lds((const void*)k_trk_greedy, W.device, 100 * 1024 + 64))) return rc;
    hipLaunchKernelGGL(k_trk_greedy, dim3(1), dim3(TRK_GT), lds_greedy, W.s, dI, d_qv, in.dev<float>(pA), d_acc, d_accn, in.dev<double>(pX), d_off, d_total, d_pairs, cand_cap, d_kps4, d_count, icap,
                       (int32_t*)(dblk + oMatch), (int32_t*)(dblk + oOwner), (int32_t*)(dblk + oFeat), d_oX, d_ouv, d_ow, d_ooff, (double*)(dblk + oPose), d_K4,
                       (TrkOut*)(dblk + oOut), ecap, tcap, force_rounds, nq, I.check_ori);
    ORBHIP_CHECK_HIP(hipGetLastError());
    if ((rc = ba_pose_optimization_batch_device(d_K4, (double*)(dblk + oPose), d_oX, d_ouv, d_ow, d_ooff, 1, dblk + oOutl, (int32_t*)(dblk + oNin),
                                                (ba_summary*)(dblk + oSum), (void*)W.s))) return rc;
    const uint8_t* hb = W.down(dblk, o, &rc);
    const double t2 = timing ? now_us() : 0.0;
    if (rc || (rc = W.sync())) return rc;
    const double t3 = timing ? now_us() : 0.0;
    const TrkOut* T = (const TrkOut*)(hb + oOut);
    if (T->n_keypoints < 0) { set_error("extractor capacity exceeded"); return ORBHIP_EOVERFLOW; }
    if ((uint32_t)T->cand_total > cand_cap) { cand_cap_tl = (uint32_t)T->cand_total + (uint32_t)T->cand_total / 4; continue; }   // (rare) once more, larger lists
    cand_cap_tl = std::max<uint32_t>(cand_cap_tl, (uint32_t)T->cand_total + (uint32_t)T->cand_total / 8);
    const int n = std::min(T->n_keypoints, icap);
    std::memcpy(kps_out, hb + oKps, (size_t)n * sizeof(orbx_keypoint)); std::memcpy(desc_out, hb + oDesc, (size_t)n * 32);
    if (nq) std::memcpy(match_out, hb + oMatch, 4 * (size_t)nq);
    std::memcpy(owner_out, hb + oOwner, 4 * (size_t)n);
    std::memset(outlier_out, 0, (size_t)n);
    const int32_t* feat = (const int32_t*)(hb + oFeat);
    for (int k = 0; k < T->nobs; k++) outlier_out[feat[k]] = hb[oOutl + k];
    TF.oKps = oKps; TF.oDesc = oDesc; TF.oCnt = oCnt; TF.icap = icap; TF.n_kp = n; TF.nlevels = nlevels; std::memcpy(TF.bounds, bounds, 16); TF.valid = true; TF.producer = ctx; TF.producer_gen = orbhip::orbx_ctx_generation(ctx);
    res->n_keypoints = n; res->nmatches = T->nmatches; res->n_correspondences = T->nobs; res->greedy_rounds = T->rounds % 1000;
    std::memcpy(res->pose7, hb + oPose, 56);
    // PoseOptimization returns 0 and leaves the pose alone with fewer than 3 correspondences (src/CeresOptimizer.cc:330)
    res->n_inliers = T->nobs < 3 ? 0 : *(const int32_t*)(hb + oNin);
    if (T->nobs < 3) std::memcpy(res->pose7, I.pose7, 56);
    if (timing) {
      const double t4 = now_us();
      t_acc[0] += t1 - t0; t_acc[1] += t2 - t1; t_acc[2] += t3 - t2; t_acc[3] += t4 - t3; t_acc[4] += t4 - t0;
      if (t4 - t0 > 2000.0)                 // an outlier: which phase waited?
        fprintf(stderr, "orbt_track_with_motion_model SLOW CALL: staging + upload enqueue %.1f us, kernel launches %.1f us, wait %.1f us, unpack %.1f us, total %.1f us\n",
                t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0);
      if (++t_n == 100) {
        fprintf(stderr, "k_trk_greedy ticks (10 ns): setup %d rounds %d (%d rounds, %d sweeps; lists inverted at %d, first pass done at %d; setup: cleared at %d, counts scanned at %d) rotation+owners %d observations %d\n", T->ticks[0], T->ticks[1], T->rounds % 1000, T->rounds / 1000, T->ticks[4], T->ticks[5], T->ticks[6], T->ticks[7], T->ticks[2], T->ticks[3]);
        fprintf(stderr, "orbt_track_with_motion_model: staging + upload enqueue %.1f us, kernel launches %.1f us, wait %.1f us, unpack %.1f us, total %.1f us\n",
                t_acc[0] / 100, t_acc[1] / 100, t_acc[2] / 100, t_acc[3] / 100, t_acc[4] / 100);
        for (double& a : t_acc) a = 0; t_n = 0;
      }
    }
    return 0;
  }
  set_error("window candidate lists did not fit after regrowing");
  return ORBHIP_ENOMEM;
}

int orbt_track_local_map(orbx_ctx* ctx, const float* K4, const float* bounds, const double* Tcw, float log_scale_factor,
                         const double* mp_Xw, const double* mp_normal, const float* mp_min_dist, const float* mp_max_dist, const uint8_t* mp_desc,
                         const uint8_t* mp_state, int n_mp, const double* slot_Xw, const uint8_t* slot_state, int n_kp, float th, float nnratio,
                         uint8_t* mp_in_view, int32_t* mp_match, int32_t* slot_owner, uint8_t* outlier_out, orbt_result* res) {
  CallClock call_clock;
  ORBHIP_REQUIRE(ctx && K4 && bounds && Tcw && res && mp_in_view && mp_match && slot_owner && outlier_out && slot_Xw && slot_state, ORBHIP_EINVAL, "NULL argument");
  ORBHIP_REQUIRE(n_mp >= 0 && n_mp <= TLM_MAXMP, ORBHIP_ECAP, "more than 16384 local map points per call");
  ORBHIP_REQUIRE(n_mp == 0 || (mp_Xw && mp_normal && mp_min_dist && mp_max_dist && mp_desc && mp_state), ORBHIP_EINVAL, "NULL map-point argument");
  TrkFrame& TF = g_trk_frame;
  ThreadWs& W = thread_ws();
  static thread_local uint32_t cand_cap_tl = 0;
  const int nlevels = orbx_get_levels(ctx);
  ORBHIP_REQUIRE(nlevels > 0 && nlevels <= 16, ORBHIP_EINVAL, "bad level count");
  for (int attempt = 0; attempt < 2; attempt++) {
    int rc = W.begin();
    if (rc) return rc;
    ORBHIP_REQUIRE(TF.valid && TF.device == W.device, ORBHIP_EINVAL, "no frame resident on this thread's device: call orbt_track_with_motion_model first (same host thread)");
    ORBHIP_REQUIRE(TF.made_by(ctx) && TF.nlevels == nlevels, ORBHIP_EINVAL, "ctx is not the extractor that produced the resident frame");
    ORBHIP_REQUIRE(n_kp == TF.n_kp, ORBHIP_EINVAL, "n_kp differs from the resident frame's keypoint count");
    const int icap = TF.icap, nq = n_mp;
    TlmIn I; std::memset(&I, 0, sizeof(I));
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) I.C.R[3 * r + c] = Tcw[4 * r + c]; I.C.t[r] = Tcw[4 * r + 3]; }
    for (int k = 0; k < 3; k++) I.C.Ow[k] = -(I.C.R[k] * I.C.t[0] + I.C.R[3 + k] * I.C.t[1] + I.C.R[6 + k] * I.C.t[2]);      // Ow = -Rcw^T tcw (src/Frame.cc:188)
    I.C.fx = K4[0]; I.C.fy = K4[1]; I.C.cx = K4[2]; I.C.cy = K4[3];
    I.C.min_x = bounds[0]; I.C.max_x = bounds[1]; I.C.min_y = bounds[2]; I.C.max_y = bounds[3];
    I.C.cos_limit = 0.5f; I.C.log_scale = log_scale_factor; I.C.nlevels = nlevels;          // isInFrustum(map_point, 0.5) (src/Tracking.cc:822)
    { double T[16]; for (int k = 0; k < 12; k++) T[k] = Tcw[k]; T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1; if (int r2 = ba_matrix4d_to_pose7(T, I.pose7)) return r2; }
    for (int k = 0; k < 4; k++) I.K4[k] = K4[k];
    if (int r2 = orbx_get_tables(ctx, I.scale, nullptr, nullptr, I.inv_sigma2, nullptr)) return r2;
    I.th = th; I.ratio = nnratio; I.n_mp = n_mp; I.n_kp_host = n_kp;
    // a distance can win only if it is <= TH_HIGH and can veto through the ratio test only if ratio * d < TH_HIGH
    { int dm = TRK_TH_HIGH; while (dm < 255 && nnratio * (float)(dm + 1) < (float)TRK_TH_HIGH) dm++; I.dmax = dm; }
    const uint32_t cand_cap = std::max<uint32_t>(cand_cap_tl, (uint32_t)std::max(nq, 1) * 32u);
    ThreadWs::Pack in;
    const int pI = in.add(&I, sizeof(I)), pX = in.add(mp_Xw, 24 * (size_t)nq), pN = in.add(mp_normal, 24 * (size_t)nq), pMi = in.add(mp_min_dist, 4 * (size_t)nq),
              pMa = in.add(mp_max_dist, 4 * (size_t)nq), pD = in.add(mp_desc, 32 * (size_t)nq), pS = in.add(mp_state, (size_t)nq), pSX = in.add(slot_Xw, 24 * (size_t)n_kp),
              pSS = in.add(slot_state, (size_t)n_kp);
    if ((rc = W.commit(in))) return rc;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~(size_t)255; return at; };
    const size_t oOut = take(sizeof(TrkOut)), oPose = take(56), oSum = take(sizeof(ba_summary)), oNin = take(4), oView = take((size_t)std::max(nq, 1)),
                 oMatch = take(4 * (size_t)std::max(nq, 1)), oOwner = take(4 * (size_t)icap), oFeat = take(4 * (size_t)icap), oOutl = take((size_t)icap);
    uint8_t* dblk = W.d<uint8_t>(o, &rc);
    float* d_quv = W.d<float>(2 * (size_t)std::max(nq, 1), &rc); float* d_qr = W.d<float>(std::max(nq, 1), &rc);
    int32_t* d_qlo = W.d<int32_t>(std::max(nq, 1), &rc); int32_t* d_qhi = W.d<int32_t>(std::max(nq, 1), &rc); uint8_t* d_qv = W.d<uint8_t>(std::max(nq, 1), &rc);
    uint32_t* d_off = W.d<uint32_t>(2 * (size_t)nq + 4, &rc); uint2* d_pairs = W.d<uint2>(cand_cap, &rc);
    uint32_t* d_total = d_off + 2 * (size_t)nq + 2;
    double* d_oX = W.d<double>(3 * (size_t)icap, &rc); double* d_ouv = W.d<double>(2 * (size_t)icap, &rc); float* d_ow = W.d<float>(icap, &rc);
    int32_t* d_ooff = W.d<int32_t>(2, &rc); double* d_K4 = W.d<double>(4, &rc);
    uint32_t* d_acc = W.d<uint32_t>((size_t)std::max(nq, 1) * TRK_LMAX, &rc); int32_t* d_accn = W.d<int32_t>(std::max(nq, 1), &rc);
    if (rc) return rc;
    const TlmIn* dI = in.dev<TlmIn>(pI);
    const uint8_t* fblk = TF.blk.as<uint8_t>();
    const float* d_kps4 = TF.kps4.as<float>();
    const FrameGridDev& grid = TF.grid;
    hipLaunchKernelGGL(k_tlm_frustum, dim3((std::max(nq, 1) + 255) / 256), dim3(256), 0, W.s, dI, in.dev<double>(pX), in.dev<double>(pN), in.dev<float>(pMi), in.dev<float>(pMa),
                       in.dev<uint8_t>(pS), dblk + oView, d_quv, d_qr, d_qlo, d_qhi, d_qv, d_total);
    if (nq > 0)
      hipLaunchKernelGGL(k_trk_windows, dim3((nq + 3) / 4), dim3(256), 0, W.s, d_kps4, grid.off.as<uint32_t>(), grid.idx.as<uint32_t>(), grid.min_x, grid.min_y, grid.winv,
                         grid.hinv, d_quv, d_qr, d_qlo, d_qhi, d_qv, nq, in.dev<uint8_t>(pD), fblk + TF.oDesc, d_total, d_off, d_pairs, cand_cap, d_acc, d_accn, I.dmax);
    if ((rc = raise_dynamic_lds((const void*)k_tlm_greedy, W.device, 4 * TLM_MAXMP))) return rc;
    hipLaunchKernelGGL(k_tlm_greedy, dim3(1), dim3(1024), 4 * TLM_MAXMP, W.s, dI, d_qv, d_acc, d_accn, d_off, d_total, d_pairs, cand_cap, d_kps4, (const int32_t*)(fblk + TF.oCnt), icap,
                       in.dev<double>(pX), in.dev<double>(pSX), in.dev<uint8_t>(pSS), (int32_t*)(dblk + oMatch), (int32_t*)(dblk + oOwner), (int32_t*)(dblk + oFeat), d_oX, d_ouv, d_ow,
                       d_ooff, (double*)(dblk + oPose), d_K4, (TrkOut*)(dblk + oOut), nq);
    ORBHIP_CHECK_HIP(hipGetLastError());
    if ((rc = ba_pose_optimization_batch_device(d_K4, (double*)(dblk + oPose), d_oX, d_ouv, d_ow, d_ooff, 1, dblk + oOutl, (int32_t*)(dblk + oNin),
                                                (ba_summary*)(dblk + oSum), (void*)W.s))) return rc;
    const uint8_t* hb = W.down(dblk, o, &rc);
    if (rc || (rc = W.sync())) return rc;
    const TrkOut* T = (const TrkOut*)(hb + oOut);
    if ((uint32_t)T->cand_total > cand_cap) { cand_cap_tl = (uint32_t)T->cand_total + (uint32_t)T->cand_total / 4; continue; }   // (rare) once more, larger lists
    cand_cap_tl = std::max<uint32_t>(cand_cap_tl, (uint32_t)T->cand_total + (uint32_t)T->cand_total / 8);
    if (nq) { std::memcpy(mp_in_view, hb + oView, (size_t)nq); std::memcpy(mp_match, hb + oMatch, 4 * (size_t)nq); }
    std::memcpy(slot_owner, hb + oOwner, 4 * (size_t)n_kp);
    std::memset(outlier_out, 0, (size_t)n_kp);
    const int32_t* feat = (const int32_t*)(hb + oFeat);
    for (int k = 0; k < T->nobs; k++) outlier_out[feat[k]] = hb[oOutl + k];
    int nview = 0; for (int i = 0; i < nq; i++) nview += mp_in_view[i];
    res->n_keypoints = n_kp; res->nmatches = T->nmatches; res->n_correspondences = T->nobs; res->greedy_rounds = T->rounds; res->reserved = nview;
    std::memcpy(res->pose7, hb + oPose, 56);
    res->n_inliers = T->nobs < 3 ? 0 : *(const int32_t*)(hb + oNin);
    if (T->nobs < 3) std::memcpy(res->pose7, I.pose7, 56);
    return 0;
  }
  set_error("window candidate lists did not fit after regrowing");
  return ORBHIP_ENOMEM;
}

int orbt_track_reference_keyframe(orbx_ctx* ctx, orbv_ctx* voc, const uint8_t* img, int w, int h, int stride, const float* K4, const float* bounds, const double* Tcw_last,
                                  const uint8_t* kf_desc, const uint8_t* kf_valid, const float* kf_angle, const double* kf_Xw, int n_kf,
                                  const uint32_t* kf_fv_node, const uint32_t* kf_fv_off, const uint32_t* kf_fv_idx, int kf_fv_n, float nnratio, int check_ori,
                                  orbx_keypoint* kps_out, uint8_t* desc_out, int cap, uint32_t* bow_word, double* bow_value, int* n_words, uint32_t* fv_node,
                                  uint32_t* fv_off, uint32_t* fv_idx, int* n_fv_nodes, int32_t* match_kf, int32_t* slot_owner, uint8_t* outlier_out, orbt_result* res) {
  CallClock call_clock;
  ORBHIP_REQUIRE(ctx && voc && K4 && bounds && Tcw_last && res && match_kf && slot_owner && outlier_out && n_words && n_fv_nodes && fv_off, ORBHIP_EINVAL, "NULL argument");
  ORBHIP_REQUIRE(!img || (w > 0 && h > 0 && stride >= w), ORBHIP_EINVAL, "bad image dimensions");
  ORBHIP_REQUIRE(n_kf >= 0 && kf_fv_n >= 0 && (n_kf == 0 || (kf_desc && kf_valid && kf_angle && kf_Xw)) && (kf_fv_n == 0 || (kf_fv_node && kf_fv_off && kf_fv_idx)), ORBHIP_EINVAL, "NULL keyframe argument");
  const int icap = orbx_max_keypoints(ctx);
  ORBHIP_REQUIRE(icap <= TRK_MAXKP, ORBHIP_ECAP, "more than 4096 features per frame");
  const int nlevels = orbx_get_levels(ctx);
  ORBHIP_REQUIRE(nlevels > 0 && nlevels <= 16, ORBHIP_EINVAL, "bad level count");
  TrkFrame& TF = g_trk_frame;
  ThreadWs& W = thread_ws();
  int rc = W.begin();
  if (rc) return rc;
  ORBHIP_REQUIRE(orbhip::orbx_ctx_device(ctx) == W.device, ORBHIP_EINVAL, "the extractor context was created on another device than orbhip_set_default_device() selects");
  if (TF.device != W.device) { TF = TrkFrame(); TF.device = W.device; }
  TrfIn I; std::memset(&I, 0, sizeof(I));
  { double T[16]; for (int k = 0; k < 12; k++) T[k] = Tcw_last[k]; T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1; if (int r2 = ba_matrix4d_to_pose7(T, I.pose7)) return r2; }
  float scale[16];
  for (int k = 0; k < 4; k++) I.K4[k] = K4[k];
  if (int r2 = orbx_get_tables(ctx, scale, nullptr, nullptr, I.inv_sigma2, nullptr)) return r2;
  I.ratio = nnratio; I.check_ori = check_ori ? 1 : 0; I.nn = kf_fv_n; I.n_kf = n_kf;
  const uint32_t n_fv_idx = kf_fv_n ? kf_fv_off[kf_fv_n] : 0u;
  ThreadWs::Pack in;
  TrkIn PI; std::memset(&PI, 0, sizeof(PI));                   // k_trk_prepare's constants (no queries: it builds the frame's records and grid)
  for (int k = 0; k < 4; k++) { PI.K4[k] = K4[k]; PI.bounds[k] = bounds[k]; }
  PI.nlevels = nlevels; std::memcpy(PI.scale, scale, sizeof(scale));
  const int pI = in.add(&I, sizeof(I)), pP = in.add(&PI, sizeof(PI)), pD = in.add(kf_desc, 32 * (size_t)n_kf), pV = in.add(kf_valid, (size_t)n_kf), pA = in.add(kf_angle, 4 * (size_t)n_kf),
            pX = in.add(kf_Xw, 24 * (size_t)n_kf), pFn = in.add(kf_fv_node, 4 * (size_t)kf_fv_n), pFo = in.add(kf_fv_off, 4 * ((size_t)kf_fv_n + 1)), pFi = in.add(kf_fv_idx, 4 * (size_t)n_fv_idx);
  uint8_t* d_img = nullptr;
  if (img) {
    ORBHIP_REQUIRE(kps_out && desc_out && cap >= icap, ORBHIP_ECAP, "output capacity below orbx_max_keypoints(ctx)");
    d_img = W.up<uint8_t>(img, (size_t)stride * (h - 1) + w, &rc);
  } else {
    ORBHIP_REQUIRE(TF.valid, ORBHIP_EINVAL, "img == NULL needs the frame of an earlier orbt_* call of this thread on the device");
    ORBHIP_REQUIRE(TF.made_by(ctx) && TF.nlevels == nlevels && TF.icap == icap, ORBHIP_EINVAL, "ctx is not the extractor that produced the resident frame");
  }
  if (rc || (rc = W.commit(in))) return rc;
  if (img) {
    // the same frame block orbt_track_with_motion_model leaves behind: [.. | count | keypoints | descriptors | ..]
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~(size_t)255; return at; };
    const size_t oCnt = take(4), oKps = take((size_t)icap * sizeof(orbx_keypoint)), oDesc = take((size_t)icap * 32);
    TF.valid = false;
    if ((rc = TF.blk.ensure(o)) || (rc = TF.kps4.ensure(16 * (size_t)icap)) || (rc = TF.grid.off.ensure((size_t)(TRK_NCELL + 1) * 4)) || (rc = TF.grid.idx.ensure((size_t)icap * 4))) return rc;
    TF.oCnt = oCnt; TF.oKps = oKps; TF.oDesc = oDesc; TF.icap = icap; TF.nlevels = nlevels; std::memcpy(TF.bounds, bounds, 16);
    TF.grid.min_x = bounds[0]; TF.grid.min_y = bounds[2];
    TF.grid.winv = static_cast<float>(FRAME_GRID_COLS) / (bounds[1] - bounds[0]); TF.grid.hinv = static_cast<float>(FRAME_GRID_ROWS) / (bounds[3] - bounds[2]);
    uint8_t* fb = TF.blk.as<uint8_t>();
    if ((rc = orbhip::orbx_extract_chained(ctx, d_img, w, h, stride, (orbx_keypoint*)(fb + oKps), fb + oDesc, icap, (int32_t*)(fb + oCnt), (void*)W.s))) return rc;
    float* d_dummy = W.d<float>(4, &rc); int32_t* d_di = W.d<int32_t>(4, &rc); uint8_t* d_db = W.d<uint8_t>(4, &rc); uint32_t* d_tot = W.d<uint32_t>(1, &rc);
    if (rc) return rc;
    hipLaunchKernelGGL(k_trk_prepare, dim3(1), dim3(1024), 0, W.s, in.dev<TrkIn>(pP), (const double*)nullptr, (const int32_t*)nullptr, (const uint8_t*)nullptr,
                       (const orbx_keypoint*)(fb + oKps), (const int32_t*)(fb + oCnt), icap, d_dummy, d_dummy, d_di, d_di, d_db, TF.kps4.as<float>(),
                       TF.grid.off.as<uint32_t>(), TF.grid.idx.as<uint32_t>(), 0, d_tot);
  }
  const uint8_t* fblk = TF.blk.as<uint8_t>();
  const float* d_kps4 = TF.kps4.as<float>();
  const int32_t* d_count = (const int32_t*)(fblk + TF.oCnt);
  const uint8_t* d_fdesc = fblk + TF.oDesc;
  const int fcap = TF.icap;
  // output block: [TrkOut | pose7 | summary | n_inliers | word | weight | node | match_kf | owner | obs_feat | outlier]
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~(size_t)255; return at; };
  const size_t oOut = take(sizeof(TrkOut)), oPose = take(56), oSum = take(sizeof(ba_summary)), oNin = take(4), oWord = take(4 * (size_t)fcap), oWt = take(8 * (size_t)fcap),
               oNode = take(4 * (size_t)fcap), oMatch = take(4 * (size_t)std::max(n_kf, 1)), oOwner = take(4 * (size_t)fcap), oFeat = take(4 * (size_t)fcap), oOutl = take((size_t)fcap);
  uint8_t* dblk = W.d<uint8_t>(o, &rc);
  int32_t* d_bin = W.d<int32_t>(fcap, &rc); int* d_hist = W.d<int>(TRK_HISTO + 2, &rc);
  double* d_oX = W.d<double>(3 * (size_t)fcap, &rc); double* d_ouv = W.d<double>(2 * (size_t)fcap, &rc); float* d_ow = W.d<float>(fcap, &rc);
  int32_t* d_ooff = W.d<int32_t>(2, &rc); double* d_K4 = W.d<double>(4, &rc);
  if (rc) return rc;
  // Frame::ComputeBoW (src/Frame.cc:322-327: levelsup 4); the rows behind the keypoint count are descended too and ignored
  if ((rc = orbv_descend_device(voc, d_fdesc, fcap, 4, (int32_t*)(dblk + oWord), (double*)(dblk + oWt), (uint32_t*)(dblk + oNode), (void*)W.s))) return rc;
  const TrfIn* dI = in.dev<TrfIn>(pI);
  hipLaunchKernelGGL(k_trf_init, dim3((std::max(std::max(n_kf, fcap), 64) + 255) / 256), dim3(256), 0, W.s, (int32_t*)(dblk + oMatch), n_kf, (int32_t*)(dblk + oOwner), d_bin, fcap, d_hist);
  if (kf_fv_n > 0)
    hipLaunchKernelGGL(k_trf_bow, dim3(std::max(kf_fv_n, 1)), dim3(64), 0, W.s, dI, in.dev<uint32_t>(pFn), in.dev<uint32_t>(pFo), in.dev<uint32_t>(pFi), in.dev<uint8_t>(pD),
                       in.dev<uint8_t>(pV), in.dev<float>(pA), d_fdesc, (const uint32_t*)(dblk + oNode), (const double*)(dblk + oWt), d_kps4, d_count, fcap,
                       (int32_t*)(dblk + oMatch), (int32_t*)(dblk + oOwner), d_bin, d_hist);
  hipLaunchKernelGGL(k_trf_finish, dim3(1), dim3(1024), 0, W.s, dI, d_hist, d_kps4, d_count, fcap, in.dev<double>(pX), (int32_t*)(dblk + oMatch), (int32_t*)(dblk + oOwner), d_bin,
                     (int32_t*)(dblk + oFeat), d_oX, d_ouv, d_ow, d_ooff, (double*)(dblk + oPose), d_K4, (TrkOut*)(dblk + oOut));
  ORBHIP_CHECK_HIP(hipGetLastError());
  if ((rc = ba_pose_optimization_batch_device(d_K4, (double*)(dblk + oPose), d_oX, d_ouv, d_ow, d_ooff, 1, dblk + oOutl, (int32_t*)(dblk + oNin), (ba_summary*)(dblk + oSum), (void*)W.s))) return rc;
  const uint8_t* hb = W.down(dblk, o, &rc);
  const uint8_t* hf = img ? W.down(fblk, TF.oDesc + (size_t)fcap * 32, &rc) : nullptr;
  if (rc || (rc = W.sync())) return rc;
  const TrkOut* T = (const TrkOut*)(hb + oOut);
  if (T->n_keypoints < 0) { set_error("extractor capacity exceeded"); return ORBHIP_EOVERFLOW; }
  const int n = std::min(T->n_keypoints, fcap);
  if (img) { TF.n_kp = n; TF.valid = true; TF.producer = ctx; TF.producer_gen = orbhip::orbx_ctx_generation(ctx); std::memcpy(kps_out, hf + TF.oKps, (size_t)n * sizeof(orbx_keypoint)); std::memcpy(desc_out, hf + TF.oDesc, (size_t)n * 32); }
  if (n != TF.n_kp) { set_error("resident frame changed"); return ORBHIP_EINVAL; }
  if (bow_word && bow_value && fv_node && fv_idx)
    orbhip::orbv_merge_host((const int32_t*)(hb + oWord), (const double*)(hb + oWt), (const uint32_t*)(hb + oNode), n, bow_word, bow_value, n_words, fv_node, fv_off, fv_idx, n_fv_nodes);
  else { *n_words = 0; *n_fv_nodes = 0; fv_off[0] = 0; }
  if (n_kf) std::memcpy(match_kf, hb + oMatch, 4 * (size_t)n_kf);
  std::memcpy(slot_owner, hb + oOwner, 4 * (size_t)n);
  std::memset(outlier_out, 0, (size_t)n);
  const int32_t* feat = (const int32_t*)(hb + oFeat);
  for (int k = 0; k < T->nobs; k++) outlier_out[feat[k]] = hb[oOutl + k];
  res->n_keypoints = n; res->nmatches = T->nmatches; res->n_correspondences = T->nobs; res->greedy_rounds = 0; res->reserved = 0;
  std::memcpy(res->pose7, hb + oPose, 56);
  res->n_inliers = T->nobs < 3 ? 0 : *(const int32_t*)(hb + oNin);
  if (T->nobs < 3) std::memcpy(res->pose7, I.pose7, 56);
  return 0;
}


// ---- Tracking::Relocalization, first stage (src/Tracking.cc:979-1029): current_frame_.ComputeBoW() and, for EVERY candidate keyframe,
// matcher.SearchByBoW(keyframe, current_frame_, map_point_matches_vector[i]) with ORBmatcher(0.75, true) - in ONE call: the frame's
// vocabulary descent once, then the three kernels of the reference-keyframe step per candidate (k_trf_init, k_trf_bow: one wave per
// vocabulary node, k_trf_finish: the rotation histogram), all on the calling thread's stream, one upload, one download.  The
// candidates are independent (each fills its own vpMapPointMatches).  What follows in the reference - PnPsolver RANSAC per candidate,
// then PoseOptimization / SearchByProjection(F, KF, found, th, ORBdist) rounds - starts from a PnP pose and stays with the caller
// (PnP is out of scope; the rounds are ba_pose_optimization and orbm_search_by_projection's reloc_kf form).
int orbt_relocalization_search_by_bow(orbx_ctx* ctx, orbv_ctx* voc, const uint8_t* img, int w, int h, int stride, const float* K4, const float* bounds,
                                      const orbt_reloc_keyframe* cand, int n_cand, float nnratio, int check_ori, orbx_keypoint* kps_out, uint8_t* desc_out, int cap,
                                      uint32_t* bow_word, double* bow_value, int* n_words, uint32_t* fv_node, uint32_t* fv_off, uint32_t* fv_idx, int* n_fv_nodes,
                                      int32_t* slot_owner, int32_t* nmatches, int* n_keypoints) {
  CallClock call_clock;
  ORBHIP_REQUIRE(ctx && voc && K4 && bounds && n_words && n_fv_nodes && fv_off && n_keypoints && n_cand >= 0 && (n_cand == 0 || (cand && slot_owner && nmatches)), ORBHIP_EINVAL, "NULL argument");
  ORBHIP_REQUIRE(!img || (w > 0 && h > 0 && stride >= w), ORBHIP_EINVAL, "bad image dimensions");
  const int icap = orbx_max_keypoints(ctx);
  ORBHIP_REQUIRE(icap <= TRK_MAXKP, ORBHIP_ECAP, "more than 4096 features per frame");
  ORBHIP_REQUIRE(cap >= icap, ORBHIP_ECAP, "output capacity below orbx_max_keypoints(ctx)");
  const int nlevels = orbx_get_levels(ctx);
  ORBHIP_REQUIRE(nlevels > 0 && nlevels <= 16, ORBHIP_EINVAL, "bad level count");
  for (int i = 0; i < n_cand; i++)
    ORBHIP_REQUIRE(cand[i].n >= 0 && cand[i].fv_n >= 0 && (cand[i].n == 0 || (cand[i].desc && cand[i].valid && cand[i].angle)) &&
                   (cand[i].fv_n == 0 || (cand[i].fv_node && cand[i].fv_off && cand[i].fv_idx)), ORBHIP_EINVAL, "NULL candidate argument");
  TrkFrame& TF = g_trk_frame;
  ThreadWs& W = thread_ws();
  int rc = W.begin();
  if (rc) return rc;
  ORBHIP_REQUIRE(orbhip::orbx_ctx_device(ctx) == W.device, ORBHIP_EINVAL, "the extractor context was created on another device than orbhip_set_default_device() selects");
  if (TF.device != W.device) { TF = TrkFrame(); TF.device = W.device; }
  float scale[16], inv_sigma2[16];
  if (int r2 = orbx_get_tables(ctx, scale, nullptr, nullptr, inv_sigma2, nullptr)) return r2;
  ThreadWs::Pack in;
  TrkIn PI; std::memset(&PI, 0, sizeof(PI));
  for (int k = 0; k < 4; k++) { PI.K4[k] = K4[k]; PI.bounds[k] = bounds[k]; }
  PI.nlevels = nlevels; std::memcpy(PI.scale, scale, sizeof(scale));
  const int pP = in.add(&PI, sizeof(PI));
  std::vector<TrfIn> h_in((size_t)std::max(n_cand, 1));
  struct Pieces { int d, v, a, fn, fo, fi; };
  std::vector<Pieces> pc((size_t)std::max(n_cand, 1));
  for (int i = 0; i < n_cand; i++) {
    const orbt_reloc_keyframe& q = cand[i];
    TrfIn& I = h_in[i]; std::memset(&I, 0, sizeof(I));
    I.pose7[6] = 1.0;
    for (int k = 0; k < 4; k++) I.K4[k] = K4[k];
    std::memcpy(I.inv_sigma2, inv_sigma2, sizeof(inv_sigma2));
    I.ratio = nnratio; I.check_ori = check_ori ? 1 : 0; I.nn = q.fv_n; I.n_kf = q.n;
    const uint32_t nidx = q.fv_n ? q.fv_off[q.fv_n] : 0u;
    pc[i].d = in.add(q.desc, 32 * (size_t)q.n); pc[i].v = in.add(q.valid, (size_t)q.n); pc[i].a = in.add(q.angle, 4 * (size_t)q.n);
    pc[i].fn = in.add(q.fv_node, 4 * (size_t)q.fv_n); pc[i].fo = in.add(q.fv_off, 4 * ((size_t)q.fv_n + 1)); pc[i].fi = in.add(q.fv_idx, 4 * (size_t)nidx);
  }
  const int pI = in.add(h_in.data(), sizeof(TrfIn) * (size_t)std::max(n_cand, 1));
  uint8_t* d_img = nullptr;
  if (img) d_img = W.up<uint8_t>(img, (size_t)stride * (h - 1) + w, &rc);
  else {
    ORBHIP_REQUIRE(TF.valid, ORBHIP_EINVAL, "img == NULL needs the frame of an earlier orbt_* call of this thread on the device");
    ORBHIP_REQUIRE(TF.made_by(ctx) && TF.nlevels == nlevels && TF.icap == icap, ORBHIP_EINVAL, "ctx is not the extractor that produced the resident frame");
  }
  if (rc || (rc = W.commit(in))) return rc;
  if (img) {
    ORBHIP_REQUIRE(kps_out && desc_out, ORBHIP_EINVAL, "NULL output");
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~(size_t)255; return at; };
    const size_t oCnt = take(4), oKps = take((size_t)icap * sizeof(orbx_keypoint)), oDesc = take((size_t)icap * 32);
    TF.valid = false;
    if ((rc = TF.blk.ensure(o)) || (rc = TF.kps4.ensure(16 * (size_t)icap)) || (rc = TF.grid.off.ensure((size_t)(TRK_NCELL + 1) * 4)) || (rc = TF.grid.idx.ensure((size_t)icap * 4))) return rc;
    TF.oCnt = oCnt; TF.oKps = oKps; TF.oDesc = oDesc; TF.icap = icap; TF.nlevels = nlevels; std::memcpy(TF.bounds, bounds, 16);
    TF.grid.min_x = bounds[0]; TF.grid.min_y = bounds[2];
    TF.grid.winv = static_cast<float>(FRAME_GRID_COLS) / (bounds[1] - bounds[0]); TF.grid.hinv = static_cast<float>(FRAME_GRID_ROWS) / (bounds[3] - bounds[2]);
    uint8_t* fb = TF.blk.as<uint8_t>();
    if ((rc = orbhip::orbx_extract_chained(ctx, d_img, w, h, stride, (orbx_keypoint*)(fb + oKps), fb + oDesc, icap, (int32_t*)(fb + oCnt), (void*)W.s))) return rc;
    float* d_dummy = W.d<float>(4, &rc); int32_t* d_di = W.d<int32_t>(4, &rc); uint8_t* d_db = W.d<uint8_t>(4, &rc); uint32_t* d_tot = W.d<uint32_t>(1, &rc);
    if (rc) return rc;
    hipLaunchKernelGGL(k_trk_prepare, dim3(1), dim3(1024), 0, W.s, in.dev<TrkIn>(pP), (const double*)nullptr, (const int32_t*)nullptr, (const uint8_t*)nullptr,
                       (const orbx_keypoint*)(fb + oKps), (const int32_t*)(fb + oCnt), icap, d_dummy, d_dummy, d_di, d_di, d_db, TF.kps4.as<float>(),
                       TF.grid.off.as<uint32_t>(), TF.grid.idx.as<uint32_t>(), 0, d_tot);
  }
  const uint8_t* fblk = TF.blk.as<uint8_t>();
  const float* d_kps4 = TF.kps4.as<float>();
  const int32_t* d_count = (const int32_t*)(fblk + TF.oCnt);
  const uint8_t* d_fdesc = fblk + TF.oDesc;
  const int fcap = TF.icap;
  int max_kf = 1;
  for (int i = 0; i < n_cand; i++) max_kf = std::max(max_kf, cand[i].n);
  // output block: [word | weight | node | per candidate {TrkOut, owner[fcap]}]; scratch shared by the candidates (they run one after the other)
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~(size_t)255; return at; };
  const size_t oWord = take(4 * (size_t)fcap), oWt = take(8 * (size_t)fcap), oNode = take(4 * (size_t)fcap);
  const size_t oOut = take(sizeof(TrkOut) * (size_t)std::max(n_cand, 1)), oOwner = take(4 * (size_t)fcap * std::max(n_cand, 1));
  uint8_t* dblk = W.d<uint8_t>(o, &rc);
  int32_t* d_match = W.d<int32_t>((size_t)max_kf, &rc); int32_t* d_bin = W.d<int32_t>(fcap, &rc); int* d_hist = W.d<int>(TRK_HISTO + 2, &rc);
  int32_t* d_feat = W.d<int32_t>(fcap, &rc); double* d_oX = W.d<double>(3 * (size_t)fcap, &rc); double* d_ouv = W.d<double>(2 * (size_t)fcap, &rc); float* d_ow = W.d<float>(fcap, &rc);
  int32_t* d_ooff = W.d<int32_t>(2, &rc); double* d_K4 = W.d<double>(4, &rc); double* d_pose = W.d<double>(7, &rc);
  double* d_kfX = W.d<double>(3 * (size_t)max_kf, &rc);        // (k_trf_finish gathers PoseOptimization's points: not used here, any readable memory)
  if (rc) return rc;
  ORBHIP_CHECK_HIP(hipMemsetAsync(d_kfX, 0, 24 * (size_t)max_kf, W.s));
  // Frame::ComputeBoW (src/Frame.cc:322-327: levelsup 4)
  if ((rc = orbv_descend_device(voc, d_fdesc, fcap, 4, (int32_t*)(dblk + oWord), (double*)(dblk + oWt), (uint32_t*)(dblk + oNode), (void*)W.s))) return rc;
  for (int i = 0; i < n_cand; i++) {
    const orbt_reloc_keyframe& q = cand[i];
    const TrfIn* dI = in.dev<TrfIn>(pI) + i;
    int32_t* d_owner = (int32_t*)(dblk + oOwner) + (size_t)i * fcap;
    hipLaunchKernelGGL(k_trf_init, dim3((std::max(std::max(q.n, fcap), 64) + 255) / 256), dim3(256), 0, W.s, d_match, q.n, d_owner, d_bin, fcap, d_hist);
    if (q.fv_n > 0)
      hipLaunchKernelGGL(k_trf_bow, dim3(q.fv_n), dim3(64), 0, W.s, dI, in.dev<uint32_t>(pc[i].fn), in.dev<uint32_t>(pc[i].fo), in.dev<uint32_t>(pc[i].fi), in.dev<uint8_t>(pc[i].d),
                         in.dev<uint8_t>(pc[i].v), in.dev<float>(pc[i].a), d_fdesc, (const uint32_t*)(dblk + oNode), (const double*)(dblk + oWt), d_kps4, d_count, fcap,
                         d_match, d_owner, d_bin, d_hist);
    hipLaunchKernelGGL(k_trf_finish, dim3(1), dim3(1024), 0, W.s, dI, d_hist, d_kps4, d_count, fcap, (const double*)d_kfX, d_match, d_owner, d_bin, d_feat, d_oX, d_ouv, d_ow, d_ooff,
                       d_pose, d_K4, (TrkOut*)(dblk + oOut) + i);
  }
  ORBHIP_CHECK_HIP(hipGetLastError());
  const uint8_t* hb = W.down(dblk, o, &rc);
  const uint8_t* hf = img ? W.down(fblk, TF.oDesc + (size_t)fcap * 32, &rc) : W.down(fblk, 256, &rc);
  if (rc || (rc = W.sync())) return rc;
  const int n_dev = *(const int32_t*)(hf + TF.oCnt);
  if (n_dev < 0) { set_error("extractor capacity exceeded"); return ORBHIP_EOVERFLOW; }
  const int n = std::min(n_dev, fcap);
  if (img) { TF.n_kp = n; TF.valid = true; TF.producer = ctx; TF.producer_gen = orbhip::orbx_ctx_generation(ctx); std::memcpy(kps_out, hf + TF.oKps, (size_t)n * sizeof(orbx_keypoint)); std::memcpy(desc_out, hf + TF.oDesc, (size_t)n * 32); }
  if (n != TF.n_kp) { set_error("resident frame changed"); return ORBHIP_EINVAL; }
  *n_keypoints = n;
  if (bow_word && bow_value && fv_node && fv_idx)
    orbhip::orbv_merge_host((const int32_t*)(hb + oWord), (const double*)(hb + oWt), (const uint32_t*)(hb + oNode), n, bow_word, bow_value, n_words, fv_node, fv_off, fv_idx, n_fv_nodes);
  else { *n_words = 0; *n_fv_nodes = 0; fv_off[0] = 0; }
  for (int i = 0; i < n_cand; i++) {
    const TrkOut* T = (const TrkOut*)(hb + oOut) + i;
    nmatches[i] = T->nmatches;
    std::memcpy(slot_owner + (size_t)i * cap, hb + oOwner + 4 * (size_t)i * fcap, 4 * (size_t)n);
    for (int k = n; k < cap; k++) slot_owner[(size_t)i * cap + k] = -1;
  }
  return 0;
}

}  // extern "C"
