// orb_localmap.hip -- the LocalMapping thread's data-parallel steps, device-resident across the neighbour keyframes (round 5).
//
//   orbl_create_new_map_points   LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:196-396): for every neighbour keyframe, in
//                                order, ORBmatcher::SearchForTriangulation(current, neighbour) (src/ORBmatcher.cc:582-722, the
//                                matcher of that call site: ORBmatcher(0.6, false), :203) + the per-match triangulation and its
//                                gates (:267-378).  ONE call: one upload, one kernel per neighbour on one stream, one download.
//   orbl_fuse_batch              LocalMapping::SearchInNeighbors (:398-505): the candidate selection of ORBmatcher::Fuse
//                                (src/ORBmatcher.cc:724-842) for ALL target keyframes of the first loop (:441) in one call.
//
// The neighbours of CreateNewMapPoints are NOT independent: a triangulated match gives keypoint idx1 of the current keyframe a map
// point (current_keyframe_->AddMapPoint, :383), and SearchForTriangulation skips keypoints that hold one (src/ORBmatcher.cc:621-623) -
// neighbour i + 1 must see the mask neighbour i left.  The mask lives on the device; the kernels of consecutive neighbours are
// ordered by the stream.  The map mutation (new MapPoint, AddObservation, AddMapPoint ... :380-393) stays in the caller's loop
// over the accepted (idx1, idx2, x3D) of each neighbour (csrc/compat/orbslam_dropin.h).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/orbslam_hip.h"
#include "common.h"
#include "orb_frame.h"
#include "tri_math.h"

namespace orbhip {

#define LM_TH_LOW 50
#define LM_QPB 16                 /* queries (keypoints of the current keyframe) per workgroup: 16 lanes each */

// one neighbour, device view (offsets into the call's packed input block)
struct LmNb {
  size_t kps, desc, unmapped, fv_node, fv_off, fv_idx;
  int n, fv_n;
  TriCam C;                       // T1 / K1 / Ow1 = the current keyframe, T2 / K2 / Ow2 = this neighbour, ratio_factor
  double F12[9]; float ex, ey;
};

__device__ __forceinline__ int lm_hamming(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b) {
  int d = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) d += __popc(a[k] ^ b[k]);
  return d;
}

// node id of every keypoint of the current keyframe (a keypoint belongs to exactly one node of the FeatureVector)
__global__ __launch_bounds__(256) void k_lm_node_of(const uint32_t* __restrict__ fv_node, const uint32_t* __restrict__ fv_off, const uint32_t* __restrict__ fv_idx, int fv_n,
                                                    int n1, uint32_t* __restrict__ node_of) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= fv_n) return;
  const uint32_t id = fv_node[m];
  for (uint32_t e = fv_off[m]; e < fv_off[m + 1]; e++) { const uint32_t i = fv_idx[e]; if (i < (uint32_t)n1) node_of[i] = id; }
}

// CheckNewKeyFrames between two neighbours (src/LocalMapping.cc:227): ONE decision per neighbour, taken by one thread before the
// neighbour's kernel starts - every workgroup of that kernel then sees the same answer.  state[0] = latch (1: stopped), state[1] =
// neighbours processed.
__global__ void k_lm_gate(const volatile unsigned char* __restrict__ stop, int k, int* __restrict__ state) {
  if (state[0]) return;
  if (k > 0 && stop && *stop) { state[0] = 1; return; }
  state[1] = k + 1;
}

// neighbour k: SearchForTriangulation + triangulation of the current keyframe's keypoints; 16 lanes per keypoint
__global__ __launch_bounds__(16 * LM_QPB) void k_lm_neighbour(const uint8_t* __restrict__ base, const LmNb* __restrict__ nbs, int k, const float* __restrict__ kps1,
                                                               const uint32_t* __restrict__ desc1, uint8_t* __restrict__ mask1, const uint32_t* __restrict__ node_of, int n1,
                                                               const float* __restrict__ scale_factors, const float* __restrict__ level_sigma2, const int* __restrict__ state,
                                                               int32_t* __restrict__ match12, uint8_t* __restrict__ ok, double* __restrict__ x3D) {
  const int tid = threadIdx.x, sub = tid & 15;
  const int i1 = blockIdx.x * LM_QPB + (tid >> 4);
  if (i1 >= n1) return;
  int32_t* m_out = match12 + (size_t)k * n1; uint8_t* ok_out = ok + (size_t)k * n1; double* x_out = x3D + 3 * (size_t)k * n1;
  if (state[0]) {                                               // stopped before this neighbour: nothing of it is produced
    if (sub == 0) { m_out[i1] = -1; ok_out[i1] = 0; x_out[3 * (size_t)i1] = 0.0; x_out[3 * (size_t)i1 + 1] = 0.0; x_out[3 * (size_t)i1 + 2] = 0.0; }
    return;
  }
  const LmNb& N = nbs[k];
  const float* kps2 = (const float*)(base + N.kps);
  const uint32_t* desc2 = (const uint32_t*)(base + N.desc);
  const uint8_t* um2 = (const uint8_t*)(base + N.unmapped);
  const uint32_t* fvn = (const uint32_t*)(base + N.fv_node);
  const uint32_t* fvo = (const uint32_t*)(base + N.fv_off);
  const uint32_t* fvi = (const uint32_t*)(base + N.fv_idx);
  unsigned long long best = ~0ull;                              // (distance << 32) | (0xFFFFFFFF - position in the node's list): least distance, LATEST position (:654 `dist > bestDist`)
  uint32_t lo = 0, hi = 0;
  const bool live = mask1[i1] != 0;                             // keypoints that already hold a MapPoint are skipped (:621-623)
  const float x1 = kps1[4 * (size_t)i1], y1 = kps1[4 * (size_t)i1 + 1];
  if (live && N.fv_n > 0) {
    const uint32_t id = node_of[i1];
    int a = 0, b = N.fv_n - 1, m = -1;                          // the neighbour's list of the same vocabulary node (ascending node ids)
    while (a <= b) { const int c = (a + b) >> 1; const uint32_t v = fvn[c]; if (v == id) { m = c; break; } if (v < id) a = c + 1; else b = c - 1; }
    if (m >= 0) { lo = fvo[m]; hi = fvo[m + 1]; }
  }
  if (hi > lo) {
    uint32_t d1[8];
#pragma unroll
    for (int q = 0; q < 8; q++) d1[q] = desc1[8 * (size_t)i1 + q];
    // CheckDistEpipolarLine (:128-149): l = x1' F12; float keypoint x double F12, narrowed to float
    const float la = (float)(x1 * N.F12[0] + y1 * N.F12[3] + N.F12[6]);
    const float lb = (float)(x1 * N.F12[1] + y1 * N.F12[4] + N.F12[7]);
    const float lc = (float)(x1 * N.F12[2] + y1 * N.F12[5] + N.F12[8]);
    const float den = la * la + lb * lb;
    for (uint32_t e = lo + sub; e < hi; e += 16) {
      const uint32_t i2 = fvi[e];
      if (i2 >= (uint32_t)N.n || !um2[i2]) continue;            // (vbMatched2 is never set in this fork; pMP2 != NULL skips, :637)
      const int dist = lm_hamming(d1, desc2 + 8 * (size_t)i2);
      if (dist > LM_TH_LOW) continue;
      const float x2 = kps2[4 * (size_t)i2], y2 = kps2[4 * (size_t)i2 + 1];
      const int o2 = (int)kps2[4 * (size_t)i2 + 2];
      const float dex = N.ex - x2, dey = N.ey - y2;
      if (dex * dex + dey * dey < 100 * scale_factors[o2]) continue;         // too close to the epipole (:658-664)
      if (den == 0) continue;
      const float num = la * x2 + lb * y2 + lc;
      const float dsqr = num * num / den;
      if (!((double)dsqr < 3.84 * (double)level_sigma2[o2])) continue;
      const unsigned long long key = ((unsigned long long)(unsigned)dist << 32) | (unsigned long long)(0xFFFFFFFFu - (e - lo));
      best = key < best ? key : best;
    }
  }
  // least key of the 16 lanes of the keypoint
#pragma unroll
  for (int s = 8; s >= 1; s >>= 1) {
    const unsigned long long o = __shfl_xor(best, s, 16);
    best = o < best ? o : best;
  }
  if (sub != 0) return;
  int32_t m2 = -1; bool good = false; double X[3] = {0.0, 0.0, 0.0};
  if (best != ~0ull) {
    const uint32_t pos = 0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFull);
    m2 = (int32_t)fvi[lo + pos];
    const int o1 = (int)kps1[4 * (size_t)i1 + 2], o2 = (int)kps2[4 * (size_t)m2 + 2];
    good = triangulate_one(N.C, x1, y1, o1, kps2[4 * (size_t)m2], kps2[4 * (size_t)m2 + 1], o2, level_sigma2, scale_factors, X);
  }
  m_out[i1] = m2; ok_out[i1] = good ? 1 : 0;
  x_out[3 * (size_t)i1] = good ? X[0] : 0.0; x_out[3 * (size_t)i1 + 1] = good ? X[1] : 0.0; x_out[3 * (size_t)i1 + 2] = good ? X[2] : 0.0;
  if (good) mask1[i1] = 0;                                      // current_keyframe_->AddMapPoint(map_point, idx1) (:383): the next neighbour skips it
}

// ---- Fuse, candidate selection for T target keyframes x M map points (src/ORBmatcher.cc:724-842) --------------------------------
struct LfKf { size_t kps, desc, grid_off, grid_idx; int n; float min_x, min_y, winv, hinv; size_t q; };   // q: offset of this keyframe's [M] query records

struct alignas(8) LfQuery { float u, v, radius; int32_t level; };       // level < 0: the map point is not projected into this keyframe (a gate failed on the host side)

// one wave per (keyframe, map point): KeyFrame::GetFeaturesInArea(u, v, r) over the keyframe's grid (src/KeyFrame.cc:575-622: cells in
// (ix, iy) order, a cell's features in list order), level in [pred - 1, pred], chi-square gate e2 * inv_level_sigma2 <= 5.99, least
// descriptor distance, the FIRST of equal ones in that order (:799 `dist < bestDist`)
// chi2_gate = 0: the Sim(3) form of LoopClosing::SearchAndFuse (src/ORBmatcher.cc:844-954), which has no reprojection gate
__global__ __launch_bounds__(256) void k_lf_select(const uint8_t* __restrict__ base, const LfKf* __restrict__ kfs, int M, const uint32_t* __restrict__ mp_desc,
                                                   const float* __restrict__ inv_level_sigma2, int chi2_gate, int32_t* __restrict__ best_idx, int32_t* __restrict__ best_dist) {
  const int t = blockIdx.y, lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= M) return;
  const LfKf& F = kfs[t];
  const LfQuery Q = ((const LfQuery*)(base + F.q))[q];
  unsigned long long mine = ~0ull;                              // (distance << 32) | position in GetFeaturesInArea's output
  uint32_t mine_idx = 0;
  if (Q.level >= 0 && F.n > 0) {
    const float* kps = (const float*)(base + F.kps);
    const uint32_t* desc = (const uint32_t*)(base + F.desc);
    const uint32_t* goff = (const uint32_t*)(base + F.grid_off);
    const uint32_t* gidx = (const uint32_t*)(base + F.grid_idx);
    uint32_t d1[8];
#pragma unroll
    for (int k = 0; k < 8; k++) d1[k] = mp_desc[8 * (size_t)q + k];
    const float r = Q.radius;
    const int cx0 = max(0, (int)floorf((Q.u - F.min_x - r) * F.winv)), cx1 = min(FRAME_GRID_COLS - 1, (int)ceilf((Q.u - F.min_x + r) * F.winv));
    const int cy0 = max(0, (int)floorf((Q.v - F.min_y - r) * F.hinv)), cy1 = min(FRAME_GRID_ROWS - 1, (int)ceilf((Q.v - F.min_y + r) * F.hinv));
    if (cx0 < FRAME_GRID_COLS && cx1 >= 0 && cy0 < FRAME_GRID_ROWS && cy1 >= 0) {
      uint32_t order = 0;                                       // features GetFeaturesInArea has returned so far (the same on every lane)
      for (int ix = cx0; ix <= cx1; ix++)
        for (int iy = cy0; iy <= cy1; iy++) {
          const int cell = ix * FRAME_GRID_ROWS + iy;
          const uint32_t lo = goff[cell], hi = goff[cell + 1];
          for (uint32_t e0 = lo; e0 < hi; e0 += 64) {
            const uint32_t e = e0 + lane;
            bool in = false; uint32_t idx = 0; float kx = 0.f, ky = 0.f; int lvl = 0;
            if (e < hi) {
              idx = gidx[e];
              const float4 k = ((const float4*)kps)[idx];
              kx = k.x; ky = k.y; lvl = (int)k.z;
              in = fabsf(kx - Q.u) < r && fabsf(ky - Q.v) < r;  // (src/KeyFrame.cc:607-611)
            }
            const unsigned long long mask = __ballot(in);
            if (in && !(lvl < Q.level - 1 || lvl > Q.level)) {   // (:781)
              const float ex = Q.u - kx, ey = Q.v - ky;
              const float e2 = ex * ex + ey * ey;
              if (!chi2_gate || !((double)(e2 * inv_level_sigma2[lvl]) > 5.99)) {     // (:789)
                const unsigned long long key = ((unsigned long long)(unsigned)lm_hamming(d1, desc + 8 * (size_t)idx) << 32) |
                                               (unsigned long long)(order + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull)));
                if (key < mine) { mine = key; mine_idx = idx; }
              }
            }
            order += (uint32_t)__popcll(mask);
          }
        }
    }
  }
  unsigned long long best = mine;
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const unsigned long long o = __shfl_xor(best, s, 64);
    best = o < best ? o : best;
  }
  int32_t widx = -1;
  if (best != ~0ull) {                                          // (positions are unique: exactly one lane holds the winner)
    const unsigned long long who = __ballot(mine == best);
    widx = (int32_t)__shfl((int)mine_idx, __ffsll((long long)who) - 1, 64);
  }
  if (lane == 0) { best_idx[(size_t)t * M + q] = widx; best_dist[(size_t)t * M + q] = best == ~0ull ? 256 : (int32_t)(best >> 32); }
}

}  // namespace orbhip

using namespace orbhip;

// a FeatureVector as CSR: fv_off[0] = 0, non-decreasing; fv_off[fv_n] is the length of fv_idx (the kernels walk fv_idx[fv_off[m] .. fv_off[m + 1]))
static int lm_check_feature_vector(const uint32_t* fv_off, int fv_n, const char* who) {
  if (fv_n == 0) return 0;
  bool ok = fv_off[0] == 0 && fv_off[fv_n] <= 0x7fffffffu;
  for (int m = 0; m < fv_n && ok; m++) ok = fv_off[m] <= fv_off[m + 1];
  if (!ok) { set_error("%s: FeatureVector offsets must start at 0 and must not decrease", who); return ORBHIP_EINVAL; }
  return 0;
}

extern "C" {

int orbl_create_new_map_points(const float* kps1, const uint8_t* desc1, const uint8_t* unmapped1, int n1, const uint32_t* fv1_node, const uint32_t* fv1_off,
                               const uint32_t* fv1_idx, int fv1_n, const double* Tcw1, const float* K1, const orbl_keyframe* nb, int n_nb,
                               const float* scale_factors, const float* level_sigma2, int n_levels, float ratio_factor, const volatile uint8_t* stop,
                               int32_t* match12, uint8_t* ok, double* x3D, int* n_processed) {
  ORBHIP_REQUIRE(n1 >= 0 && n_nb >= 0 && fv1_n >= 0 && n_levels > 0 && n_levels <= 64, ORBHIP_EINVAL, "bad size");
  ORBHIP_REQUIRE(Tcw1 && K1 && scale_factors && level_sigma2 && (n_nb == 0 || nb), ORBHIP_EINVAL, "NULL argument");
  if (n_processed) *n_processed = 0;
  if (n_nb == 0 || n1 == 0) { if (n_processed) *n_processed = n_nb; return 0; }
  ORBHIP_REQUIRE(kps1 && desc1 && match12 && ok && x3D && (fv1_n == 0 || (fv1_node && fv1_off && fv1_idx)), ORBHIP_EINVAL, "NULL argument");
  for (int i = 0; i < n1; i++) ORBHIP_REQUIRE(kps1[4 * (size_t)i + 2] >= 0 && kps1[4 * (size_t)i + 2] < n_levels, ORBHIP_EINVAL, "octave out of range");
  for (int k = 0; k < n_nb; k++) {
    const orbl_keyframe& q = nb[k];
    ORBHIP_REQUIRE(q.n >= 0 && q.fv_n >= 0 && (q.n == 0 || (q.kps && q.desc)) && (q.fv_n == 0 || (q.fv_node && q.fv_off && q.fv_idx)), ORBHIP_EINVAL, "NULL neighbour argument");
    for (int i = 0; i < q.n; i++) ORBHIP_REQUIRE(q.kps[4 * (size_t)i + 2] >= 0 && q.kps[4 * (size_t)i + 2] < n_levels, ORBHIP_EINVAL, "octave out of range");
    if (int r = lm_check_feature_vector(q.fv_off, q.fv_n, "neighbour")) return r;
  }
  if (int r = lm_check_feature_vector(fv1_off, fv1_n, "current keyframe")) return r;
  ThreadWs& W = thread_ws();
  int rc = W.begin();
  if (rc) return rc;
  std::vector<uint8_t> all1;                                     // unmapped1 == NULL: every keypoint is a candidate
  if (!unmapped1) { all1.assign((size_t)n1, 1); unmapped1 = all1.data(); }
  ThreadWs::Pack in;
  const int pK = in.add(kps1, 16 * (size_t)n1), pD = in.add(desc1, 32 * (size_t)n1), pU = in.add(unmapped1, (size_t)n1);
  const int pFn = in.add(fv1_node, 4 * (size_t)fv1_n), pFo = in.add(fv1_off, 4 * ((size_t)fv1_n + 1)), pFi = in.add(fv1_idx, fv1_n ? 4 * (size_t)fv1_off[fv1_n] : 0);
  const int pS = in.add(scale_factors, 4 * (size_t)n_levels), pL = in.add(level_sigma2, 4 * (size_t)n_levels);
  std::vector<LmNb> h_nb((size_t)n_nb);
  std::vector<std::vector<uint8_t>> all2((size_t)n_nb);
  for (int k = 0; k < n_nb; k++) {
    const orbl_keyframe& q = nb[k];
    LmNb& N = h_nb[k];
    std::memset(&N, 0, sizeof(N));
    const uint8_t* um = q.unmapped;
    if (!um) { all2[k].assign((size_t)std::max(q.n, 1), 1); um = all2[k].data(); }
    const size_t nidx = q.fv_n ? (size_t)q.fv_off[q.fv_n] : 0;
    auto off_of = [&](int piece) { return in.pieces[piece].off; };
    N.kps = off_of(in.add(q.kps, 16 * (size_t)q.n)); N.desc = off_of(in.add(q.desc, 32 * (size_t)q.n)); N.unmapped = off_of(in.add(um, (size_t)q.n));
    N.fv_node = off_of(in.add(q.fv_node, 4 * (size_t)q.fv_n)); N.fv_off = off_of(in.add(q.fv_off, 4 * ((size_t)q.fv_n + 1))); N.fv_idx = off_of(in.add(q.fv_idx, 4 * nidx));
    N.n = q.n; N.fv_n = q.fv_n;
    for (int j = 0; j < 12; j++) { N.C.T1[j] = Tcw1[j]; N.C.T2[j] = q.Tcw[j]; }
    for (int j = 0; j < 3; j++) {                                // camera centres Ow = -Rcw^T tcw (KeyFrame::SetPose)
      N.C.Ow1[j] = -(Tcw1[j] * Tcw1[3] + Tcw1[4 + j] * Tcw1[7] + Tcw1[8 + j] * Tcw1[11]);
      N.C.Ow2[j] = -(q.Tcw[j] * q.Tcw[3] + q.Tcw[4 + j] * q.Tcw[7] + q.Tcw[8 + j] * q.Tcw[11]);
    }
    for (int j = 0; j < 4; j++) { N.C.K1[j] = K1[j]; N.C.K2[j] = q.K4[j]; }
    N.C.ratio_factor = ratio_factor;
    for (int j = 0; j < 9; j++) N.F12[j] = q.F12[j];
    N.ex = q.ex; N.ey = q.ey;
  }
  const int pN = in.add(h_nb.data(), sizeof(LmNb) * (size_t)n_nb);
  // outputs in one block: [state (2 ints) | match12 | ok | x3D]
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~(size_t)255; return at; };
  const size_t oState = take(8), oM = take(4 * (size_t)n_nb * n1), oOk = take((size_t)n_nb * n1), oX = take(24 * (size_t)n_nb * n1);
  uint8_t* dblk = W.d<uint8_t>(o, &rc);
  uint32_t* d_node_of = W.d<uint32_t>((size_t)n1, &rc);
  unsigned char* h_stop = W.h<unsigned char>(16, &rc);            // pinned, device-visible mirror of the caller's flag
  if (rc || (rc = W.commit(in))) return rc;
  // the mirror starts from the caller's flag: a flag that is already up when the call begins stops the chain after neighbour 0, every time
  // (src/LocalMapping.cc:227 `if (i > 0 && CheckNewKeyFrames()) return;` - ADVICE r5: it used to start at 0 and be raised only after the enqueues)
  *(volatile unsigned char*)h_stop = (stop && *stop) ? 1 : 0;
  hipEvent_t done = nullptr;
  if (stop) ORBHIP_CHECK_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
  ORBHIP_CHECK_HIP(hipMemsetAsync(dblk + oState, 0, 8, W.s));
  ORBHIP_CHECK_HIP(hipMemsetAsync(d_node_of, 0xFF, 4 * (size_t)n1, W.s));
  if (fv1_n) hipLaunchKernelGGL(k_lm_node_of, dim3((fv1_n + 255) / 256), dim3(256), 0, W.s, in.dev<uint32_t>(pFn), in.dev<uint32_t>(pFo), in.dev<uint32_t>(pFi), fv1_n, n1, d_node_of);
  for (int k = 0; k < n_nb; k++) {
    hipLaunchKernelGGL(k_lm_gate, dim3(1), dim3(1), 0, W.s, stop ? (const volatile unsigned char*)h_stop : (const volatile unsigned char*)nullptr, k, (int*)(dblk + oState));
    hipLaunchKernelGGL(k_lm_neighbour, dim3((n1 + LM_QPB - 1) / LM_QPB), dim3(16 * LM_QPB), 0, W.s, (const uint8_t*)in.dbase, in.dev<LmNb>(pN), k, in.dev<float>(pK),
                       in.dev<uint32_t>(pD), in.dev<uint8_t>(pU), d_node_of, n1, in.dev<float>(pS), in.dev<float>(pL), (const int*)(dblk + oState),
                       (int32_t*)(dblk + oM), dblk + oOk, (double*)(dblk + oX));
    if (stop && *stop) *(volatile unsigned char*)h_stop = 1;     // (between the enqueues too: the gates of the neighbours behind see it)
  }
  if (hipGetLastError() != hipSuccess) { if (done) (void)hipEventDestroy(done); set_error("orbl_create_new_map_points: a launch failed"); return ORBHIP_ENODEV; }
  const uint8_t* hb = W.down(dblk, o, &rc);
  if (rc) { if (done) (void)hipEventDestroy(done); return rc; }
  if (stop) {                                                    // forward the caller's flag while the chain runs (the kernels read the pinned mirror)
    (void)hipEventRecord(done, W.s);
    while (hipEventQuery(done) == hipErrorNotReady) { if (*stop) *(volatile unsigned char*)h_stop = 1; }
    (void)hipEventDestroy(done);
  }
  if ((rc = W.sync())) return rc;
  const int* st = (const int*)(hb + oState);
  if (n_processed) *n_processed = st[1];
  std::memcpy(match12, hb + oM, 4 * (size_t)n_nb * n1);
  std::memcpy(ok, hb + oOk, (size_t)n_nb * n1);
  std::memcpy(x3D, hb + oX, 24 * (size_t)n_nb * n1);
  return 0;
}

static int fuse_batch_impl(const orbl_fuse_keyframe* kf, int n_kf, const float* q_uv, const float* q_radius, const int32_t* q_level, int n_mp, const uint8_t* mp_desc,
                           const float* inv_level_sigma2, int n_levels, int chi2_gate, int32_t* best_idx, int32_t* best_dist) {
  ORBHIP_REQUIRE(n_kf >= 0 && n_mp >= 0 && n_levels > 0, ORBHIP_EINVAL, "bad size");
  if (n_kf == 0 || n_mp == 0) return 0;
  ORBHIP_REQUIRE(kf && q_uv && q_radius && q_level && mp_desc && (inv_level_sigma2 || !chi2_gate) && best_idx && best_dist, ORBHIP_EINVAL, "NULL argument");
  std::vector<float> no_gate;
  if (!inv_level_sigma2) { no_gate.assign((size_t)n_levels, 0.f); inv_level_sigma2 = no_gate.data(); }
  ThreadWs& W = thread_ws();
  int rc = W.begin();
  if (rc) return rc;
  ThreadWs::Pack in;
  const int pD = in.add(mp_desc, 32 * (size_t)n_mp), pS = in.add(inv_level_sigma2, 4 * (size_t)n_levels);
  std::vector<LfKf> h_kf((size_t)n_kf);
  std::vector<std::vector<LfQuery>> hq((size_t)n_kf);
  std::vector<std::vector<uint32_t>> goff((size_t)n_kf), gidx((size_t)n_kf);
  const int NC = FRAME_GRID_COLS * FRAME_GRID_ROWS;
  for (int t = 0; t < n_kf; t++) {
    const orbl_fuse_keyframe& q = kf[t];
    ORBHIP_REQUIRE(q.n >= 0 && (q.n == 0 || (q.kps && q.desc)), ORBHIP_EINVAL, "NULL keyframe argument");
    LfKf& F = h_kf[t];
    std::memset(&F, 0, sizeof(F));
    F.n = q.n; F.min_x = q.bounds[0]; F.min_y = q.bounds[2];
    F.winv = static_cast<float>(FRAME_GRID_COLS) / (q.bounds[1] - q.bounds[0]); F.hinv = static_cast<float>(FRAME_GRID_ROWS) / (q.bounds[3] - q.bounds[2]);
    // the keyframe's grid (Frame::AssignFeaturesToGrid + PosInGrid, src/Frame.cc:158-173, 309-320): cells in (x, y) order, features in index order
    std::vector<int> cell_of((size_t)q.n, -1);
    goff[t].assign((size_t)NC + 1, 0);
    for (int i = 0; i < q.n; i++) {
      const float x = q.kps[4 * (size_t)i], y = q.kps[4 * (size_t)i + 1];
      const int px = (int)std::round((x - F.min_x) * F.winv), py = (int)std::round((y - F.min_y) * F.hinv);
      if (px < 0 || px >= FRAME_GRID_COLS || py < 0 || py >= FRAME_GRID_ROWS) continue;
      cell_of[i] = px * FRAME_GRID_ROWS + py; goff[t][cell_of[i] + 1]++;
    }
    for (int c = 0; c < NC; c++) goff[t][c + 1] += goff[t][c];
    gidx[t].assign((size_t)std::max<uint32_t>(goff[t][NC], 1), 0);
    { std::vector<uint32_t> cur(goff[t].begin(), goff[t].end() - 1); for (int i = 0; i < q.n; i++) if (cell_of[i] >= 0) gidx[t][cur[cell_of[i]]++] = (uint32_t)i; }
    hq[t].resize((size_t)n_mp);
    for (int m = 0; m < n_mp; m++) {
      LfQuery& Q = hq[t][m];
      Q.u = q_uv[2 * ((size_t)t * n_mp + m)]; Q.v = q_uv[2 * ((size_t)t * n_mp + m) + 1]; Q.radius = q_radius[(size_t)t * n_mp + m]; Q.level = q_level[(size_t)t * n_mp + m];
      if (Q.level >= n_levels) Q.level = -1;
    }
    auto off_of = [&](int piece) { return in.pieces[piece].off; };
    F.kps = off_of(in.add(q.kps, 16 * (size_t)q.n)); F.desc = off_of(in.add(q.desc, 32 * (size_t)q.n));
    F.grid_off = off_of(in.add(goff[t].data(), 4 * ((size_t)NC + 1))); F.grid_idx = off_of(in.add(gidx[t].data(), 4 * gidx[t].size()));
    F.q = off_of(in.add(hq[t].data(), sizeof(LfQuery) * (size_t)n_mp));
  }
  const int pF = in.add(h_kf.data(), sizeof(LfKf) * (size_t)n_kf);
  int32_t* d_out = W.d<int32_t>(2 * (size_t)n_kf * n_mp, &rc);
  if (rc || (rc = W.commit(in))) return rc;
  hipLaunchKernelGGL(k_lf_select, dim3((n_mp + 3) / 4, n_kf), dim3(256), 0, W.s, (const uint8_t*)in.dbase, in.dev<LfKf>(pF), n_mp, in.dev<uint32_t>(pD), in.dev<float>(pS),
                     chi2_gate, d_out, d_out + (size_t)n_kf * n_mp);
  ORBHIP_CHECK_HIP(hipGetLastError());
  const int32_t* h = W.down(d_out, 2 * (size_t)n_kf * n_mp, &rc);
  if (rc || (rc = W.sync())) return rc;
  std::memcpy(best_idx, h, 4 * (size_t)n_kf * n_mp);
  std::memcpy(best_dist, h + (size_t)n_kf * n_mp, 4 * (size_t)n_kf * n_mp);
  return 0;
}

int orbl_fuse_batch(const orbl_fuse_keyframe* kf, int n_kf, const float* q_uv, const float* q_radius, const int32_t* q_level, int n_mp, const uint8_t* mp_desc,
                    const float* inv_level_sigma2, int n_levels, int32_t* best_idx, int32_t* best_dist) {
  return fuse_batch_impl(kf, n_kf, q_uv, q_radius, q_level, n_mp, mp_desc, inv_level_sigma2, n_levels, 1, best_idx, best_dist);
}

int orbl_fuse_batch_sim3(const orbl_fuse_keyframe* kf, int n_kf, const float* q_uv, const float* q_radius, const int32_t* q_level, int n_mp, const uint8_t* mp_desc,
                         int n_levels, int32_t* best_idx, int32_t* best_dist) {
  return fuse_batch_impl(kf, n_kf, q_uv, q_radius, q_level, n_mp, mp_desc, nullptr, n_levels, 0, best_idx, best_dist);
}

}  // extern "C"
