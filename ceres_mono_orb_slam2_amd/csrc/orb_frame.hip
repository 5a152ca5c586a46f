// ============================================================================
// orb_frame.hip -- the steps either side of extract -> match on the device:
//   Frame::UndistortKeyPoints      (src/Frame.cc:329-355, cv::undistortPoints underneath)
//   Frame::AssignFeaturesToGrid    (src/Frame.cc:158-173, PosInGrid :309-320), 64 x 48 grid
//   Frame::GetFeaturesInArea       (src/Frame.cc:243-307; KeyFrame::GetFeaturesInArea src/KeyFrame.cc:575-622 is the
//                                   same walk without the level test)
//   Frame::isInFrustum             (src/Frame.cc:191-241) + MapPoint::PredictScale (src/MapPoint.cc:406-420)
// so that projected map points -> window candidates -> Hamming distances chain on the GPU without the host building
// the grid and the candidate lists.  Candidate ORDER is the reference's (cells ix-major then iy, entries of a cell in
// keypoint-index order): first-minimum tie-breaking downstream depends on it.  Float / double arithmetic follows the
// reference expression by expression, un-contracted (the library is built with -ffp-contract=off).
// ============================================================================
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <vector>

#include "common.h"
#include "orb_frame.h"
#include "tri_math.h"

namespace orbhip {

// ---------------------------------------------------------------------------- undistort
// cv::undistortPoints(src, dst, K, dist, noArray(), K): 5 fixed-point iterations in double (OpenCV 2.4 / 3.2,
// SURVEY Appendix A), distortion (k1, k2, p1, p2, k3); result rounded to float like the CV_32F destination.
__global__ void k_undistort(const float* __restrict__ xy, int n, double fx, double fy, double cx, double cy, double k1, double k2,
                            double p1, double p2, double k3, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double ifx = 1. / fx, ify = 1. / fy;
  double x = ((double)xy[2 * i] - cx) * ifx, y = ((double)xy[2 * i + 1] - cy) * ify;
  const double x0 = x, y0 = y;
  for (int j = 0; j < 5; j++) {
    const double r2 = x * x + y * y;
    const double icdist = 1. / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
    const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
    const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  out[2 * i] = (float)(fx * x + cx);
  out[2 * i + 1] = (float)(fy * y + cy);
}

// ---------------------------------------------------------------------------- grid
// cell id of every keypoint (0xFFFF = outside, PosInGrid returns false) + per-cell counts
__global__ __launch_bounds__(256) void k_grid_cells(const float* __restrict__ kps4, int n, float min_x, float min_y, float winv,
                                                    float hinv, unsigned short* __restrict__ cellid, int* __restrict__ cell_cnt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int px = (int)roundf((kps4[4 * i] - min_x) * winv), py = (int)roundf((kps4[4 * i + 1] - min_y) * hinv);   // :310-311
  unsigned short id = 0xFFFF;
  if (!(px < 0 || px >= FRAME_GRID_COLS || py < 0 || py >= FRAME_GRID_ROWS)) {
    id = (unsigned short)(px * FRAME_GRID_ROWS + py);               // grid_[x][y]
    atomicAdd(&cell_cnt[id], 1);
  }
  cellid[i] = id;
}

// exclusive scan of a[0..n) -> off[0..n], one 1024-thread workgroup (n up to ~1e6, looped)
__global__ __launch_bounds__(1024) void k_excl_scan(const int* __restrict__ a, int n, uint32_t* __restrict__ off) {
  __shared__ int s_w[16];
  __shared__ int s_carry;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    const int v = i < n ? a[i] : 0;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { const int t = s_w[k]; if (k < w) woff += t; tot += t; }
    const int carry = s_carry;
    if (i < n) off[i] = (uint32_t)(carry + woff + inc - v);
    __syncthreads();
    if (tid == 0) s_carry = carry + tot;
    __syncthreads();
  }
  if (tid == 0) off[n] = (uint32_t)s_carry;
}

// one thread per cell walks the keypoints in index order (their cell ids staged through LDS) and appends its own:
// the lists come out exactly as the reference's push_back order, with no atomics on the output.
__global__ __launch_bounds__(256) void k_grid_fill(const unsigned short* __restrict__ cellid, int n, const uint32_t* __restrict__ cell_off,
                                                   uint32_t* __restrict__ cell_idx) {
  __shared__ unsigned short s_id[2048];
  const int c = blockIdx.x * 256 + threadIdx.x;
  uint32_t pos = c < FRAME_GRID_COLS * FRAME_GRID_ROWS ? cell_off[c] : 0u;
  const uint32_t end = c < FRAME_GRID_COLS * FRAME_GRID_ROWS ? cell_off[c + 1] : 0u;
  for (int base = 0; base < n; base += 2048) {
    const int m = min(2048, n - base);
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += 256) s_id[i] = cellid[base + i];
    __syncthreads();
    if (pos < end)
      for (int i = 0; i < m; i++)
        if (s_id[i] == (unsigned short)c) cell_idx[pos++] = (uint32_t)(base + i);
  }
}

// ---------------------------------------------------------------------------- GetFeaturesInArea
// One wave per query.  Lanes take the cells of the window in the reference's loop order (ix outer, iy inner), 64 at a
// time; FILL = false counts the hits, FILL = true writes them at cand_off[q] in (cell, entry) order.
template <bool FILL>
__global__ __launch_bounds__(256) void k_area(const float* __restrict__ kps4, const uint32_t* __restrict__ cell_off,
                                              const uint32_t* __restrict__ cell_idx, float min_x, float min_y, float winv, float hinv,
                                              const float* __restrict__ q_xy, const float* __restrict__ q_r,
                                              const int* __restrict__ q_minl, const int* __restrict__ q_maxl,
                                              const uint8_t* __restrict__ q_valid, int nq, int* __restrict__ q_cnt,
                                              const uint32_t* __restrict__ cand_off, uint32_t* __restrict__ cand_idx, uint32_t cap, int stride) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= nq) return;
  int total = 0;
  const bool valid = !q_valid || q_valid[q];
  if (valid) {
    const float x = q_xy[2 * q], y = q_xy[2 * q + 1], r = q_r[q];
    const int minLevel = q_minl ? q_minl[q] : -1, maxLevel = q_maxl ? q_maxl[q] : -1;
    const int min_cx = max(0, (int)floorf((x - min_x - r) * winv));
    const int max_cx = min(FRAME_GRID_COLS - 1, (int)ceilf((x - min_x + r) * winv));
    const int min_cy = max(0, (int)floorf((y - min_y - r) * hinv));
    const int max_cy = min(FRAME_GRID_ROWS - 1, (int)ceilf((y - min_y + r) * hinv));
    if (!(min_cx >= FRAME_GRID_COLS || max_cx < 0 || min_cy >= FRAME_GRID_ROWS || max_cy < 0) && max_cx >= min_cx && max_cy >= min_cy) {
      const bool check = (minLevel > 0) || (maxLevel >= 0);
      const int ny = max_cy - min_cy + 1, ncell = (max_cx - min_cx + 1) * ny;
      const uint32_t obase = FILL ? cand_off[q] : 0u;
      auto hit = [&](uint32_t j) {
        const float4 k = ((const float4*)kps4)[j];
        const int oct = (int)k.z;
        if (check) {
          if (oct < minLevel) return false;
          if (maxLevel >= 0 && oct > maxLevel) return false;
        }
        return fabsf(k.x - x) < r && fabsf(k.y - y) < r;
      };
      for (int w0 = 0; w0 < ncell; w0 += 64) {
        const int w = w0 + lane;
        uint32_t lo = 0, hi = 0;
        if (w < ncell) {
          const int ix = min_cx + w / ny, iy = min_cy + w % ny;
          const int c = ix * FRAME_GRID_ROWS + iy;
          lo = cell_off[c]; hi = cell_off[c + 1];
        }
        int mine = 0;
        for (uint32_t e = lo; e < hi; e++) mine += hit(cell_idx[e]) ? 1 : 0;
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (FILL) {
          uint32_t pos = obase + (uint32_t)(total + incl - mine);
          for (uint32_t e = lo; e < hi; e++) { const uint32_t j = cell_idx[e]; if (hit(j)) { if (pos < cap) cand_idx[(size_t)pos * stride] = j; pos++; } }
        }
        total += __shfl(incl, 63);
      }
    }
  }
  if (!FILL && lane == 0) q_cnt[q] = total;
}

// ---------------------------------------------------------------------------- isInFrustum
__global__ __launch_bounds__(256) void k_frustum(FrustumCam C, const double* __restrict__ P, const double* __restrict__ Pn,
                                                 const float* __restrict__ min_dist, const float* __restrict__ max_dist, int n,
                                                 uint8_t* __restrict__ in_view, float* __restrict__ uv, int* __restrict__ level,
                                                 float* __restrict__ view_cos, int invariance_bounds, float* __restrict__ dist_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float u, v, vc, dist; int nScale;
  const bool ok = frustum_eval(C, P + 3 * (size_t)i, Pn + 3 * (size_t)i, min_dist[i], max_dist[i], invariance_bounds, u, v, nScale, vc, dist);
  in_view[i] = ok ? 1 : 0;
  uv[2 * i] = u; uv[2 * i + 1] = v;
  level[i] = nScale;
  view_cos[i] = vc;
  if (dist_out) dist_out[i] = dist;
}

// ---------------------------------------------------------------------------- host side
int frame_grid_build(FrameGridDev& g, const float* d_kps4, int n, const float* bounds, hipStream_t s) {
  g.min_x = bounds[0]; g.min_y = bounds[2];
  g.winv = static_cast<float>(FRAME_GRID_COLS) / (bounds[1] - bounds[0]);      // src/Frame.cc:130-133
  g.hinv = static_cast<float>(FRAME_GRID_ROWS) / (bounds[3] - bounds[2]);
  const int NC = FRAME_GRID_COLS * FRAME_GRID_ROWS;
  int rc = 0;
  if ((rc = g.cellid.ensure((size_t)std::max(n, 1) * 2)) || (rc = g.cnt.ensure((size_t)NC * 4)) || (rc = g.off.ensure((size_t)(NC + 1) * 4)) ||
      (rc = g.idx.ensure((size_t)std::max(n, 1) * 4)))
    return rc;
  ORBHIP_CHECK_HIP(hipMemsetAsync(g.cnt.p, 0, (size_t)NC * 4, s));
  if (n > 0) hipLaunchKernelGGL(k_grid_cells, dim3((n + 255) / 256), dim3(256), 0, s, d_kps4, n, g.min_x, g.min_y, g.winv, g.hinv,
                                g.cellid.as<unsigned short>(), g.cnt.as<int>());
  hipLaunchKernelGGL(k_excl_scan, dim3(1), dim3(1024), 0, s, g.cnt.as<int>(), NC, g.off.as<uint32_t>());
  if (n > 0) hipLaunchKernelGGL(k_grid_fill, dim3(NC / 256), dim3(256), 0, s, g.cellid.as<unsigned short>(), n, g.off.as<uint32_t>(), g.idx.as<uint32_t>());
  ORBHIP_CHECK_HIP(hipGetLastError());
  g.n = n;
  return 0;
}

int frame_area_candidates(const FrameGridDev& g, const float* d_kps4, const float* d_q_xy, const float* d_q_r, const int* d_q_minl,
                          const int* d_q_maxl, const uint8_t* d_q_valid, int nq, DevBuf& cnt, DevBuf& cand_off, DevBuf& cand_idx,
                          uint32_t* total_out, hipStream_t s) {
  *total_out = 0;
  if (nq <= 0) return 0;
  int rc = 0;
  if ((rc = cnt.ensure((size_t)nq * 4)) || (rc = cand_off.ensure((size_t)(nq + 1) * 4))) return rc;
  hipLaunchKernelGGL(k_area<false>, dim3((nq + 3) / 4), dim3(256), 0, s, d_kps4, g.off.as<uint32_t>(), g.idx.as<uint32_t>(), g.min_x, g.min_y,
                     g.winv, g.hinv, d_q_xy, d_q_r, d_q_minl, d_q_maxl, d_q_valid, nq, cnt.as<int>(), (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, 1);
  hipLaunchKernelGGL(k_excl_scan, dim3(1), dim3(1024), 0, s, cnt.as<int>(), nq, cand_off.as<uint32_t>());
  uint32_t total = 0;
  ORBHIP_CHECK_HIP(hipMemcpyAsync(&total, cand_off.as<uint32_t>() + nq, 4, hipMemcpyDeviceToHost, s));
  ORBHIP_CHECK_HIP(hipStreamSynchronize(s));
  *total_out = total;
  if (total == 0) return 0;
  if ((rc = cand_idx.ensure((size_t)total * 4))) return rc;
  hipLaunchKernelGGL(k_area<true>, dim3((nq + 3) / 4), dim3(256), 0, s, d_kps4, g.off.as<uint32_t>(), g.idx.as<uint32_t>(), g.min_x, g.min_y,
                     g.winv, g.hinv, d_q_xy, d_q_r, d_q_minl, d_q_maxl, d_q_valid, nq, (int*)nullptr, cand_off.as<uint32_t>(), cand_idx.as<uint32_t>(), total, 1);
  ORBHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

int frame_scan_enqueue(const int* d_cnt, int n, uint32_t* d_off, hipStream_t s) {
  hipLaunchKernelGGL(k_excl_scan, dim3(1), dim3(1024), 0, s, d_cnt, n, d_off);
  ORBHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

int frame_area_candidates_enqueue(const FrameGridDev& g, const float* d_kps4, const float* d_q_xy, const float* d_q_r, const int* d_q_minl,
                                  const int* d_q_maxl, const uint8_t* d_q_valid, int nq, int* d_cnt, uint32_t* d_cand_off, uint32_t* d_cand_idx,
                                  uint32_t cap, int idx_stride, hipStream_t s) {
  if (nq <= 0) return 0;
  hipLaunchKernelGGL(k_area<false>, dim3((nq + 3) / 4), dim3(256), 0, s, d_kps4, g.off.as<uint32_t>(), g.idx.as<uint32_t>(), g.min_x, g.min_y,
                     g.winv, g.hinv, d_q_xy, d_q_r, d_q_minl, d_q_maxl, d_q_valid, nq, d_cnt, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, 1);
  hipLaunchKernelGGL(k_excl_scan, dim3(1), dim3(1024), 0, s, d_cnt, nq, d_cand_off);
  hipLaunchKernelGGL(k_area<true>, dim3((nq + 3) / 4), dim3(256), 0, s, d_kps4, g.off.as<uint32_t>(), g.idx.as<uint32_t>(), g.min_x, g.min_y,
                     g.winv, g.hinv, d_q_xy, d_q_r, d_q_minl, d_q_maxl, d_q_valid, nq, (int*)nullptr, d_cand_off, d_cand_idx, cap, idx_stride);
  ORBHIP_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace orbhip

using namespace orbhip;

#define FCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { set_error("%s failed: %s", #expr, hipGetErrorString(_e)); return ORBHIP_ENODEV; } } while (0)

static int frame_need_device() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (the HIP path has no CPU fallback)"); return ORBHIP_ENODEV; }
  return use_default_device();
}

extern "C" {

int orbm_undistort_keypoints(const float* xy, int n, const float* K4, const float* dist5, float* xy_out) {
  ORBHIP_REQUIRE(n >= 0 && K4 && dist5 && (n == 0 || (xy && xy_out)), ORBHIP_EINVAL, "NULL argument");
  if (n == 0) return 0;
  if (dist5[0] == 0.0f) {                                   // src/Frame.cc:330-333: k1 == 0 -> keypoints are taken as they are
    if (xy_out != xy) std::memcpy(xy_out, xy, sizeof(float) * 2 * (size_t)n);
    return 0;
  }
  if (int rc = frame_need_device()) return rc;
  DevBuf din, dout;
  int rc = 0;
  if ((rc = din.ensure((size_t)n * 8)) || (rc = dout.ensure((size_t)n * 8))) { din.release(); dout.release(); return rc; }
  FCHK(hipMemcpy(din.p, xy, (size_t)n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256), dim3(256), 0, 0, din.as<float>(), n, (double)K4[0], (double)K4[1], (double)K4[2],
                     (double)K4[3], (double)dist5[0], (double)dist5[1], (double)dist5[2], (double)dist5[3], (double)dist5[4], dout.as<float>());
  FCHK(hipGetLastError());
  FCHK(hipMemcpy(xy_out, dout.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  din.release(); dout.release();
  return 0;
}

static int is_in_frustum_impl(const double* Rcw, const double* tcw, const float* K4, const float* bounds, const double* P, const double* Pn,
                              const float* min_dist, const float* max_dist, int n, float viewing_cos_limit, float log_scale_factor,
                              int n_levels, uint8_t* in_view, float* uv, int32_t* level, float* view_cos, int invariance_bounds, float* dist) {
  ORBHIP_REQUIRE(n >= 0 && Rcw && tcw && K4 && bounds && n_levels > 0, ORBHIP_EINVAL, "NULL argument");
  if (n == 0) return 0;
  ORBHIP_REQUIRE(P && Pn && min_dist && max_dist && in_view && uv && level && view_cos, ORBHIP_EINVAL, "NULL argument");
  if (int rc = frame_need_device()) return rc;
  FrustumCam C;
  for (int k = 0; k < 9; k++) C.R[k] = Rcw[k];
  for (int k = 0; k < 3; k++) C.t[k] = tcw[k];
  for (int k = 0; k < 3; k++) C.Ow[k] = -(Rcw[k] * tcw[0] + Rcw[3 + k] * tcw[1] + Rcw[6 + k] * tcw[2]);   // Ow = -Rcw^T tcw (src/Frame.cc:188)
  C.fx = K4[0]; C.fy = K4[1]; C.cx = K4[2]; C.cy = K4[3];
  C.min_x = bounds[0]; C.max_x = bounds[1]; C.min_y = bounds[2]; C.max_y = bounds[3];
  C.cos_limit = viewing_cos_limit; C.log_scale = log_scale_factor; C.nlevels = n_levels;
  DevBuf dP, dN, dmin, dmax, dflag, duv, dlv, dvc, ddist;
  DevBuf* all[] = {&dP, &dN, &dmin, &dmax, &dflag, &duv, &dlv, &dvc, &ddist};
  auto cleanup = [&]() { for (DevBuf* b : all) b->release(); };
  int rc = 0;
  if ((rc = dP.ensure((size_t)n * 24)) || (rc = dN.ensure((size_t)n * 24)) || (rc = dmin.ensure((size_t)n * 4)) || (rc = dmax.ensure((size_t)n * 4)) ||
      (rc = dflag.ensure((size_t)n)) || (rc = duv.ensure((size_t)n * 8)) || (rc = dlv.ensure((size_t)n * 4)) || (rc = dvc.ensure((size_t)n * 4)) ||
      (dist && (rc = ddist.ensure((size_t)n * 4)))) { cleanup(); return rc; }
  FCHK(hipMemcpy(dP.p, P, (size_t)n * 24, hipMemcpyHostToDevice)); FCHK(hipMemcpy(dN.p, Pn, (size_t)n * 24, hipMemcpyHostToDevice));
  FCHK(hipMemcpy(dmin.p, min_dist, (size_t)n * 4, hipMemcpyHostToDevice)); FCHK(hipMemcpy(dmax.p, max_dist, (size_t)n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_frustum, dim3((n + 255) / 256), dim3(256), 0, 0, C, dP.as<double>(), dN.as<double>(), dmin.as<float>(), dmax.as<float>(), n,
                     dflag.as<uint8_t>(), duv.as<float>(), dlv.as<int>(), dvc.as<float>(), invariance_bounds, dist ? ddist.as<float>() : (float*)nullptr);
  FCHK(hipGetLastError());
  if (dist) FCHK(hipMemcpy(dist, ddist.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  FCHK(hipMemcpy(in_view, dflag.p, (size_t)n, hipMemcpyDeviceToHost)); FCHK(hipMemcpy(uv, duv.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  FCHK(hipMemcpy(level, dlv.p, (size_t)n * 4, hipMemcpyDeviceToHost)); FCHK(hipMemcpy(view_cos, dvc.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  cleanup();
  return 0;
}

int orbm_is_in_frustum(const double* Rcw, const double* tcw, const float* K4, const float* bounds, const double* P, const double* Pn,
                       const float* min_dist, const float* max_dist, int n, float viewing_cos_limit, float log_scale_factor,
                       int n_levels, uint8_t* in_view, float* uv, int32_t* level, float* view_cos) {
  return is_in_frustum_impl(Rcw, tcw, K4, bounds, P, Pn, min_dist, max_dist, n, viewing_cos_limit, log_scale_factor, n_levels, in_view, uv, level, view_cos, 0, nullptr);
}

// the form a caller outside MapPoint can feed: the PUBLIC accessors GetMinDistanceInvariance() / GetMaxDistanceInvariance()
// (min_distance_ / max_distance_ are protected, include/MapPoint.h:125-151) and the distance back, for MapPoint::PredictScale
int orbm_is_in_frustum_gates(const double* Rcw, const double* tcw, const float* K4, const float* bounds, const double* P, const double* Pn,
                             const float* min_dist_invariance, const float* max_dist_invariance, int n, float viewing_cos_limit,
                             uint8_t* in_view, float* uv, float* view_cos, float* dist) {
  ORBHIP_REQUIRE(n == 0 || dist, ORBHIP_EINVAL, "NULL argument");
  std::vector<int32_t> level((size_t)std::max(n, 1));
  return is_in_frustum_impl(Rcw, tcw, K4, bounds, P, Pn, min_dist_invariance, max_dist_invariance, n, viewing_cos_limit, 1.0f, 1, in_view, uv, level.data(), view_cos, 1, dist);
}

int orbm_assign_features_to_grid(const float* kps4, int n, const float* bounds, uint32_t* cell_offsets, uint32_t* cell_idx, int* n_assigned) {
  ORBHIP_REQUIRE(n >= 0 && bounds && cell_offsets && n_assigned && (n == 0 || (kps4 && cell_idx)), ORBHIP_EINVAL, "NULL argument");
  if (int rc = frame_need_device()) return rc;
  FrameGridDev g; DevBuf dk;
  auto cleanup = [&]() { g.release(); dk.release(); };
  int rc = 0;
  if ((rc = dk.ensure((size_t)std::max(n, 1) * 16))) { cleanup(); return rc; }
  if (n) FCHK(hipMemcpy(dk.p, kps4, (size_t)n * 16, hipMemcpyHostToDevice));
  if ((rc = frame_grid_build(g, dk.as<float>(), n, bounds, nullptr))) { cleanup(); return rc; }
  const int NC = FRAME_GRID_COLS * FRAME_GRID_ROWS;
  FCHK(hipMemcpy(cell_offsets, g.off.p, (size_t)(NC + 1) * 4, hipMemcpyDeviceToHost));
  *n_assigned = (int)cell_offsets[NC];
  if (*n_assigned) FCHK(hipMemcpy(cell_idx, g.idx.p, (size_t)(*n_assigned) * 4, hipMemcpyDeviceToHost));
  cleanup();
  return 0;
}

int orbm_features_in_area(const float* kps4, int n, const float* bounds, const float* q_xy, const float* q_radius,
                          const int32_t* q_min_level, const int32_t* q_max_level, int nq, uint32_t* cand_offsets, uint32_t* cand_idx,
                          int cap, int* total) {
  ORBHIP_REQUIRE(n >= 0 && nq >= 0 && bounds && cand_offsets && total && cap >= 0, ORBHIP_EINVAL, "NULL argument");
  *total = 0;
  for (int i = 0; i <= nq; i++) cand_offsets[i] = 0;
  if (n == 0 || nq == 0) return 0;
  ORBHIP_REQUIRE(kps4 && q_xy && q_radius, ORBHIP_EINVAL, "NULL argument");
  if (int rc = frame_need_device()) return rc;
  FrameGridDev g; DevBuf dk, dq, dr, dmn, dmx, dcnt, doff, didx;
  DevBuf* all[] = {&dk, &dq, &dr, &dmn, &dmx, &dcnt, &doff, &didx};
  auto cleanup = [&]() { g.release(); for (DevBuf* b : all) b->release(); };
  int rc = 0;
  if ((rc = dk.ensure((size_t)n * 16)) || (rc = dq.ensure((size_t)nq * 8)) || (rc = dr.ensure((size_t)nq * 4)) ||
      (q_min_level && (rc = dmn.ensure((size_t)nq * 4))) || (q_max_level && (rc = dmx.ensure((size_t)nq * 4)))) { cleanup(); return rc; }
  FCHK(hipMemcpy(dk.p, kps4, (size_t)n * 16, hipMemcpyHostToDevice)); FCHK(hipMemcpy(dq.p, q_xy, (size_t)nq * 8, hipMemcpyHostToDevice));
  FCHK(hipMemcpy(dr.p, q_radius, (size_t)nq * 4, hipMemcpyHostToDevice));
  if (q_min_level) FCHK(hipMemcpy(dmn.p, q_min_level, (size_t)nq * 4, hipMemcpyHostToDevice));
  if (q_max_level) FCHK(hipMemcpy(dmx.p, q_max_level, (size_t)nq * 4, hipMemcpyHostToDevice));
  if ((rc = frame_grid_build(g, dk.as<float>(), n, bounds, nullptr))) { cleanup(); return rc; }
  uint32_t tot = 0;
  if ((rc = frame_area_candidates(g, dk.as<float>(), dq.as<float>(), dr.as<float>(), q_min_level ? dmn.as<int>() : nullptr,
                                  q_max_level ? dmx.as<int>() : nullptr, nullptr, nq, dcnt, doff, didx, &tot, nullptr))) { cleanup(); return rc; }
  FCHK(hipMemcpy(cand_offsets, doff.p, (size_t)(nq + 1) * 4, hipMemcpyDeviceToHost));
  *total = (int)tot;
  if (tot > (uint32_t)cap) { cleanup(); if (cand_idx) { set_error("candidate capacity %d too small for %u candidates", cap, tot); return ORBHIP_ECAP; } return 0; }
  if (tot && cand_idx) FCHK(hipMemcpy(cand_idx, didx.p, (size_t)tot * 4, hipMemcpyDeviceToHost));
  cleanup();
  return 0;
}

}  // extern "C"

// ============================================================================ triangulation of matches
// LocalMapping::CreateNewMapPoints, per-match body (src/LocalMapping.cc:267-378, monocular): one thread per match.
namespace orbhip {

__global__ __launch_bounds__(128) void k_triangulate(TriCam C, const float* __restrict__ kp1, const float* __restrict__ kp2, int n,
                                                     const float* __restrict__ level_sigma2, const float* __restrict__ scale_factors,
                                                     double* __restrict__ x3D, uint8_t* __restrict__ ok) {
  const int m = blockIdx.x * 128 + threadIdx.x;
  if (m >= n) return;
  double X[3] = {0.0, 0.0, 0.0};
  const bool good = triangulate_one(C, kp1[3 * m], kp1[3 * m + 1], (int)kp1[3 * m + 2], kp2[3 * m], kp2[3 * m + 1], (int)kp2[3 * m + 2], level_sigma2, scale_factors, X);
  ok[m] = good ? 1 : 0;
  x3D[3 * m] = good ? X[0] : 0.0; x3D[3 * m + 1] = good ? X[1] : 0.0; x3D[3 * m + 2] = good ? X[2] : 0.0;
}

}  // namespace orbhip

extern "C" int orbm_triangulate_matches(const double* Tcw1, const double* Tcw2, const float* K1, const float* K2, const float* kp1,
                                        const float* kp2, int n, const float* level_sigma2, const float* scale_factors, int n_levels,
                                        float ratio_factor, double* x3D, uint8_t* ok) {
  ORBHIP_REQUIRE(n >= 0 && Tcw1 && Tcw2 && K1 && K2 && level_sigma2 && scale_factors && n_levels > 0, ORBHIP_EINVAL, "NULL argument");
  if (n == 0) return 0;
  ORBHIP_REQUIRE(kp1 && kp2 && x3D && ok, ORBHIP_EINVAL, "NULL argument");
  for (int i = 0; i < n; i++)
    ORBHIP_REQUIRE(kp1[3 * i + 2] >= 0 && kp1[3 * i + 2] < n_levels && kp2[3 * i + 2] >= 0 && kp2[3 * i + 2] < n_levels, ORBHIP_EINVAL, "octave out of range");
  if (int rc = frame_need_device()) return rc;
  TriCam C;
  for (int k = 0; k < 12; k++) { C.T1[k] = Tcw1[k]; C.T2[k] = Tcw2[k]; }
  for (int k = 0; k < 3; k++) {
    C.Ow1[k] = -(Tcw1[k] * Tcw1[3] + Tcw1[4 + k] * Tcw1[7] + Tcw1[8 + k] * Tcw1[11]);
    C.Ow2[k] = -(Tcw2[k] * Tcw2[3] + Tcw2[4 + k] * Tcw2[7] + Tcw2[8 + k] * Tcw2[11]);
  }
  for (int k = 0; k < 4; k++) { C.K1[k] = K1[k]; C.K2[k] = K2[k]; }
  C.ratio_factor = ratio_factor;
  DevBuf d1, d2, ds, dsf, dx, dok;
  DevBuf* all[] = {&d1, &d2, &ds, &dsf, &dx, &dok};
  auto cleanup = [&]() { for (DevBuf* b : all) b->release(); };
  int rc = 0;
  if ((rc = d1.ensure((size_t)n * 12)) || (rc = d2.ensure((size_t)n * 12)) || (rc = ds.ensure((size_t)n_levels * 4)) || (rc = dsf.ensure((size_t)n_levels * 4)) ||
      (rc = dx.ensure((size_t)n * 24)) || (rc = dok.ensure((size_t)n))) { cleanup(); return rc; }
  FCHK(hipMemcpy(d1.p, kp1, (size_t)n * 12, hipMemcpyHostToDevice)); FCHK(hipMemcpy(d2.p, kp2, (size_t)n * 12, hipMemcpyHostToDevice));
  FCHK(hipMemcpy(ds.p, level_sigma2, (size_t)n_levels * 4, hipMemcpyHostToDevice)); FCHK(hipMemcpy(dsf.p, scale_factors, (size_t)n_levels * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_triangulate, dim3((n + 127) / 128), dim3(128), 0, 0, C, d1.as<float>(), d2.as<float>(), n, ds.as<float>(), dsf.as<float>(),
                     dx.as<double>(), dok.as<uint8_t>());
  FCHK(hipGetLastError());
  FCHK(hipMemcpy(x3D, dx.p, (size_t)n * 24, hipMemcpyDeviceToHost)); FCHK(hipMemcpy(ok, dok.p, (size_t)n, hipMemcpyDeviceToHost));
  cleanup();
  return 0;
}
