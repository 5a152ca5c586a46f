// Device-side frame grid and window-candidate generation shared by orb_frame.hip and orb_matcher.hip
// (reference src/Frame.cc:158-173, 243-320).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "common.h"

namespace orbhip {

#define FRAME_GRID_COLS 64     // include/Frame.h:44-45
#define FRAME_GRID_ROWS 48

struct FrameGridDev {            // grid_[x][y] as CSR: cell c = x * ROWS + y -> idx[off[c] .. off[c+1])
  DevBuf cellid, cnt, off, idx;
  float min_x = 0, min_y = 0, winv = 0, hinv = 0;
  int n = 0;
  void release() { cellid.release(); cnt.release(); off.release(); idx.release(); }
};

// Frame::AssignFeaturesToGrid for n undistorted keypoints (device float4 records x, y, octave, angle); bounds = {min_x, max_x, min_y, max_y}
int frame_grid_build(FrameGridDev& g, const float* d_kps4, int n, const float* bounds, hipStream_t s);
// Frame::GetFeaturesInArea for nq queries -> CSR (cand_off[nq+1], cand_idx[total]) on the device, reference order
int frame_area_candidates(const FrameGridDev& g, const float* d_kps4, const float* d_q_xy, const float* d_q_r, const int* d_q_minl,
                          const int* d_q_maxl, const uint8_t* d_q_valid, int nq, DevBuf& cnt, DevBuf& cand_off, DevBuf& cand_idx,
                          uint32_t* total_out, hipStream_t s);

// the same without a host round trip: the lists are written into a buffer of `cap` entries the caller sized beforehand
// (entries beyond cap are dropped); the true total is cand_off[nq] on the device - the caller reads it with its own
// downloads, and re-runs with a larger buffer if it exceeds cap.  Entry k of the lists is written at d_cand_idx[k * idx_stride]
// (2 = interleaved with another per-candidate value).  Enqueue only.
int frame_area_candidates_enqueue(const FrameGridDev& g, const float* d_kps4, const float* d_q_xy, const float* d_q_r, const int* d_q_minl,
                                  const int* d_q_maxl, const uint8_t* d_q_valid, int nq, int* d_cnt, uint32_t* d_cand_off /*[nq+1]*/,
                                  uint32_t* d_cand_idx, uint32_t cap, int idx_stride, hipStream_t s);
// exclusive scan of d_cnt[0..n) into d_off[0..n] (one workgroup); enqueue only
int frame_scan_enqueue(const int* d_cnt, int n, uint32_t* d_off, hipStream_t s);
// Frame::isInFrustum (src/Frame.cc:191-241) + MapPoint::PredictScale (src/MapPoint.cc:406-420) for one map point, in the reference's
// float / double mix (k_frustum of orb_frame.hip and the TrackLocalMap step of orb_track.hip share it)
struct FrustumCam { double R[9], t[3], Ow[3]; float fx, fy, cx, cy, min_x, max_x, min_y, max_y, cos_limit, log_scale; int nlevels; };
__device__ __forceinline__ bool frustum_eval(const FrustumCam& C, const double* __restrict__ P, const double* __restrict__ Pn, const float min_dist, const float max_dist,
                                             const int invariance_bounds, float& u, float& v, int& nScale, float& vc, float& dist) {
  const double X = P[0], Y = P[1], Z = P[2];
  // Pc = Rcw * P + tcw in double, then narrowed to float (":199-202")
  const float PcX = (float)(C.R[0] * X + C.R[1] * Y + C.R[2] * Z + C.t[0]);
  const float PcY = (float)(C.R[3] * X + C.R[4] * Y + C.R[5] * Z + C.t[1]);
  const float PcZ = (float)(C.R[6] * X + C.R[7] * Y + C.R[8] * Z + C.t[2]);
  bool ok = !(PcZ < 0.0f);
  const float invz = 1.0f / PcZ;
  u = C.fx * PcX * invz + C.cx;
  v = C.fy * PcY * invz + C.cy;
  if (u < C.min_x || u > C.max_x) ok = false;
  if (v < C.min_y || v > C.max_y) ok = false;
  // GetMax/MinDistanceInvariance (src/MapPoint.cc:379-387): applied here, or by the caller (invariance_bounds)
  const float maxD = invariance_bounds ? max_dist : 1.2f * max_dist, minD = invariance_bounds ? min_dist : 0.8f * min_dist;
  const double POx = X - C.Ow[0], POy = Y - C.Ow[1], POz = Z - C.Ow[2];
  dist = (float)sqrt(POx * POx + POy * POy + POz * POz);
  if (dist < minD || dist > maxD) ok = false;
  vc = (float)((POx * Pn[0] + POy * Pn[1] + POz * Pn[2]) / (double)dist);
  if (vc < C.cos_limit) ok = false;
  // PredictScale (src/MapPoint.cc:406-420): float ratio, float log, ceil, clamp
  const float ratio = max_dist / dist;
  nScale = (int)ceilf(logf(ratio) / C.log_scale);
  if (nScale < 0) nScale = 0; else if (nScale >= C.nlevels) nScale = C.nlevels - 1;
  return ok;
}
}  // namespace orbhip
