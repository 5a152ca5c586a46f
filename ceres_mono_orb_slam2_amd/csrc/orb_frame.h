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
}  // namespace orbhip
