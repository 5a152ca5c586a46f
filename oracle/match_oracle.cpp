// ============================================================================
// oracle/match_oracle.cpp -- CPU restatement of the reference ORB matcher core.
//
// TEST INFRASTRUCTURE ONLY (see oracle/orb_oracle.cpp header for the rule).
// PARITY STATUS: integer arithmetic restated directly from the reference source
// (src/ORBmatcher.cc, src/Frame.cc); no third-party arithmetic is involved, but
// the reference ships no tests, so parity is pinned only by the known-answer
// identities in tests/ (d(x,x)=0, d(0,~0)=256, popcount identity) -> "unpinned".
// ============================================================================
#include <cmath>
#include <climits>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

namespace {

const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;   // src/ORBmatcher.cc:35-37

// DescriptorDistance (src/ORBmatcher.cc:1422-1437): SWAR popcount over 8 x u32
int descriptor_distance(const uint8_t* a, const uint8_t* b) {
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t pa, pb;
    std::memcpy(&pa, a + 4 * i, 4); std::memcpy(&pb, b + 4 * i, 4);
    uint32_t v = pa ^ pb;
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

// ComputeThreeMaxima (src/ORBmatcher.cc:1386-1418) on bin counts
void three_maxima(const int* cnt, int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  ind1 = ind2 = ind3 = -1;
  for (int i = 0; i < L; i++) {
    const int s = cnt[i];
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

// rotation-histogram bin idiom (e.g. src/ORBmatcher.cc:431-437)
inline int rot_bin(float a1, float a2) {
  const float factor = 1.0f / HISTO_LENGTH;
  float rot = a1 - a2;
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)std::round(rot * factor);
  if (bin == HISTO_LENGTH) bin = 0;
  return bin;
}

// Frame grid (src/Frame.cc:158-173, 243-320), FRAME_GRID_COLS=64, FRAME_GRID_ROWS=48
struct Grid {
  static const int COLS = 64, ROWS = 48;
  float min_x, min_y, winv, hinv;
  std::vector<int> cell[COLS][ROWS];
  const float* kps;   // n x 4 floats: x, y, octave, angle
  int n;
  void build(const float* k, int n_, float minx, float maxx, float miny, float maxy) {
    kps = k; n = n_; min_x = minx; min_y = miny;
    winv = static_cast<float>(COLS) / (maxx - minx);     // src/Frame.cc:130-133 (grid_element_*_inv_)
    hinv = static_cast<float>(ROWS) / (maxy - miny);
    for (int i = 0; i < n; i++) {
      int px = (int)std::round((k[4 * i] - min_x) * winv);
      int py = (int)std::round((k[4 * i + 1] - min_y) * hinv);
      if (px < 0 || px >= COLS || py < 0 || py >= ROWS) continue;
      cell[px][py].push_back(i);
    }
  }
  void features_in_area(float x, float y, float r, int minLevel, int maxLevel, std::vector<int>& out) const {
    out.clear();
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
          const float dx = kps[4 * j] - x, dy = kps[4 * j + 1] - y;
          if (std::fabs(dx) < r && std::fabs(dy) < r) out.push_back(j);
        }
  }
};

}  // namespace

extern "C" {

int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) { return descriptor_distance(a, b); }

// best / second-best over candidates in candidate order, first minimum wins
// (the `dist < bestDist ... else if dist < bestDist2` idiom, e.g. src/ORBmatcher.cc:201-207).
// cand_offsets == NULL -> brute force over all nt targets in index order.
// Outputs: best_idx (-1 if no candidate), best_d, second_d (256 = none).
void orc_hamming_best2(const uint8_t* q, int nq, const uint8_t* t, int nt, const uint32_t* cand_offsets,
                       const uint32_t* cand_idx, int32_t* best_idx, int32_t* best_d, int32_t* second_d) {
  for (int i = 0; i < nq; i++) {
    int b1 = 256, b2 = 256, bi = -1;
    uint32_t lo = cand_offsets ? cand_offsets[i] : 0, hi = cand_offsets ? cand_offsets[i + 1] : (uint32_t)nt;
    for (uint32_t c = lo; c < hi; c++) {
      int j = cand_offsets ? (int)cand_idx[c] : (int)c;
      int d = descriptor_distance(q + 32 * (size_t)i, t + 32 * (size_t)j);
      if (d < b1) { b2 = b1; b1 = d; bi = j; }
      else if (d < b2) { b2 = d; }
    }
    best_idx[i] = bi; best_d[i] = b1; second_d[i] = b2;
  }
}

void orc_three_maxima(const int* cnt, int L, int* ind) { three_maxima(cnt, L, ind[0], ind[1], ind[2]); }
int orc_rot_bin(float a1, float a2) { return rot_bin(a1, a2); }

// The bench "match" workload: brute-force best/second-best of every query
// against every target, accept iff best <= th and best < ratio*second
// (SearchByBoW idiom, src/ORBmatcher.cc:210-212), then rotation-consistency filter
// (histogram of angle differences, keep the top-3 bins; :217-252).
// match12[i] = target index or -1.  Returns the number of surviving matches.
int orc_match_frames(const uint8_t* d1, const float* ang1, int n1, const uint8_t* d2, const float* ang2, int n2,
                     float ratio, int th, int check_ori, int32_t* match12) {
  std::vector<int32_t> bi(n1), bd(n1), sd(n1);
  orc_hamming_best2(d1, n1, d2, n2, nullptr, nullptr, bi.data(), bd.data(), sd.data());
  int cnt[HISTO_LENGTH] = {0};
  std::vector<int> bin(n1, -1);
  int nm = 0;
  for (int i = 0; i < n1; i++) {
    match12[i] = -1;
    if (bi[i] < 0) continue;
    if (bd[i] <= th && static_cast<float>(bd[i]) < ratio * static_cast<float>(sd[i])) {
      match12[i] = bi[i];
      nm++;
      if (check_ori) { bin[i] = rot_bin(ang1[i], ang2[bi[i]]); cnt[bin[i]]++; }
    }
  }
  if (check_ori) {
    int i1, i2, i3;
    three_maxima(cnt, HISTO_LENGTH, i1, i2, i3);
    for (int i = 0; i < n1; i++)
      if (match12[i] >= 0 && bin[i] != i1 && bin[i] != i2 && bin[i] != i3) { match12[i] = -1; nm--; }
  }
  return nm;
}

// candidate generation: Frame::GetFeaturesInArea for a list of queries -> CSR
// kps = n x 4 floats (x, y, octave, angle).  Returns total candidates (cand_idx may be NULL to size).
int orc_features_in_area(const float* kps, int n, const float* bounds /*minx,maxx,miny,maxy*/, const float* qxy,
                         const float* qr, const int* qminl, const int* qmaxl, int nq, uint32_t* offsets,
                         uint32_t* cand_idx, int cap) {
  static Grid* G = nullptr;
  delete G; G = new Grid();
  G->build(kps, n, bounds[0], bounds[1], bounds[2], bounds[3]);
  std::vector<int> v;
  int tot = 0;
  for (int i = 0; i < nq; i++) {
    offsets[i] = tot;
    G->features_in_area(qxy[2 * i], qxy[2 * i + 1], qr[i], qminl[i], qmaxl[i], v);
    for (int j : v) { if (cand_idx && tot < cap) cand_idx[tot] = j; tot++; }
  }
  offsets[nq] = tot;
  return tot;
}

// SearchForInitialization (src/ORBmatcher.cc:363-468) on flattened frames.
// kps1/kps2 = n x 4 floats (x,y,octave,angle) of the UNDISTORTED keypoints; prev_matched = n1 x 2 (in/out).
int orc_search_for_initialization(const float* kps1, const uint8_t* d1, int n1, const float* kps2, const uint8_t* d2,
                                  int n2, const float* bounds2, float* prev_matched, int window, float nnratio,
                                  int check_ori, int32_t* matches12) {
  Grid* G = new Grid();
  G->build(kps2, n2, bounds2[0], bounds2[1], bounds2[2], bounds2[3]);
  int nmatches = 0;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  std::vector<int> matchedDist(n2, INT_MAX), matches21(n2, -1);
  std::vector<int> cand;
  for (int i1 = 0; i1 < n1; i1++) {
    int level1 = (int)kps1[4 * i1 + 2];
    if (level1 > 0) continue;
    G->features_in_area(prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)window, level1, level1, cand);
    if (cand.empty()) continue;
    int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
    for (int i2 : cand) {
      int dist = descriptor_distance(d1 + 32 * (size_t)i1, d2 + 32 * (size_t)i2);
      if (matchedDist[i2] <= dist) continue;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
      else if (dist < bestDist2) { bestDist2 = dist; }
    }
    if (bestDist <= TH_LOW) {
      if (bestDist < (float)bestDist2 * nnratio) {
        if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; nmatches--; }
        matches12[i1] = bestIdx2; matches21[bestIdx2] = i1; matchedDist[bestIdx2] = bestDist;
        nmatches++;
        if (check_ori) rotHist[rot_bin(kps1[4 * i1 + 3], kps2[4 * bestIdx2 + 3])].push_back(i1);
      }
    }
  }
  if (check_ori) {
    int cnt[HISTO_LENGTH], i1, i2, i3;
    for (int i = 0; i < HISTO_LENGTH; i++) cnt[i] = (int)rotHist[i].size();
    three_maxima(cnt, HISTO_LENGTH, i1, i2, i3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == i1 || i == i2 || i == i3) continue;
      for (int idx1 : rotHist[i]) if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
    }
  }
  for (int i1 = 0; i1 < n1; i1++)
    if (matches12[i1] >= 0) { prev_matched[2 * i1] = kps2[4 * matches12[i1]]; prev_matched[2 * i1 + 1] = kps2[4 * matches12[i1] + 1]; }
  delete G;
  return nmatches;
}

// Projection-guided searches on flattened data (restates the common body of src/ORBmatcher.cc:42-119,
// :258-361, :1161-1271, :1273-1384 and the candidate selection of Fuse :724-954): per query the caller
// supplies the projected position, window radius, GetFeaturesInArea level range, optional predicted level
// ([pred-1, pred] post-filter) and descriptor; `taken` carries the "keypoint already holds a map point" state.
int orc_search_by_projection(const float* kps4, const uint8_t* desc, int n, const float* bounds, const float* q_uv,
                             const float* q_radius, const int32_t* q_min_level, const int32_t* q_max_level,
                             const int32_t* q_pred_level, const uint8_t* q_desc, const uint8_t* q_valid,
                             const float* q_angle, int nq, const float* inv_level_sigma2, float chi2_gate,
                             uint8_t* taken, int mode_best2, float ratio, int th, int check_ori, int32_t* q_match,
                             int32_t* q_best_dist) {
  Grid* G = new Grid();
  G->build(kps4, n, bounds[0], bounds[1], bounds[2], bounds[3]);
  std::vector<int> rotHist[HISTO_LENGTH];
  std::vector<int> cand;
  int nmatches = 0;
  for (int i = 0; i < nq; i++) {
    q_match[i] = -1;
    if (q_best_dist) q_best_dist[i] = 256;
    if (q_valid && !q_valid[i]) continue;
    G->features_in_area(q_uv[2 * i], q_uv[2 * i + 1], q_radius[i], q_min_level ? q_min_level[i] : -1,
                        q_max_level ? q_max_level[i] : -1, cand);
    if (cand.empty()) continue;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int idx : cand) {
      if (taken && taken[idx]) continue;
      const int kpLevel = (int)kps4[4 * idx + 2];
      if (q_pred_level && q_pred_level[i] >= 0)
        if (kpLevel < q_pred_level[i] - 1 || kpLevel > q_pred_level[i]) continue;
      if (chi2_gate > 0) {
        const float ex = q_uv[2 * i] - kps4[4 * idx], ey = q_uv[2 * i + 1] - kps4[4 * idx + 1];
        const float e2 = ex * ex + ey * ey;
        if (e2 * inv_level_sigma2[kpLevel] > chi2_gate) continue;
      }
      const int dist = descriptor_distance(q_desc + 32 * (size_t)i, desc + 32 * (size_t)idx);
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kpLevel; bestIdx = idx; }
      else if (mode_best2 && dist < bestDist2) { bestLevel2 = kpLevel; bestDist2 = dist; }
    }
    if (q_best_dist) q_best_dist[i] = bestDist;
    if (bestIdx >= 0 && bestDist <= th) {
      if (mode_best2 && bestLevel == bestLevel2 && bestDist > ratio * bestDist2) continue;
      q_match[i] = bestIdx;
      // (:83-84, :1220-1221: a feature whose map point has no observations stays open; q_valid bit 1 marks such a query point)
      if (taken && !(q_valid && (q_valid[i] & 2))) taken[bestIdx] = 1;
      nmatches++;
      if (check_ori) rotHist[rot_bin(q_angle[i], kps4[4 * bestIdx + 3])].push_back(i);
    }
  }
  if (check_ori) {
    int cnt[HISTO_LENGTH], i1, i2, i3;
    for (int b = 0; b < HISTO_LENGTH; b++) cnt[b] = (int)rotHist[b].size();
    three_maxima(cnt, HISTO_LENGTH, i1, i2, i3);
    for (int b = 0; b < HISTO_LENGTH; b++) {
      if (b == i1 || b == i2 || b == i3) continue;
      for (int qi : rotHist[b]) { if (taken) taken[q_match[qi]] = 0; q_match[qi] = -2 - q_match[qi]; nmatches--; }   // (:1260-1264: the slot is reset; the value keeps which slot)
    }
  }
  delete G;
  return nmatches;
}

// SearchBySim3 (src/ORBmatcher.cc:956-1159) on flattened data, from the two GetFeaturesInArea calls on.  The geometry in front
// of them (sR21 / sR12 transforms, depth, IsInImage, distance-invariance gates, PredictScale) stays with the caller, which
// passes per feature: q_valid (the feature holds a usable, not already matched map point and passed the gates), the projected
// position, radius = th * scale_factors_[nPredictedLevel] and nPredictedLevel; q12_desc / q21_desc rows = pMP->GetDescriptor().
int orc_search_by_sim3(const float* kps1, const uint8_t* desc1, int n1, const float* kps2, const uint8_t* desc2, int n2,
                       const float* bounds1, const float* bounds2, const float* q12_uv, const float* q12_radius, const int32_t* q12_pred,
                       const uint8_t* q12_valid, const uint8_t* q12_desc, const float* q21_uv, const float* q21_radius, const int32_t* q21_pred, const uint8_t* q21_valid,
                       const uint8_t* q21_desc, int32_t* match12) {
  if (!q12_desc) q12_desc = desc1;             // dMP = pMP->GetDescriptor() (:1036, :1112); default: the keyframe's own row
  if (!q21_desc) q21_desc = desc2;
  const int TH_HIGH = 100;
  Grid* G1 = new Grid(); Grid* G2 = new Grid();
  G1->build(kps1, n1, bounds1[0], bounds1[1], bounds1[2], bounds1[3]);     // every keyframe has its own grid bounds
  G2->build(kps2, n2, bounds2[0], bounds2[1], bounds2[2], bounds2[3]);
  std::vector<int> vnMatch1(n1, -1), vnMatch2(n2, -1), vIndices;
  // Transform from KF1 to KF2 and search (:995-1064)
  for (int i1 = 0; i1 < n1; i1++) {
    if (q12_valid && !q12_valid[i1]) continue;
    const int nPredictedLevel = q12_pred[i1];
    G2->features_in_area(q12_uv[2 * i1], q12_uv[2 * i1 + 1], q12_radius[i1], -1, -1, vIndices);       // KeyFrame::GetFeaturesInArea
    if (vIndices.empty()) continue;
    int bestDist = INT_MAX, bestIdx = -1;
    for (int idx : vIndices) {
      const int octave = (int)kps2[4 * idx + 2];
      if (octave < nPredictedLevel - 1 || octave > nPredictedLevel) continue;
      const int dist = descriptor_distance(q12_desc + 32 * (size_t)i1, desc2 + 32 * (size_t)idx);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    if (bestDist <= TH_HIGH) vnMatch1[i1] = bestIdx;
  }
  // Transform from KF2 to KF1 and search (:1066-1140)
  for (int i2 = 0; i2 < n2; i2++) {
    if (q21_valid && !q21_valid[i2]) continue;
    const int nPredictedLevel = q21_pred[i2];
    G1->features_in_area(q21_uv[2 * i2], q21_uv[2 * i2 + 1], q21_radius[i2], -1, -1, vIndices);
    if (vIndices.empty()) continue;
    int bestDist = INT_MAX, bestIdx = -1;
    for (int idx : vIndices) {
      const int octave = (int)kps1[4 * idx + 2];
      if (octave < nPredictedLevel - 1 || octave > nPredictedLevel) continue;
      const int dist = descriptor_distance(q21_desc + 32 * (size_t)i2, desc1 + 32 * (size_t)idx);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    if (bestDist <= TH_HIGH) vnMatch2[i2] = bestIdx;
  }
  // Check agreement (:1142-1157)
  int nFound = 0;
  for (int i1 = 0; i1 < n1; i1++) {
    match12[i1] = -1;
    const int idx2 = vnMatch1[i1];
    if (idx2 >= 0) {
      const int idx1 = vnMatch2[idx2];
      if (idx1 == i1) { match12[i1] = idx2; nFound++; }
    }
  }
  delete G1; delete G2;
  return nFound;
}

// SearchByBoW (src/ORBmatcher.cc:151-256 with strict = 0, :470-580 with strict = 1) on flattened data.
int orc_search_by_bow(const uint8_t* desc1, int n1, const uint8_t* valid1, const float* angle1, const uint8_t* desc2, int n2,
                      const uint8_t* valid2, const float* angle2, const uint32_t* fv1_node, const uint32_t* fv1_off,
                      const uint32_t* fv1_idx, int fv1_n, const uint32_t* fv2_node, const uint32_t* fv2_off,
                      const uint32_t* fv2_idx, int fv2_n, float ratio, int th, int strict, int check_ori, int32_t* match12) {
  for (int i = 0; i < n1; i++) match12[i] = -1;
  std::vector<bool> matched2(n2, false);
  std::vector<int> rotHist[HISTO_LENGTH];
  int nmatches = 0, a = 0, b = 0;
  while (a < fv1_n && b < fv2_n) {
    if (fv1_node[a] == fv2_node[b]) {
      for (uint32_t e1 = fv1_off[a]; e1 < fv1_off[a + 1]; e1++) {
        const int idx1 = (int)fv1_idx[e1];
        if (valid1 && !valid1[idx1]) continue;
        int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
        for (uint32_t e2 = fv2_off[b]; e2 < fv2_off[b + 1]; e2++) {
          const int idx2 = (int)fv2_idx[e2];
          if (matched2[idx2] || (valid2 && !valid2[idx2])) continue;
          const int dist = descriptor_distance(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
          if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
          else if (dist < bestDist2) { bestDist2 = dist; }
        }
        const bool under = strict ? (bestDist1 < th) : (bestDist1 <= th);
        if (under) {
          if (static_cast<float>(bestDist1) < ratio * static_cast<float>(bestDist2)) {
            match12[idx1] = bestIdx2; matched2[bestIdx2] = true;
            if (check_ori) rotHist[rot_bin(angle1[idx1], angle2[bestIdx2])].push_back(idx1);
            nmatches++;
          }
        }
      }
      a++; b++;
    } else if (fv1_node[a] < fv2_node[b]) {
      while (a < fv1_n && fv1_node[a] < fv2_node[b]) a++;          // lower_bound
    } else {
      while (b < fv2_n && fv2_node[b] < fv1_node[a]) b++;
    }
  }
  if (check_ori) {
    int cnt[HISTO_LENGTH], i1, i2, i3;
    for (int k = 0; k < HISTO_LENGTH; k++) cnt[k] = (int)rotHist[k].size();
    three_maxima(cnt, HISTO_LENGTH, i1, i2, i3);
    for (int k = 0; k < HISTO_LENGTH; k++) {
      if (k == i1 || k == i2 || k == i3) continue;
      for (int idx1 : rotHist[k]) { match12[idx1] = -1; nmatches--; }
    }
  }
  return nmatches;
}

// SearchForTriangulation (src/ORBmatcher.cc:582-722, mono: no stereo keypoints) + CheckDistEpipolarLine (:128-149)
int orc_search_for_triangulation(const float* kps1, const uint8_t* desc1, const uint8_t* unmapped1, int n1, const float* kps2,
                                 const uint8_t* desc2, const uint8_t* unmapped2, int n2, const uint32_t* fv1_node,
                                 const uint32_t* fv1_off, const uint32_t* fv1_idx, int fv1_n, const uint32_t* fv2_node,
                                 const uint32_t* fv2_off, const uint32_t* fv2_idx, int fv2_n, const double* F12, float ex, float ey,
                                 const float* scale_factors, const float* level_sigma2, int check_ori, int32_t* match12) {
  for (int i = 0; i < n1; i++) match12[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  int nmatches = 0, a = 0, b = 0;
  while (a < fv1_n && b < fv2_n) {
    if (fv1_node[a] == fv2_node[b]) {
      for (uint32_t e1 = fv1_off[a]; e1 < fv1_off[a + 1]; e1++) {
        const int idx1 = (int)fv1_idx[e1];
        if (unmapped1 && !unmapped1[idx1]) continue;
        const float k1x = kps1[4 * idx1], k1y = kps1[4 * idx1 + 1];
        int bestDist = TH_LOW, bestIdx2 = -1;
        for (uint32_t e2 = fv2_off[b]; e2 < fv2_off[b + 1]; e2++) {
          const int idx2 = (int)fv2_idx[e2];
          if (unmapped2 && !unmapped2[idx2]) continue;
          const int dist = descriptor_distance(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
          if (dist > TH_LOW || dist > bestDist) continue;
          const float k2x = kps2[4 * idx2], k2y = kps2[4 * idx2 + 1];
          const int oct2 = (int)kps2[4 * idx2 + 2];
          const float distex = ex - k2x, distey = ey - k2y;
          if (distex * distex + distey * distey < 100 * scale_factors[oct2]) continue;
          // CheckDistEpipolarLine
          const float la = k1x * F12[0] + k1y * F12[3] + F12[6];
          const float lb = k1x * F12[1] + k1y * F12[4] + F12[7];
          const float lc = k1x * F12[2] + k1y * F12[5] + F12[8];
          const float num = la * k2x + lb * k2y + lc;
          const float den = la * la + lb * lb;
          if (den == 0) continue;
          const float dsqr = num * num / den;
          if (dsqr < 3.84 * level_sigma2[oct2]) { bestIdx2 = idx2; bestDist = dist; }
        }
        if (bestIdx2 >= 0) {
          match12[idx1] = bestIdx2; nmatches++;
          if (check_ori) rotHist[rot_bin(kps1[4 * idx1 + 3], kps2[4 * bestIdx2 + 3])].push_back(idx1);
        }
      }
      a++; b++;
    } else if (fv1_node[a] < fv2_node[b]) a++;
    else b++;
  }
  if (check_ori) {
    int cnt[HISTO_LENGTH], i1, i2, i3;
    for (int k = 0; k < HISTO_LENGTH; k++) cnt[k] = (int)rotHist[k].size();
    three_maxima(cnt, HISTO_LENGTH, i1, i2, i3);
    for (int k = 0; k < HISTO_LENGTH; k++) {
      if (k == i1 || k == i2 || k == i3) continue;
      for (int idx1 : rotHist[k]) { match12[idx1] = -1; nmatches--; }
    }
  }
  return nmatches;
}

int orc_th_low() { return TH_LOW; }
int orc_th_high() { return TH_HIGH; }
int orc_histo_length() { return HISTO_LENGTH; }

}  // extern "C"

// ============================================================================ SURVEY N2: the steps either side of extract -> match
extern "C" {

// Frame::AssignFeaturesToGrid (src/Frame.cc:158-173): CSR form of grid_[x][y], cell = x * 48 + y
int orc_assign_features_to_grid(const float* kps, int n, const float* bounds /*minx,maxx,miny,maxy*/, uint32_t* cell_off, uint32_t* cell_idx) {
  Grid* G = new Grid();
  G->build(kps, n, bounds[0], bounds[1], bounds[2], bounds[3]);
  uint32_t tot = 0;
  for (int x = 0; x < Grid::COLS; x++)
    for (int y = 0; y < Grid::ROWS; y++) {
      cell_off[x * Grid::ROWS + y] = tot;
      for (int j : G->cell[x][y]) cell_idx[tot++] = (uint32_t)j;
    }
  cell_off[Grid::COLS * Grid::ROWS] = tot;
  delete G;
  return (int)tot;
}

// Frame::UndistortKeyPoints (src/Frame.cc:329-355) = cv::undistortPoints(src, dst, K, dist, noArray(), K): OpenCV 2.4 / 3.2,
// five fixed-point iterations in double, (k1, k2, p1, p2, k3), destination CV_32F.
void orc_undistort_keypoints(const float* xy, int n, const float* K4, const float* dist5, float* out) {
  if (dist5[0] == 0.0f) { for (int i = 0; i < 2 * n; i++) out[i] = xy[i]; return; }
  const double fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
  const double k1 = dist5[0], k2 = dist5[1], p1 = dist5[2], p2 = dist5[3], k3 = dist5[4];
  const double ifx = 1. / fx, ify = 1. / fy;
  for (int i = 0; i < n; i++) {
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
}

// Frame::isInFrustum (src/Frame.cc:191-241) + MapPoint::PredictScale (src/MapPoint.cc:406-420).  Canonical choice where the
// reference is ambiguous: log / ceil of the float ratio use the float overloads.
void orc_is_in_frustum(const double* R, const double* t, const float* K4, const float* bounds, const double* P, const double* Pn,
                       const float* min_dist, const float* max_dist, int n, float cos_limit, float log_scale, int nlevels,
                       uint8_t* in_view, float* uv, int* level, float* view_cos) {
  double Ow[3];
  for (int k = 0; k < 3; k++) Ow[k] = -(R[k] * t[0] + R[3 + k] * t[1] + R[6 + k] * t[2]);
  for (int i = 0; i < n; i++) {
    const double X = P[3 * i], Y = P[3 * i + 1], Z = P[3 * i + 2];
    const float PcX = (float)(R[0] * X + R[1] * Y + R[2] * Z + t[0]);
    const float PcY = (float)(R[3] * X + R[4] * Y + R[5] * Z + t[1]);
    const float PcZ = (float)(R[6] * X + R[7] * Y + R[8] * Z + t[2]);
    bool ok = !(PcZ < 0.0f);
    const float invz = 1.0f / PcZ;
    const float u = K4[0] * PcX * invz + K4[2];
    const float v = K4[1] * PcY * invz + K4[3];
    if (u < bounds[0] || u > bounds[1]) ok = false;
    if (v < bounds[2] || v > bounds[3]) ok = false;
    const float maxD = 1.2f * max_dist[i], minD = 0.8f * min_dist[i];
    const double POx = X - Ow[0], POy = Y - Ow[1], POz = Z - Ow[2];
    const float dist = (float)std::sqrt(POx * POx + POy * POy + POz * POz);
    if (dist < minD || dist > maxD) ok = false;
    const float vc = (float)((POx * Pn[3 * i] + POy * Pn[3 * i + 1] + POz * Pn[3 * i + 2]) / (double)dist);
    if (vc < cos_limit) ok = false;
    const float ratio = max_dist[i] / dist;
    int nScale = (int)std::ceil(std::log(ratio) / log_scale);       // float overloads
    if (nScale < 0) nScale = 0; else if (nScale >= nlevels) nScale = nlevels - 1;
    in_view[i] = ok ? 1 : 0; uv[2 * i] = u; uv[2 * i + 1] = v; level[i] = nScale; view_cos[i] = vc;
  }
}

}  // extern "C"
