// Literal CPU restatements of the reference's WHOLE matcher / optimizer entry points on the mock data model
// (tests/cpp/mock_orbslam.h): the loops of src/ORBmatcher.cc and src/CeresOptimizer.cc as they are written there - per-query
// GetFeaturesInArea on the host grid, DescriptorDistance by popcount, the pointer-graph walks and mutations - with the Ceres
// solve replaced by the CPU oracle's (oracle/_build/liborb_oracle.so).  TEST INFRASTRUCTURE: the checker the drop-in shims
// (csrc/compat/orbslam_dropin.h, HIP underneath) are compared with in tests/cpp/test_dropin.cpp.
#pragma once
#include <climits>
#include <unordered_map>

#include "mock_orbslam.h"

extern "C" {
struct orc_ba_opts { int max_iters; double huber_delta; int fix_points; const volatile uint8_t* stop; };
struct orc_ba_summary { double initial_cost, final_cost; int iterations, successful_steps, termination; double final_radius; };
int orc_ba_solve(const double* K4, double* poses7, const uint8_t* cam_fixed, int ncam, double* pts3, int npts, const int32_t* obs_cam, const int32_t* obs_pt,
                 const double* obs_uv, const double* obs_w, const uint8_t* obs_robust, int nobs, const orc_ba_opts* opts, orc_ba_summary* sum);
int orc_pose_optimization(const double* K4, double* pose7, const double* Xw, const double* uv, const float* inv_sigma2, int n, uint8_t* outlier, orc_ba_summary* sum);
int orc_local_ba(const double* K4, double* poses7, const uint8_t* cam_fixed, const uint8_t* cam_local, int ncam, double* pts3, int npts, const int32_t* obs_cam,
                 const int32_t* obs_pt, const double* obs_uv, const float* obs_inv_sigma2, int nobs, const volatile uint8_t* stop, int duplicate_blocks,
                 uint8_t* obs_erase, orc_ba_summary* s1, orc_ba_summary* s2);
void orc_matrix4d_to_pose7(const double* T, double* out);
void orc_pose7_to_matrix4d(const double* p, double* T);
}

namespace literal {
using namespace mock;
const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;

inline int DescriptorDistance(const Mat& a, const Mat& b) {          // src/ORBmatcher.cc:1422-1437
  int dist = 0;
  for (int i = 0; i < 8; i++) { uint32_t x, y; std::memcpy(&x, a.ptr(0) + 4 * i, 4); std::memcpy(&y, b.ptr(0) + 4 * i, 4); dist += __builtin_popcount(x ^ y); }
  return dist;
}
inline void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {     // :1386-1418
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = histo[i].size();
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; } else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}
inline int RotBin(float a1, float a2) { const float factor = 1.0f / HISTO_LENGTH; float rot = a1 - a2; if (rot < 0.0) rot += 360.0f; int bin = std::round(rot * factor); if (bin == HISTO_LENGTH) bin = 0; return bin; }

struct ORBmatcher {
  float mfNNratio; bool mbCheckOrientation;
  ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
  float RadiusByViewingCos(const float& viewCos) { return viewCos > 0.998 ? 2.5 : 4.0; }

  int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3) {            // :42-119
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++) {
      MapPoint* pMP = vpMapPoints[iMP];
      if (!pMP->is_track_in_view_) continue;
      if (pMP->isBad()) continue;
      const int& nPredictedLevel = pMP->track_scale_level_;
      float r = RadiusByViewingCos(pMP->track_view_cos_);
      if (bFactor) r *= th;
      const std::vector<size_t> vIndices = F.GetFeaturesInArea(pMP->track_proj_x_, pMP->track_proj_y_, r * F.scale_factors_[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel);
      if (vIndices.empty()) continue;
      const Mat MPdescriptor = pMP->GetDescriptor();
      int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
      for (size_t idx : vIndices) {
        if (F.map_points_[idx]) if (F.map_points_[idx]->Observations() > 0) continue;
        const int dist = DescriptorDistance(MPdescriptor, F.descriptors_.row(idx));
        if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = F.undistort_keypoints_[idx].octave; bestIdx = idx; }
        else if (dist < bestDist2) { bestLevel2 = F.undistort_keypoints_[idx].octave; bestDist2 = dist; }
      }
      if (bestDist <= TH_HIGH) {
        if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;
        F.map_points_[bestIdx] = pMP;
        nmatches++;
      }
    }
    return nmatches;
  }

  int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th) {                       // :1161-1271
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    Matrix3d Rcw; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rcw(r, c) = CurrentFrame.Tcw_(r, c);
    const Vector3d tcw(CurrentFrame.Tcw_(0, 3), CurrentFrame.Tcw_(1, 3), CurrentFrame.Tcw_(2, 3));
    for (int i = 0; i < LastFrame.N_; i++) {
      MapPoint* map_point = LastFrame.map_points_[i];
      if (map_point) {
        if (!LastFrame.is_outliers_[i]) {
          Vector3d x3Dw = map_point->GetWorldPos();
          Vector3d x3Dc = Rcw * x3Dw + tcw;
          const float xc = x3Dc[0], yc = x3Dc[1];
          const float invzc = 1.0 / x3Dc[2];
          if (invzc < 0) continue;
          float u = CurrentFrame.fx_ * xc * invzc + CurrentFrame.cx_;
          float v = CurrentFrame.fy_ * yc * invzc + CurrentFrame.cy_;
          if (u < CurrentFrame.min_x_ || u > CurrentFrame.max_x_) continue;
          if (v < CurrentFrame.min_y_ || v > CurrentFrame.max_y_) continue;
          int nLastOctave = LastFrame.keypoints_[i].octave;
          float radius = th * CurrentFrame.scale_factors_[nLastOctave];
          std::vector<size_t> vIndices2 = CurrentFrame.GetFeaturesInArea(u, v, radius, nLastOctave - 1, nLastOctave + 1);
          if (vIndices2.empty()) continue;
          const Mat dMP = map_point->GetDescriptor();
          int bestDist = 256, bestIdx2 = -1;
          for (size_t i2 : vIndices2) {
            if (CurrentFrame.map_points_[i2]) if (CurrentFrame.map_points_[i2]->Observations() > 0) continue;
            const int dist = DescriptorDistance(dMP, CurrentFrame.descriptors_.row(i2));
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
          }
          if (bestDist <= TH_HIGH) {
            CurrentFrame.map_points_[bestIdx2] = map_point;
            nmatches++;
            if (mbCheckOrientation) rotHist[RotBin(LastFrame.undistort_keypoints_[i].angle, CurrentFrame.undistort_keypoints_[bestIdx2].angle)].push_back(bestIdx2);
          }
        }
      }
    }
    if (mbCheckOrientation) {
      int ind1 = -1, ind2 = -1, ind3 = -1;
      ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
      for (int i = 0; i < HISTO_LENGTH; i++)
        if (i != ind1 && i != ind2 && i != ind3)
          for (size_t j = 0; j < rotHist[i].size(); j++) { CurrentFrame.map_points_[rotHist[i][j]] = nullptr; nmatches--; }
    }
    return nmatches;
  }

  int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist) {   // :1273-1384
    int nmatches = 0;
    Matrix3d Rcw; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rcw(r, c) = CurrentFrame.Tcw_(r, c);
    const Vector3d tcw(CurrentFrame.Tcw_(0, 3), CurrentFrame.Tcw_(1, 3), CurrentFrame.Tcw_(2, 3));
    const Vector3d Rt = Rcw.transpose() * tcw; const Vector3d Ow(-Rt[0], -Rt[1], -Rt[2]);
    std::vector<int> rotHist[HISTO_LENGTH];
    const std::vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
    for (size_t i = 0; i < vpMPs.size(); i++) {
      MapPoint* pMP = vpMPs[i];
      if (pMP) {
        if (!pMP->isBad() && !sAlreadyFound.count(pMP)) {
          Vector3d x3Dw = pMP->GetWorldPos();
          Vector3d x3Dc = Rcw * x3Dw + tcw;
          const float xc = x3Dc[0], yc = x3Dc[1];
          const float invzc = 1.0 / x3Dc[2];
          const float u = CurrentFrame.fx_ * xc * invzc + CurrentFrame.cx_;
          const float v = CurrentFrame.fy_ * yc * invzc + CurrentFrame.cy_;
          if (u < CurrentFrame.min_x_ || u > CurrentFrame.max_x_) continue;
          if (v < CurrentFrame.min_y_ || v > CurrentFrame.max_y_) continue;
          Vector3d PO = x3Dw - Ow;
          float dist3D = PO.norm();
          const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
          if (dist3D < minDistance || dist3D > maxDistance) continue;
          int nPredictedLevel = pMP->PredictScale(dist3D, &CurrentFrame);
          const float radius = th * CurrentFrame.scale_factors_[nPredictedLevel];
          const std::vector<size_t> vIndices2 = CurrentFrame.GetFeaturesInArea(u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1);
          if (vIndices2.empty()) continue;
          const Mat dMP = pMP->GetDescriptor();
          int bestDist = 256, bestIdx2 = -1;
          for (size_t i2 : vIndices2) {
            if (CurrentFrame.map_points_[i2]) continue;
            const int dist = DescriptorDistance(dMP, CurrentFrame.descriptors_.row(i2));
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
          }
          if (bestDist <= ORBdist) {
            CurrentFrame.map_points_[bestIdx2] = pMP;
            nmatches++;
            if (mbCheckOrientation) rotHist[RotBin(pKF->undistort_keypoints_[i].angle, CurrentFrame.undistort_keypoints_[bestIdx2].angle)].push_back(bestIdx2);
          }
        }
      }
    }
    if (mbCheckOrientation) {
      int ind1 = -1, ind2 = -1, ind3 = -1;
      ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
      for (int i = 0; i < HISTO_LENGTH; i++)
        if (i != ind1 && i != ind2 && i != ind3)
          for (size_t j = 0; j < rotHist[i].size(); j++) { CurrentFrame.map_points_[rotHist[i][j]] = nullptr; nmatches--; }
    }
    return nmatches;
  }

  struct Dec { Matrix3d Rcw; Vector3d tcw, Ow; };
  static Dec Decompose(const Matrix4d& Scw) {                                                                  // :269-274
    Dec D; Matrix3d sRcw; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) sRcw(r, c) = Scw(r, c);
    const float scw = std::sqrt(sRcw(0, 0) * sRcw(0, 0) + sRcw(0, 1) * sRcw(0, 1) + sRcw(0, 2) * sRcw(0, 2));
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) D.Rcw(r, c) = sRcw(r, c) / scw;
    D.tcw = Vector3d(Scw(0, 3) / scw, Scw(1, 3) / scw, Scw(2, 3) / scw);
    const Vector3d o = D.Rcw.transpose() * D.tcw; D.Ow = Vector3d(-o[0], -o[1], -o[2]);
    return D;
  }

  int SearchByProjection(KeyFrame* pKF, const Matrix4d& Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th) {   // :258-361
    const float &fx = pKF->fx_, &fy = pKF->fy_, &cx = pKF->cx_, &cy = pKF->cy_;
    const Dec D = Decompose(Scw);
    std::set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MapPoint*>(nullptr));
    int nmatches = 0;
    for (int iMP = 0, iendMP = vpPoints.size(); iMP < iendMP; iMP++) {
      MapPoint* pMP = vpPoints[iMP];
      if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
      Vector3d p3Dw = pMP->GetWorldPos();
      Vector3d p3Dc = D.Rcw * p3Dw + D.tcw;
      if (p3Dc[2] < 0.0) continue;
      const float invz = 1 / p3Dc[2];
      const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
      const float u = fx * x + cx, v = fy * y + cy;
      if (!pKF->IsInImage(u, v)) continue;
      const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
      Vector3d PO = p3Dw - D.Ow;
      const float dist = PO.norm();
      if (dist < minDistance || dist > maxDistance) continue;
      Vector3d Pn = pMP->GetNormal();
      if (PO.dot(Pn) < 0.5 * dist) continue;
      int nPredictedLevel = pMP->PredictScale(dist, pKF);
      const float radius = th * pKF->scale_factors_[nPredictedLevel];
      const std::vector<size_t> vIndices = pKF->GetFeaturesInArea(u, v, radius);
      if (vIndices.empty()) continue;
      const Mat dMP = pMP->GetDescriptor();
      int bestDist = 256, bestIdx = -1;
      for (size_t idx : vIndices) {
        if (vpMatched[idx]) continue;
        const int& kpLevel = pKF->undistort_keypoints_[idx].octave;
        if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
        const int dist2 = DescriptorDistance(dMP, pKF->descriptors_.row(idx));
        if (dist2 < bestDist) { bestDist = dist2; bestIdx = idx; }
      }
      if (bestDist <= TH_LOW) { vpMatched[bestIdx] = pMP; nmatches++; }
    }
    return nmatches;
  }

  // the merge-walk of the two feature vectors shared by :151-256, :470-580, :582-722
  template <class Body> static void WalkNodes(const FeatureVector& a, const FeatureVector& b, Body body) {
    auto ait = a.begin(), bit = b.begin();
    while (ait != a.end() && bit != b.end()) {
      if (ait->first == bit->first) { body(ait->second, bit->second); ait++; bit++; }
      else if (ait->first < bit->first) ait = a.lower_bound(bit->first);
      else bit = b.lower_bound(ait->first);
    }
  }

  int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) {                       // :151-256
    const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPoint*>(F.N_, static_cast<MapPoint*>(nullptr));
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    WalkNodes(pKF->feature_vector_, F.feature_vector_, [&](const std::vector<unsigned>& vIndicesKF, const std::vector<unsigned>& vIndicesF) {
      for (size_t iKF = 0; iKF < vIndicesKF.size(); iKF++) {
        const unsigned realIdxKF = vIndicesKF[iKF];
        MapPoint* pMP = vpMapPointsKF[realIdxKF];
        if (!pMP) continue;
        if (pMP->isBad()) continue;
        const Mat dKF = pKF->descriptors_.row(realIdxKF);
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        for (size_t iF = 0; iF < vIndicesF.size(); iF++) {
          const unsigned realIdxF = vIndicesF[iF];
          if (vpMapPointMatches[realIdxF]) continue;
          const int dist = DescriptorDistance(dKF, F.descriptors_.row(realIdxF));
          if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
          else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist1 <= TH_LOW) {
          if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
            vpMapPointMatches[bestIdxF] = pMP;
            if (mbCheckOrientation) rotHist[RotBin(pKF->undistort_keypoints_[realIdxKF].angle, F.keypoints_[bestIdxF].angle)].push_back(bestIdxF);
            nmatches++;
          }
        }
      }
    });
    if (mbCheckOrientation) {
      int ind1 = -1, ind2 = -1, ind3 = -1;
      ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
      for (int i = 0; i < HISTO_LENGTH; i++) {
        if (i == ind1 || i == ind2 || i == ind3) continue;
        for (size_t j = 0; j < rotHist[i].size(); j++) { vpMapPointMatches[rotHist[i][j]] = nullptr; nmatches--; }
      }
    }
    return nmatches;
  }

  int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) {                      // :470-580
    const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = std::vector<MapPoint*>(vpMapPoints1.size(), static_cast<MapPoint*>(nullptr));
    std::vector<bool> vbMatched2(vpMapPoints2.size(), false);
    std::vector<int> rotHist[HISTO_LENGTH];
    int nmatches = 0;
    WalkNodes(pKF1->feature_vector_, pKF2->feature_vector_, [&](const std::vector<unsigned>& l1, const std::vector<unsigned>& l2) {
      for (size_t i1 = 0; i1 < l1.size(); i1++) {
        const size_t idx1 = l1[i1];
        MapPoint* pMP1 = vpMapPoints1[idx1];
        if (!pMP1) continue;
        if (pMP1->isBad()) continue;
        const Mat d1 = pKF1->descriptors_.row(idx1);
        int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
        for (size_t i2 = 0; i2 < l2.size(); i2++) {
          const size_t idx2 = l2[i2];
          MapPoint* pMP2 = vpMapPoints2[idx2];
          if (vbMatched2[idx2] || !pMP2) continue;
          if (pMP2->isBad()) continue;
          int dist = DescriptorDistance(d1, pKF2->descriptors_.row(idx2));
          if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
          else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist1 < TH_LOW) {
          if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
            vpMatches12[idx1] = vpMapPoints2[bestIdx2];
            vbMatched2[bestIdx2] = true;
            if (mbCheckOrientation) rotHist[RotBin(pKF1->undistort_keypoints_[idx1].angle, pKF2->undistort_keypoints_[bestIdx2].angle)].push_back(idx1);
            nmatches++;
          }
        }
      }
    });
    if (mbCheckOrientation) {
      int ind1 = -1, ind2 = -1, ind3 = -1;
      ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
      for (int i = 0; i < HISTO_LENGTH; i++) {
        if (i == ind1 || i == ind2 || i == ind3) continue;
        for (size_t j = 0; j < rotHist[i].size(); j++) { vpMatches12[rotHist[i][j]] = nullptr; nmatches--; }
      }
    }
    return nmatches;
  }

  int SearchForInitialization(Frame& F1, Frame& F2, std::vector<Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10) {   // :363-468
    int nmatches = 0;
    vnMatches12 = std::vector<int>(F1.undistort_keypoints_.size(), -1);
    std::vector<int> rotHist[HISTO_LENGTH];
    std::vector<int> vMatchedDistance(F2.undistort_keypoints_.size(), INT_MAX), vnMatches21(F2.undistort_keypoints_.size(), -1);
    for (size_t i1 = 0; i1 < F1.undistort_keypoints_.size(); i1++) {
      KeyPoint kp1 = F1.undistort_keypoints_[i1];
      int level1 = kp1.octave;
      if (level1 > 0) continue;
      std::vector<size_t> vIndices2 = F2.GetFeaturesInArea(vbPrevMatched[i1].x, vbPrevMatched[i1].y, windowSize, level1, level1);
      if (vIndices2.empty()) continue;
      Mat d1 = F1.descriptors_.row(i1);
      int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
      for (size_t i2 : vIndices2) {
        int dist = DescriptorDistance(d1, F2.descriptors_.row(i2));
        if (vMatchedDistance[i2] <= dist) continue;
        if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
        else if (dist < bestDist2) bestDist2 = dist;
      }
      if (bestDist <= TH_LOW) {
        if (bestDist < (float)bestDist2 * mfNNratio) {
          if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
          vnMatches12[i1] = bestIdx2; vnMatches21[bestIdx2] = i1; vMatchedDistance[bestIdx2] = bestDist; nmatches++;
          if (mbCheckOrientation) rotHist[RotBin(F1.undistort_keypoints_[i1].angle, F2.undistort_keypoints_[bestIdx2].angle)].push_back(i1);
        }
      }
    }
    if (mbCheckOrientation) {
      int ind1 = -1, ind2 = -1, ind3 = -1;
      ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
      for (int i = 0; i < HISTO_LENGTH; i++) {
        if (i == ind1 || i == ind2 || i == ind3) continue;
        for (size_t j = 0; j < rotHist[i].size(); j++) { int idx1 = rotHist[i][j]; if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; } }
      }
    }
    for (size_t i1 = 0; i1 < vnMatches12.size(); i1++) if (vnMatches12[i1] >= 0) vbPrevMatched[i1] = F2.undistort_keypoints_[vnMatches12[i1]].pt;
    return nmatches;
  }

  bool CheckDistEpipolarLine(const KeyPoint& kp1, const KeyPoint& kp2, const Matrix3d& F12, const KeyFrame* pKF2) {   // :128-149
    const float a = kp1.pt.x * F12(0, 0) + kp1.pt.y * F12(1, 0) + F12(2, 0);
    const float b = kp1.pt.x * F12(0, 1) + kp1.pt.y * F12(1, 1) + F12(2, 1);
    const float c = kp1.pt.x * F12(0, 2) + kp1.pt.y * F12(1, 2) + F12(2, 2);
    const float num = a * kp2.pt.x + b * kp2.pt.y + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * pKF2->level_sigma2s_[kp2.octave];
  }

  int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, const Matrix3d& F12, std::vector<std::pair<size_t, size_t> >& vMatchedPairs, const bool bOnlyStereo) {   // :582-722
    Vector3d Cw = pKF1->GetCameraCenter();
    Matrix3d R2w = pKF2->GetRotation(); Vector3d t2w = pKF2->GetTranslation();
    Vector3d C2 = R2w * Cw + t2w;
    const float invz = 1.0f / C2[2];
    const float ex = pKF2->fx_ * C2[0] * invz + pKF2->cx_, ey = pKF2->fy_ * C2[1] * invz + pKF2->cy_;
    int nmatches = 0;
    std::vector<bool> vbMatched2(pKF2->N_, false);
    std::vector<int> vMatches12(pKF1->N_, -1);
    std::vector<int> rotHist[HISTO_LENGTH];
    WalkNodes(pKF1->feature_vector_, pKF2->feature_vector_, [&](const std::vector<unsigned>& l1, const std::vector<unsigned>& l2) {
      for (size_t i1 = 0; i1 < l1.size(); i1++) {
        const size_t idx1 = l1[i1];
        MapPoint* pMP1 = pKF1->GetMapPoint(idx1);
        if (pMP1) continue;
        const bool bStereo1 = false;
        if (bOnlyStereo) if (!bStereo1) continue;
        const KeyPoint& kp1 = pKF1->undistort_keypoints_[idx1];
        const Mat d1 = pKF1->descriptors_.row(idx1);
        int bestDist = TH_LOW, bestIdx2 = -1;
        for (size_t i2 = 0; i2 < l2.size(); i2++) {
          size_t idx2 = l2[i2];
          MapPoint* pMP2 = pKF2->GetMapPoint(idx2);
          if (vbMatched2[idx2] || pMP2) continue;
          const int dist = DescriptorDistance(d1, pKF2->descriptors_.row(idx2));
          if (dist > TH_LOW || dist > bestDist) continue;
          const KeyPoint& kp2 = pKF2->undistort_keypoints_[idx2];
          const float distex = ex - kp2.pt.x, distey = ey - kp2.pt.y;
          if (distex * distex + distey * distey < 100 * pKF2->scale_factors_[kp2.octave]) continue;
          if (CheckDistEpipolarLine(kp1, kp2, F12, pKF2)) { bestIdx2 = idx2; bestDist = dist; }
        }
        if (bestIdx2 >= 0) {
          const KeyPoint& kp2 = pKF2->undistort_keypoints_[bestIdx2];
          vMatches12[idx1] = bestIdx2; nmatches++;
          if (mbCheckOrientation) rotHist[RotBin(kp1.angle, kp2.angle)].push_back(idx1);
        }
      }
    });
    if (mbCheckOrientation) {
      int ind1 = -1, ind2 = -1, ind3 = -1;
      ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
      for (int i = 0; i < HISTO_LENGTH; i++) {
        if (i == ind1 || i == ind2 || i == ind3) continue;
        for (size_t j = 0; j < rotHist[i].size(); j++) { vMatches12[rotHist[i][j]] = -1; nmatches--; }
      }
    }
    vMatchedPairs.clear();
    for (size_t i = 0; i < vMatches12.size(); i++) { if (vMatches12[i] < 0) continue; vMatchedPairs.push_back(std::make_pair(i, (size_t)vMatches12[i])); }
    return nmatches;
  }

  int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const Matrix3d& R12, const Vector3d& t12, const float th) {   // :956-1159
    const float &fx = pKF1->fx_, &fy = pKF1->fy_, &cx = pKF1->cx_, &cy = pKF1->cy_;
    Matrix3d R1w = pKF1->GetRotation(); Vector3d t1w = pKF1->GetTranslation();
    Matrix3d R2w = pKF2->GetRotation(); Vector3d t2w = pKF2->GetTranslation();
    Matrix3d sR12, sR21;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { sR12(r, c) = s12 * R12(r, c); sR21(r, c) = (1.0 / s12) * R12(c, r); }
    const Vector3d t21r = sR21 * t12; const Vector3d t21(-t21r[0], -t21r[1], -t21r[2]);
    const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches(); const int N1 = vpMapPoints1.size();
    const std::vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches(); const int N2 = vpMapPoints2.size();
    std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
    for (int i = 0; i < N1; i++) {
      MapPoint* pMP = vpMatches12[i];
      if (pMP) { vbAlreadyMatched1[i] = true; int idx2 = pMP->GetIndexInKeyFrame(pKF2); if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true; }
    }
    std::vector<int> vnMatch1(N1, -1), vnMatch2(N2, -1);
    for (int i1 = 0; i1 < N1; i1++) {
      MapPoint* pMP = vpMapPoints1[i1];
      if (!pMP || vbAlreadyMatched1[i1]) continue;
      if (pMP->isBad()) continue;
      Vector3d p3Dw = pMP->GetWorldPos();
      Vector3d p3Dc1 = R1w * p3Dw + t1w;
      Vector3d p3Dc2 = sR21 * p3Dc1 + t21;
      if (p3Dc2[2] < 0.0) continue;
      const float invz = 1.0 / p3Dc2[2];
      const float x = p3Dc2[0] * invz, y = p3Dc2[1] * invz;
      const float u = fx * x + cx, v = fy * y + cy;
      if (!pKF2->IsInImage(u, v)) continue;
      const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
      const float dist3D = p3Dc2.norm();
      if (dist3D < minDistance || dist3D > maxDistance) continue;
      const int nPredictedLevel = pMP->PredictScale(dist3D, pKF2);
      const float radius = th * pKF2->scale_factors_[nPredictedLevel];
      const std::vector<size_t> vIndices = pKF2->GetFeaturesInArea(u, v, radius);
      if (vIndices.empty()) continue;
      const Mat dMP = pMP->GetDescriptor();
      int bestDist = INT_MAX, bestIdx = -1;
      for (size_t idx : vIndices) {
        const KeyPoint& kp = pKF2->undistort_keypoints_[idx];
        if (kp.octave < nPredictedLevel - 1 || kp.octave > nPredictedLevel) continue;
        const int dist = DescriptorDistance(dMP, pKF2->descriptors_.row(idx));
        if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
      }
      if (bestDist <= TH_HIGH) vnMatch1[i1] = bestIdx;
    }
    for (int i2 = 0; i2 < N2; i2++) {
      MapPoint* pMP = vpMapPoints2[i2];
      if (!pMP || vbAlreadyMatched2[i2]) continue;
      if (pMP->isBad()) continue;
      Vector3d p3Dw = pMP->GetWorldPos();
      Vector3d p3Dc2 = R2w * p3Dw + t2w;
      Vector3d p3Dc1 = sR12 * p3Dc2 + t12;
      if (p3Dc1[2] < 0.0) continue;
      const float invz = 1.0 / p3Dc1[2];
      const float x = p3Dc1[0] * invz, y = p3Dc1[1] * invz;
      const float u = fx * x + cx, v = fy * y + cy;
      if (!pKF1->IsInImage(u, v)) continue;
      const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
      const float dist3D = p3Dc1.norm();
      if (dist3D < minDistance || dist3D > maxDistance) continue;
      const int nPredictedLevel = pMP->PredictScale(dist3D, pKF1);
      const float radius = th * pKF1->scale_factors_[nPredictedLevel];
      const std::vector<size_t> vIndices = pKF1->GetFeaturesInArea(u, v, radius);
      if (vIndices.empty()) continue;
      const Mat dMP = pMP->GetDescriptor();
      int bestDist = INT_MAX, bestIdx = -1;
      for (size_t idx : vIndices) {
        const KeyPoint& kp = pKF1->undistort_keypoints_[idx];
        if (kp.octave < nPredictedLevel - 1 || kp.octave > nPredictedLevel) continue;
        const int dist = DescriptorDistance(dMP, pKF1->descriptors_.row(idx));
        if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
      }
      if (bestDist <= TH_HIGH) vnMatch2[i2] = bestIdx;
    }
    int nFound = 0;
    for (int i1 = 0; i1 < N1; i1++) {
      int idx2 = vnMatch1[i1];
      if (idx2 >= 0) { int idx1 = vnMatch2[idx2]; if (idx1 == i1) { vpMatches12[i1] = vpMapPoints2[idx2]; nFound++; } }
    }
    return nFound;
  }

  int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0) {                  // :724-842
    Matrix3d Rcw = pKF->GetRotation(); Vector3d tcw = pKF->GetTranslation();
    const float &fx = pKF->fx_, &fy = pKF->fy_, &cx = pKF->cx_, &cy = pKF->cy_;
    Vector3d Ow = pKF->GetCameraCenter();
    int nFused = 0;
    const int nMPs = vpMapPoints.size();
    for (int i = 0; i < nMPs; i++) {
      MapPoint* pMP = vpMapPoints[i];
      if (!pMP) continue;
      if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
      Vector3d p3Dw = pMP->GetWorldPos();
      Vector3d p3Dc = Rcw * p3Dw + tcw;
      if (p3Dc[2] < 0.0f) continue;
      const float invz = 1 / p3Dc[2];
      const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
      const float u = fx * x + cx, v = fy * y + cy;
      if (!pKF->IsInImage(u, v)) continue;
      const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
      Vector3d PO = p3Dw - Ow;
      const float dist3D = PO.norm();
      if (dist3D < minDistance || dist3D > maxDistance) continue;
      Vector3d Pn = pMP->GetNormal();
      if (PO.dot(Pn) < 0.5 * dist3D) continue;
      int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
      const float radius = th * pKF->scale_factors_[nPredictedLevel];
      const std::vector<size_t> vIndices = pKF->GetFeaturesInArea(u, v, radius);
      if (vIndices.empty()) continue;
      const Mat dMP = pMP->GetDescriptor();
      int bestDist = 256, bestIdx = -1;
      for (size_t idx : vIndices) {
        const KeyPoint& kp = pKF->undistort_keypoints_[idx];
        const int& kpLevel = kp.octave;
        if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
        const float ex = u - kp.pt.x, ey = v - kp.pt.y;
        const float e2 = ex * ex + ey * ey;
        if (e2 * pKF->inv_level_sigma2s_[kpLevel] > 5.99) continue;
        const int dist = DescriptorDistance(dMP, pKF->descriptors_.row(idx));
        if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
      }
      if (bestDist <= TH_LOW) {
        MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);
        if (pMPinKF) {
          if (!pMPinKF->isBad()) { if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF); else pMPinKF->Replace(pMP); }
        } else { pMP->AddObservation(pKF, bestIdx); pKF->AddMapPoint(pMP, bestIdx); }
        nFused++;
      }
    }
    return nFused;
  }

  int Fuse(KeyFrame* pKF, Matrix4d Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint) {   // :844-954
    const float &fx = pKF->fx_, &fy = pKF->fy_, &cx = pKF->cx_, &cy = pKF->cy_;
    const Dec D = Decompose(Scw);
    const std::set<MapPoint*> spAlreadyFound = pKF->GetMapPoints();
    int nFused = 0;
    const int nPoints = vpPoints.size();
    for (int iMP = 0; iMP < nPoints; iMP++) {
      MapPoint* pMP = vpPoints[iMP];
      if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
      Vector3d p3Dw = pMP->GetWorldPos();
      Vector3d p3Dc = D.Rcw * p3Dw + D.tcw;
      if (p3Dc[2] < 0.0f) continue;
      const float invz = 1.0 / p3Dc[2];
      const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
      const float u = fx * x + cx, v = fy * y + cy;
      if (!pKF->IsInImage(u, v)) continue;
      const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
      Vector3d PO = p3Dw - D.Ow;
      const float dist3D = PO.norm();
      if (dist3D < minDistance || dist3D > maxDistance) continue;
      Vector3d Pn = pMP->GetNormal();
      if (PO.dot(Pn) < 0.5 * dist3D) continue;
      const int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
      const float radius = th * pKF->scale_factors_[nPredictedLevel];
      const std::vector<size_t> vIndices = pKF->GetFeaturesInArea(u, v, radius);
      if (vIndices.empty()) continue;
      const Mat dMP = pMP->GetDescriptor();
      int bestDist = INT_MAX, bestIdx = -1;
      for (size_t idx : vIndices) {
        const int& kpLevel = pKF->undistort_keypoints_[idx].octave;
        if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
        int dist = DescriptorDistance(dMP, pKF->descriptors_.row(idx));
        if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
      }
      if (bestDist <= TH_LOW) {
        MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);
        if (pMPinKF) { if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF; }
        else { pMP->AddObservation(pKF, bestIdx); pKF->AddMapPoint(pMP, bestIdx); }
        nFused++;
      }
    }
    return nFused;
  }
};

// ----------------------------------------------------------------------------------------------------- CeresOptimizer
struct CeresOptimizer {
  static void To7(const Matrix4d& T, double* p7) { double t[16]; for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) t[4 * r + c] = T(r, c); orc_matrix4d_to_pose7(t, p7); }
  static Matrix4d From7(const double* p7) { double t[16]; orc_pose7_to_matrix4d(p7, t); Matrix4d T; for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T(r, c) = t[4 * r + c]; return T; }

  static int PoseOptimization(Frame* frame) {                                                                  // src/CeresOptimizer.cc:275-342
    const int N = frame->N_;
    double pose7[7];
    std::vector<double> Xw, uv; std::vector<float> isg; std::vector<int> slot;
    int n_initial_correspondences = 0;
    std::unique_lock<std::mutex> lock(MapPoint::global_mutex_);
    To7(frame->Tcw_, pose7);
    const double K4[4] = {frame->fx_, frame->fy_, frame->cx_, frame->cy_};
    for (int i = 0; i < N; i++) {
      MapPoint* map_point = frame->map_points_[i];
      if (map_point) {
        n_initial_correspondences++;
        frame->is_outliers_[i] = false;
        Vector3d p = map_point->GetWorldPos();
        const KeyPoint& kp = frame->undistort_keypoints_[i];
        Xw.push_back(p[0]); Xw.push_back(p[1]); Xw.push_back(p[2]); uv.push_back(kp.pt.x); uv.push_back(kp.pt.y);
        isg.push_back(frame->inv_level_sigma2s_[kp.octave]); slot.push_back(i);
      }
    }
    if (n_initial_correspondences < 3) return 0;
    std::vector<uint8_t> out(slot.size());
    const int ninl = orc_pose_optimization(K4, pose7, Xw.data(), uv.data(), isg.data(), (int)slot.size(), out.data(), nullptr);   // Ceres solve + CheckOutliers + normalisation
    for (size_t k = 0; k < slot.size(); k++) frame->is_outliers_[slot[k]] = out[k] != 0;
    lock.unlock();
    frame->SetPose(From7(pose7));
    return ninl;
  }

  static void BundleAdjustment(const std::vector<KeyFrame*>& keyframes, const std::vector<MapPoint*>& map_points, int n_iterations, bool* stop_flag,
                               const unsigned long n_loop_keyframe, const bool is_robust) {                    // :59-225
    std::vector<bool> is_not_optimized_map_point(map_points.size());
    if (keyframes.empty()) return;
    unsigned long max_keyframe_id = 0;
    std::map<KeyFrame*, int> ided_keyframe_pose; std::vector<KeyFrame*> order;
    std::vector<double> K4, poses7; std::vector<uint8_t> fixed;
    for (size_t i = 0; i < keyframes.size(); i++) {
      KeyFrame* keyframe = keyframes[i];
      if (keyframe->isBad()) continue;
      double p7[7]; To7(keyframe->GetPose(), p7);
      ided_keyframe_pose[keyframe] = (int)order.size(); order.push_back(keyframe);
      poses7.insert(poses7.end(), p7, p7 + 7);
      K4.push_back(keyframe->fx_); K4.push_back(keyframe->fy_); K4.push_back(keyframe->cx_); K4.push_back(keyframe->cy_);
      fixed.push_back(keyframe->id_ == 0);
      if (keyframe->id_ > max_keyframe_id) max_keyframe_id = keyframe->id_;
    }
    std::vector<int> pid(map_points.size(), -1);
    std::vector<double> pts3, uv, w; std::vector<int32_t> oc, op; std::vector<uint8_t> rob;
    for (size_t i = 0; i < map_points.size(); i++) {
      MapPoint* map_point = map_points[i];
      if (map_point->isBad()) continue;
      const Vector3d X = map_point->GetWorldPos();
      const std::map<KeyFrame*, size_t> observations = map_point->GetObservations();
      int n_edges = 0;
      for (auto it = observations.begin(); it != observations.end(); it++) {
        KeyFrame* keyframe = it->first;
        if (keyframe->isBad() || keyframe->id_ > max_keyframe_id) continue;
        if (!ided_keyframe_pose.count(keyframe)) continue;
        n_edges++;
        const KeyPoint& kp = keyframe->undistort_keypoints_[it->second];
        oc.push_back(ided_keyframe_pose[keyframe]); op.push_back((int)(pts3.size() / 3));
        uv.push_back(kp.pt.x); uv.push_back(kp.pt.y); w.push_back((double)keyframe->inv_level_sigma2s_[kp.octave]); rob.push_back(is_robust ? 1 : 0);
      }
      if (n_edges == 0) { is_not_optimized_map_point[i] = true; continue; }
      is_not_optimized_map_point[i] = false;
      pid[i] = (int)(pts3.size() / 3);
      pts3.push_back(X[0]); pts3.push_back(X[1]); pts3.push_back(X[2]);
    }
    orc_ba_opts o{n_iterations, std::sqrt(5.991), 0, reinterpret_cast<const volatile uint8_t*>(stop_flag)};
    orc_ba_summary sum;
    orc_ba_solve(K4.data(), poses7.data(), fixed.data(), (int)order.size(), pts3.data(), (int)(pts3.size() / 3), oc.data(), op.data(), uv.data(), w.data(), rob.data(), (int)oc.size(), &o, &sum);
    for (size_t c = 0; c < order.size(); c++) {
      KeyFrame* keyframe = order[c];
      if (keyframe->isBad()) continue;
      Matrix4d pose = From7(&poses7[7 * c]);
      if (n_loop_keyframe == 0) keyframe->SetPose(pose); else { keyframe->global_BA_Tcw_ = pose; keyframe->n_BA_global_for_keyframe_ = n_loop_keyframe; }
    }
    for (size_t i = 0; i < map_points.size(); i++) {
      if (pid[i] < 0) continue;
      MapPoint* map_point = map_points[i];
      if (map_point->isBad()) continue;
      Vector3d X(pts3[3 * pid[i]], pts3[3 * pid[i] + 1], pts3[3 * pid[i] + 2]);
      if (n_loop_keyframe == 0) { map_point->SetWorldPos(X); map_point->UpdateNormalAndDepth(); }
      else { map_point->global_BA_pose_ = X; map_point->n_BA_global_for_keyframe_ = n_loop_keyframe; }
    }
  }
  static void GlobalBundleAdjustemnt(Map* map, int n_iterations = 200, bool* stop_flag = nullptr, const unsigned long n_loop_keyframe = 0, const bool is_robust = true) {
    BundleAdjustment(map->GetAllKeyFrames(), map->GetAllMapPoints(), n_iterations, stop_flag, n_loop_keyframe, is_robust);
  }

  static void LocalBundleAdjustment(KeyFrame* keyframe, bool* stop_flag, Map* map) {                           // :344-599
    std::vector<KeyFrame*> local_order; std::map<KeyFrame*, int> ided_local_keyframes;
    ided_local_keyframes[keyframe] = 0; local_order.push_back(keyframe);
    keyframe->n_BA_local_for_keyframe_ = keyframe->id_;
    const std::vector<KeyFrame*> neighbor_keyframes = keyframe->GetVectorCovisibleKeyFrames();
    for (int i = 0, iend = neighbor_keyframes.size(); i < iend; i++) {
      KeyFrame* nb = neighbor_keyframes[i];
      nb->n_BA_local_for_keyframe_ = keyframe->id_;
      if (!nb->isBad() && !ided_local_keyframes.count(nb)) { ided_local_keyframes[nb] = (int)local_order.size(); local_order.push_back(nb); }
    }
    std::map<MapPoint*, int> ided_local_map_points;
    for (KeyFrame* kf : local_order) {
      std::vector<MapPoint*> map_points = kf->GetMapPointMatches();
      for (MapPoint* map_point : map_points)
        if (map_point) if (!map_point->isBad()) if (map_point->n_BA_local_for_keyframe_ != keyframe->id_) { ided_local_map_points[map_point] = -1; map_point->n_BA_local_for_keyframe_ = keyframe->id_; }
    }
    std::map<KeyFrame*, int> ided_fixed_keyframes;
    for (auto it = ided_local_map_points.begin(); it != ided_local_map_points.end(); it++) {
      std::map<KeyFrame*, size_t> observations = it->first->GetObservations();
      for (auto ob = observations.begin(); ob != observations.end(); ob++) {
        KeyFrame* keyframe_i = ob->first;
        if (keyframe_i->n_BA_local_for_keyframe_ != keyframe->id_ && keyframe_i->n_BA_fixed_for_keyframe_ != keyframe->id_) {
          keyframe_i->n_BA_fixed_for_keyframe_ = keyframe->id_;
          if (!keyframe_i->isBad()) ided_fixed_keyframes[keyframe_i] = -1;
        }
      }
    }
    std::vector<KeyFrame*> cams; std::vector<double> K4, poses7; std::vector<uint8_t> fixed, local;
    std::map<KeyFrame*, int> cam_of;
    auto push = [&](KeyFrame* kf, bool is_local) {
      double p7[7]; To7(kf->GetPose(), p7);
      cam_of[kf] = (int)cams.size(); cams.push_back(kf); poses7.insert(poses7.end(), p7, p7 + 7);
      K4.push_back(kf->fx_); K4.push_back(kf->fy_); K4.push_back(kf->cx_); K4.push_back(kf->cy_);
      local.push_back(is_local); fixed.push_back(!is_local || kf->id_ == 0);
    };
    for (KeyFrame* kf : local_order) push(kf, true);
    for (auto it = ided_fixed_keyframes.begin(); it != ided_fixed_keyframes.end(); it++) push(it->first, false);
    std::vector<MapPoint*> pts; std::vector<double> pts3, uv; std::vector<float> isg; std::vector<int32_t> oc, op;
    std::vector<std::pair<KeyFrame*, MapPoint*> > edge;
    for (auto it = ided_local_map_points.begin(); it != ided_local_map_points.end(); it++) {
      MapPoint* map_point = it->first;
      const Vector3d X = map_point->GetWorldPos();
      const int p = (int)pts.size(); pts.push_back(map_point); pts3.push_back(X[0]); pts3.push_back(X[1]); pts3.push_back(X[2]);
      const std::map<KeyFrame*, size_t> observations = map_point->GetObservations();
      for (auto ob = observations.begin(); ob != observations.end(); ob++) {
        KeyFrame* kf = ob->first;
        if (kf->isBad()) continue;
        if (!cam_of.count(kf)) continue;
        const KeyPoint& kp = kf->undistort_keypoints_[ob->second];
        oc.push_back(cam_of[kf]); op.push_back(p); uv.push_back(kp.pt.x); uv.push_back(kp.pt.y); isg.push_back(kf->inv_level_sigma2s_[kp.octave]);
        edge.push_back(std::make_pair(kf, map_point));
      }
    }
    if (stop_flag) if (*stop_flag) return;
    std::vector<uint8_t> erase(oc.size() + 1, 0);
    orc_ba_summary s1, s2;
    const int aborted = orc_local_ba(K4.data(), poses7.data(), fixed.data(), local.data(), (int)cams.size(), pts3.data(), (int)pts.size(), oc.data(), op.data(), uv.data(),
                                     isg.data(), (int)oc.size(), reinterpret_cast<const volatile uint8_t*>(stop_flag), 1, erase.data(), &s1, &s2);
    if (aborted) return;
    std::unique_lock<std::mutex> lock(map->mutex_map_update_);
    for (size_t i = 0; i < edge.size(); i++) if (erase[i]) { edge[i].first->EraseMapPointMatch(edge[i].second); edge[i].second->EraseObservation(edge[i].first); }
    for (size_t c = 0; c < local_order.size(); c++) local_order[c]->SetPose(From7(&poses7[7 * c]));
    for (size_t p = 0; p < pts.size(); p++) { pts[p]->SetWorldPos(Vector3d(pts3[3 * p], pts3[3 * p + 1], pts3[3 * p + 2])); pts[p]->UpdateNormalAndDepth(); }
  }
};

}  // namespace literal
