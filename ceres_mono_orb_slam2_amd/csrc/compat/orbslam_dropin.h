// ============================================================================
// orbslam_dropin.h -- ORB_SLAM2::ORBmatcher and ORB_SLAM2::CeresOptimizer with the REFERENCE's member-function
// signatures (reference include/ORBmatcher.h:36-97, include/CeresOptimizer.h:351-376), implemented over the C ABI of
// include/orbslam_hip.h.  The call sites in Tracking / LocalMapping / LoopClosing compile unchanged:
//
//     ORBmatcher matcher(0.9, true);
//     int nmatches = matcher.SearchByProjection(current_frame_, last_frame_, th);       // src/Tracking.cc:632
//     CeresOptimizer::PoseOptimization(&current_frame_);                                // src/Tracking.cc:646
//     CeresOptimizer::LocalBundleAdjustment(current_keyframe_, &is_abort_BA_, map_);    // src/LocalMapping.cc:89
//
// The classes are templates over a `Types` bundle naming the reference's own data model (Frame, KeyFrame, MapPoint, Map,
// the Eigen fixed-size types, cv::Mat / cv::KeyPoint / cv::Point2f).  Inside the reference tree
//     #define ORBSLAM_DROPIN_REFERENCE_TYPES       (before including this header; needs Frame.h, KeyFrame.h, MapPoint.h, Map.h)
// instantiates them as ORB_SLAM2::ORBmatcher / ORB_SLAM2::CeresOptimizer; tests/cpp/ instantiates them over mock structs
// that copy the member names of include/Frame.h, KeyFrame.h, MapPoint.h, Map.h.
//
// What runs where: every method walks the pointer graph exactly as the reference does (which map points, validity tests,
// projection geometry with the reference's float / double mix, the mutations Replace / AddObservation / SetPose /
// EraseObservation under the reference's lock scopes) and hands the data-parallel part - frame grid, window candidates,
// Hamming distances; residuals, Jacobians, Schur complement, MFMA Cholesky - to the HIP library in ONE call per method.
// Only element access is used on the math types (M(i, j), v[i]); nothing here needs Eigen to compile.
// ============================================================================
#pragma once
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <algorithm>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../../include/orbslam_hip.h"

namespace ORB_SLAM2 {
namespace dropin {

inline void check(int rc, const char* what) {
  if (rc != 0) throw std::runtime_error(std::string(what) + " failed: " + orbhip_last_error());
}

struct P3 { double x, y, z; };
template <class V> inline P3 p3(const V& v) { return {v[0], v[1], v[2]}; }
// R (3x3 block of any matrix type, row r0 / col c0) * p + t, plain left-to-right double arithmetic
template <class M> inline P3 rot(const M& R, const P3& p) {
  return {R(0, 0) * p.x + R(0, 1) * p.y + R(0, 2) * p.z, R(1, 0) * p.x + R(1, 1) * p.y + R(1, 2) * p.z, R(2, 0) * p.x + R(2, 1) * p.y + R(2, 2) * p.z};
}
inline P3 add(const P3& a, const P3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline P3 sub(const P3& a, const P3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline double dot(const P3& a, const P3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double norm(const P3& a) { return std::sqrt(dot(a, a)); }
struct R33 { double m[3][3]; double operator()(int r, int c) const { return m[r][c]; } };
template <class M> inline R33 block33(const M& T) { R33 R; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R.m[r][c] = T(r, c); return R; }
template <class M> inline P3 col3(const M& T) { return {T(0, 3), T(1, 3), T(2, 3)}; }
inline P3 rot_t(const R33& R, const P3& p) {   // R^T p
  return {R.m[0][0] * p.x + R.m[1][0] * p.y + R.m[2][0] * p.z, R.m[0][1] * p.x + R.m[1][1] * p.y + R.m[2][1] * p.z, R.m[0][2] * p.x + R.m[1][2] * p.y + R.m[2][2] * p.z};
}

// flattened view of one frame / keyframe: what the C ABI takes
struct Flat {
  std::vector<float> kps4; const uint8_t* desc = nullptr; int n = 0; float bounds[4] = {0, 0, 0, 0};
  std::vector<uint8_t> desc_store;
};
template <class F> inline void flatten(const F& f, Flat* o) {          // Frame& or KeyFrame&
  o->n = (int)f.undistort_keypoints_.size();
  o->kps4.resize(4 * (size_t)o->n);
  for (int i = 0; i < o->n; i++) {
    const auto& kp = f.undistort_keypoints_[i];
    o->kps4[4 * i] = kp.pt.x; o->kps4[4 * i + 1] = kp.pt.y; o->kps4[4 * i + 2] = (float)kp.octave; o->kps4[4 * i + 3] = kp.angle;
  }
  o->desc_store.resize(32 * (size_t)o->n);                               // (cv::Mat rows may be padded: copy row by row)
  for (int i = 0; i < o->n; i++) std::memcpy(&o->desc_store[32 * (size_t)i], f.descriptors_.ptr(i), 32);
  o->desc = o->desc_store.data();
  o->bounds[0] = (float)f.min_x_; o->bounds[1] = (float)f.max_x_; o->bounds[2] = (float)f.min_y_; o->bounds[3] = (float)f.max_y_;
}
// DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned int>>) -> ascending node ids + CSR lists
struct FlatFV { std::vector<uint32_t> node, off{0}, idx; };
template <class FV> inline void flatten_fv(const FV& fv, FlatFV* o) {
  for (auto it = fv.begin(); it != fv.end(); ++it) {
    o->node.push_back((uint32_t)it->first);
    for (auto v : it->second) o->idx.push_back((uint32_t)v);
    o->off.push_back((uint32_t)o->idx.size());
  }
}
// per-query arrays of the projection engine
struct Queries {
  std::vector<float> uv, radius, angle; std::vector<int32_t> lo, hi, pred; std::vector<uint8_t> valid, desc;
  explicit Queries(size_t n) : uv(2 * n, 0.f), radius(n, 0.f), angle(n, 0.f), lo(n, -1), hi(n, -1), pred(n, -1), valid(n, 0), desc(32 * n, 0) {}
  template <class MatT> void set_desc(size_t q, const MatT& d) { std::memcpy(&desc[32 * q], d.ptr(0), 32); }
};

}  // namespace dropin

// ============================================================================================== ORBmatcher
template <class Types>
class ORBmatcherT {
 public:
  typedef typename Types::Frame Frame;
  typedef typename Types::KeyFrame KeyFrame;
  typedef typename Types::MapPoint MapPoint;
  typedef typename Types::Map Map;
  typedef typename Types::Matrix3d Matrix3d;
  typedef typename Types::Matrix4d Matrix4d;
  typedef typename Types::Vector3d Vector3d;
  typedef typename Types::Mat Mat;
  typedef typename Types::Point2f Point2f;

  static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;        // src/ORBmatcher.cc:35-37

  ORBmatcherT(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

  // src/ORBmatcher.cc:1422-1437
  static int DescriptorDistance(const Mat& a, const Mat& b) { return orbm_descriptor_distance(a.ptr(0), b.ptr(0)); }

  // ---- Tracking::SearchLocalPoints (src/Tracking.cc:834), src/ORBmatcher.cc:42-119 -------------------------------------
  int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3) {
    using namespace dropin;
    const bool bFactor = th != 1.0;
    const size_t nq = vpMapPoints.size();
    Queries Q(nq);
    for (size_t iMP = 0; iMP < nq; iMP++) {
      MapPoint* pMP = vpMapPoints[iMP];
      if (!pMP->is_track_in_view_) continue;
      if (pMP->isBad()) continue;
      const int nPredictedLevel = pMP->track_scale_level_;
      float r = RadiusByViewingCos(pMP->track_view_cos_);
      if (bFactor) r *= th;
      Q.valid[iMP] = pMP->Observations() > 0 ? 1 : 3;          // (3: once assigned, the feature stays open for later points, ":83-84")
      Q.uv[2 * iMP] = pMP->track_proj_x_; Q.uv[2 * iMP + 1] = pMP->track_proj_y_;
      Q.radius[iMP] = r * F.scale_factors_[nPredictedLevel];
      Q.lo[iMP] = nPredictedLevel - 1; Q.hi[iMP] = nPredictedLevel;
      Q.set_desc(iMP, pMP->GetDescriptor());
    }
    Flat T; flatten(F, &T);
    std::vector<uint8_t> taken(T.n, 0);                       // F.map_points_[idx] holds a point with observations (":83-84")
    for (int i = 0; i < T.n; i++) taken[i] = (F.map_points_[i] && F.map_points_[i]->Observations() > 0) ? 1 : 0;
    std::vector<int32_t> match(nq ? nq : 1, -1);
    int nmatches = 0;
    check(orbm_search_by_projection(T.kps4.data(), T.desc, T.n, T.bounds, Q.uv.data(), Q.radius.data(), Q.lo.data(), Q.hi.data(), nullptr, Q.desc.data(),
                                    Q.valid.data(), nullptr, (int)nq, nullptr, 0.f, taken.data(), 1, mfNNratio, TH_HIGH, 0, match.data(), nullptr, &nmatches),
          "orbm_search_by_projection");
    for (size_t iMP = 0; iMP < nq; iMP++) if (match[iMP] >= 0) F.map_points_[match[iMP]] = vpMapPoints[iMP];
    return nmatches;
  }

  // ---- Tracking::TrackWithMotionModel (src/Tracking.cc:632,638), src/ORBmatcher.cc:1161-1271 ---------------------------
  int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th) {
    using namespace dropin;
    const R33 Rcw = block33(CurrentFrame.Tcw_); const P3 tcw = col3(CurrentFrame.Tcw_);
    const size_t nq = (size_t)LastFrame.N_;
    Queries Q(nq);
    for (size_t i = 0; i < nq; i++) {
      MapPoint* map_point = LastFrame.map_points_[i];
      if (!map_point || LastFrame.is_outliers_[i]) continue;
      const P3 x3Dc = add(rot(Rcw, p3(map_point->GetWorldPos())), tcw);
      const float xc = x3Dc.x, yc = x3Dc.y;
      const float invzc = 1.0 / x3Dc.z;
      if (invzc < 0) continue;
      const float u = CurrentFrame.fx_ * xc * invzc + CurrentFrame.cx_;
      const float v = CurrentFrame.fy_ * yc * invzc + CurrentFrame.cy_;
      if (u < CurrentFrame.min_x_ || u > CurrentFrame.max_x_) continue;
      if (v < CurrentFrame.min_y_ || v > CurrentFrame.max_y_) continue;
      const int nLastOctave = LastFrame.keypoints_[i].octave;
      Q.valid[i] = map_point->Observations() > 0 ? 1 : 3;      // (":1220-1221" looks at the point assigned earlier in this loop, too)
      Q.uv[2 * i] = u; Q.uv[2 * i + 1] = v;
      Q.radius[i] = th * CurrentFrame.scale_factors_[nLastOctave];
      Q.lo[i] = nLastOctave - 1; Q.hi[i] = nLastOctave + 1;
      Q.angle[i] = LastFrame.undistort_keypoints_[i].angle;
      Q.set_desc(i, map_point->GetDescriptor());
    }
    Flat T; flatten(CurrentFrame, &T);
    std::vector<uint8_t> taken(T.n, 0);
    for (int i = 0; i < T.n; i++) taken[i] = (CurrentFrame.map_points_[i] && CurrentFrame.map_points_[i]->Observations() > 0) ? 1 : 0;
    std::vector<int32_t> match(nq ? nq : 1, -1);
    int nmatches = 0;
    check(orbm_search_by_projection(T.kps4.data(), T.desc, T.n, T.bounds, Q.uv.data(), Q.radius.data(), Q.lo.data(), Q.hi.data(), nullptr, Q.desc.data(),
                                    Q.valid.data(), Q.angle.data(), (int)nq, nullptr, 0.f, taken.data(), 0, mfNNratio, TH_HIGH, mbCheckOrientation ? 1 : 0,
                                    match.data(), nullptr, &nmatches), "orbm_search_by_projection");
    // the reference assigns in query order (":1232") and then resets the slots of the removed rotation bins (":1260-1264"):
    // a slot that held a point without observations, or that two queries shared, ends as nullptr when ANY of its matches is removed
    for (size_t i = 0; i < nq; i++) {
      if (match[i] >= 0) CurrentFrame.map_points_[match[i]] = LastFrame.map_points_[i];
      else if (match[i] <= -2) CurrentFrame.map_points_[-2 - match[i]] = LastFrame.map_points_[i];
    }
    for (size_t i = 0; i < nq; i++) if (match[i] <= -2) CurrentFrame.map_points_[-2 - match[i]] = static_cast<MapPoint*>(nullptr);
    return nmatches;
  }

  // ---- Tracking::Relocalization (src/Tracking.cc:1085,1101), src/ORBmatcher.cc:1273-1384 -------------------------------
  int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist) {
    using namespace dropin;
    const R33 Rcw = block33(CurrentFrame.Tcw_); const P3 tcw = col3(CurrentFrame.Tcw_);
    const P3 Rt = rot_t(Rcw, tcw); const P3 Ow = {-Rt.x, -Rt.y, -Rt.z};
    const std::vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
    const size_t nq = vpMPs.size();
    Queries Q(nq);
    for (size_t i = 0; i < nq; i++) {
      MapPoint* pMP = vpMPs[i];
      if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;
      const P3 x3Dw = p3(pMP->GetWorldPos());
      const P3 x3Dc = add(rot(Rcw, x3Dw), tcw);
      const float xc = x3Dc.x, yc = x3Dc.y;
      const float invzc = 1.0 / x3Dc.z;
      const float u = CurrentFrame.fx_ * xc * invzc + CurrentFrame.cx_;
      const float v = CurrentFrame.fy_ * yc * invzc + CurrentFrame.cy_;
      if (u < CurrentFrame.min_x_ || u > CurrentFrame.max_x_) continue;
      if (v < CurrentFrame.min_y_ || v > CurrentFrame.max_y_) continue;
      float dist3D = norm(sub(x3Dw, Ow));
      const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
      if (dist3D < minDistance || dist3D > maxDistance) continue;
      const int nPredictedLevel = pMP->PredictScale(dist3D, &CurrentFrame);
      Q.valid[i] = 1; Q.uv[2 * i] = u; Q.uv[2 * i + 1] = v;
      Q.radius[i] = th * CurrentFrame.scale_factors_[nPredictedLevel];
      Q.lo[i] = nPredictedLevel - 1; Q.hi[i] = nPredictedLevel + 1;
      Q.angle[i] = pKF->undistort_keypoints_[i].angle;
      Q.set_desc(i, pMP->GetDescriptor());
    }
    Flat T; flatten(CurrentFrame, &T);
    std::vector<uint8_t> taken(T.n, 0);
    for (int i = 0; i < T.n; i++) taken[i] = CurrentFrame.map_points_[i] ? 1 : 0;          // (":1336": any map point closes the feature)
    std::vector<int32_t> match(nq ? nq : 1, -1);
    int nmatches = 0;
    check(orbm_search_by_projection(T.kps4.data(), T.desc, T.n, T.bounds, Q.uv.data(), Q.radius.data(), Q.lo.data(), Q.hi.data(), nullptr, Q.desc.data(),
                                    Q.valid.data(), Q.angle.data(), (int)nq, nullptr, 0.f, taken.data(), 0, mfNNratio, ORBdist, mbCheckOrientation ? 1 : 0,
                                    match.data(), nullptr, &nmatches), "orbm_search_by_projection");
    // (every candidate slot was empty, ":1336", so a removed match leaves nullptr behind: assign only the kept ones)
    for (size_t i = 0; i < nq; i++) if (match[i] >= 0) CurrentFrame.map_points_[match[i]] = vpMPs[i];
    return nmatches;
  }

  // ---- LoopClosing::ComputeSim3 (src/LoopClosing.cc:374), src/ORBmatcher.cc:258-361 ------------------------------------
  int SearchByProjection(KeyFrame* pKF, const Matrix4d& Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th) {
    using namespace dropin;
    Sim3Cam C = decompose(Scw);
    std::set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MapPoint*>(nullptr));
    const size_t nq = vpPoints.size();
    Queries Q(nq);
    for (size_t iMP = 0; iMP < nq; iMP++) {
      MapPoint* pMP = vpPoints[iMP];
      if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
      float u, v, dist; int level;
      if (!project_with_gates(pKF, C, pMP, 0.0, true, &u, &v, &dist, &level)) continue;
      Q.valid[iMP] = 1; Q.uv[2 * iMP] = u; Q.uv[2 * iMP + 1] = v;
      Q.radius[iMP] = th * pKF->scale_factors_[level];
      Q.pred[iMP] = level;
      Q.set_desc(iMP, pMP->GetDescriptor());
    }
    Flat T; flatten(*pKF, &T);
    std::vector<uint8_t> taken(T.n, 0);
    for (int i = 0; i < T.n && i < (int)vpMatched.size(); i++) taken[i] = vpMatched[i] ? 1 : 0;
    std::vector<int32_t> match(nq ? nq : 1, -1);
    int nmatches = 0;
    check(orbm_search_by_projection(T.kps4.data(), T.desc, T.n, T.bounds, Q.uv.data(), Q.radius.data(), nullptr, nullptr, Q.pred.data(), Q.desc.data(),
                                    Q.valid.data(), nullptr, (int)nq, nullptr, 0.f, taken.data(), 0, mfNNratio, TH_LOW, 0, match.data(), nullptr, &nmatches),
          "orbm_search_by_projection");
    for (size_t iMP = 0; iMP < nq; iMP++) if (match[iMP] >= 0) vpMatched[match[iMP]] = vpPoints[iMP];
    return nmatches;
  }

  // ---- Tracking::TrackReferenceKeyFrame / Relocalization (src/Tracking.cc:576,1019), src/ORBmatcher.cc:151-256 ----------
  int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) {
    using namespace dropin;
    const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPoint*>(F.N_, static_cast<MapPoint*>(nullptr));
    Flat A, B; flatten(*pKF, &A); flatten(F, &B);
    std::vector<uint8_t> valid1(A.n, 0);
    std::vector<float> a1(A.n), a2(B.n);
    for (int i = 0; i < A.n; i++) { MapPoint* p = i < (int)vpMapPointsKF.size() ? vpMapPointsKF[i] : nullptr; valid1[i] = (p && !p->isBad()) ? 1 : 0; a1[i] = pKF->undistort_keypoints_[i].angle; }
    for (int i = 0; i < B.n; i++) a2[i] = F.keypoints_[i].angle;                         // (":221": the frame's raw keypoint angle)
    FlatFV f1, f2; flatten_fv(pKF->feature_vector_, &f1); flatten_fv(F.feature_vector_, &f2);
    std::vector<int32_t> m12(A.n ? A.n : 1, -1);
    int nmatches = 0;
    check(orbm_search_by_bow(A.desc, A.n, valid1.data(), a1.data(), B.desc, B.n, nullptr, a2.data(), f1.node.data(), f1.off.data(), f1.idx.data(), (int)f1.node.size(),
                             f2.node.data(), f2.off.data(), f2.idx.data(), (int)f2.node.size(), mfNNratio, TH_LOW, 0, mbCheckOrientation ? 1 : 0, m12.data(), &nmatches),
          "orbm_search_by_bow");
    for (int i = 0; i < A.n; i++) if (m12[i] >= 0) vpMapPointMatches[m12[i]] = vpMapPointsKF[i];
    return nmatches;
  }

  // ---- LoopClosing::ComputeSim3 (src/LoopClosing.cc:262), src/ORBmatcher.cc:470-580 ------------------------------------
  int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) {
    using namespace dropin;
    const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = std::vector<MapPoint*>(vpMapPoints1.size(), static_cast<MapPoint*>(nullptr));
    Flat A, B; flatten(*pKF1, &A); flatten(*pKF2, &B);
    std::vector<uint8_t> valid1(A.n, 0), valid2(B.n, 0);
    std::vector<float> a1(A.n), a2(B.n);
    for (int i = 0; i < A.n; i++) { MapPoint* p = i < (int)vpMapPoints1.size() ? vpMapPoints1[i] : nullptr; valid1[i] = (p && !p->isBad()) ? 1 : 0; a1[i] = pKF1->undistort_keypoints_[i].angle; }
    for (int i = 0; i < B.n; i++) { MapPoint* p = i < (int)vpMapPoints2.size() ? vpMapPoints2[i] : nullptr; valid2[i] = (p && !p->isBad()) ? 1 : 0; a2[i] = pKF2->undistort_keypoints_[i].angle; }
    FlatFV f1, f2; flatten_fv(pKF1->feature_vector_, &f1); flatten_fv(pKF2->feature_vector_, &f2);
    std::vector<int32_t> m12(A.n ? A.n : 1, -1);
    int nmatches = 0;
    check(orbm_search_by_bow(A.desc, A.n, valid1.data(), a1.data(), B.desc, B.n, valid2.data(), a2.data(), f1.node.data(), f1.off.data(), f1.idx.data(),
                             (int)f1.node.size(), f2.node.data(), f2.off.data(), f2.idx.data(), (int)f2.node.size(), mfNNratio, TH_LOW, 1,
                             mbCheckOrientation ? 1 : 0, m12.data(), &nmatches), "orbm_search_by_bow");
    for (int i = 0; i < A.n && i < (int)vpMatches12.size(); i++) if (m12[i] >= 0) vpMatches12[i] = vpMapPoints2[m12[i]];
    return nmatches;
  }

  // ---- Tracking::MonocularInitialization (src/Tracking.cc:416), src/ORBmatcher.cc:363-468 ------------------------------
  int SearchForInitialization(Frame& F1, Frame& F2, std::vector<Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10) {
    using namespace dropin;
    Flat A, B; flatten(F1, &A); flatten(F2, &B);
    std::vector<float> pm(2 * (size_t)A.n);
    for (int i = 0; i < A.n; i++) { pm[2 * i] = vbPrevMatched[i].x; pm[2 * i + 1] = vbPrevMatched[i].y; }
    vnMatches12 = std::vector<int>(A.n, -1);
    std::vector<int32_t> m(A.n ? A.n : 1, -1);
    int nmatches = 0;
    check(orbm_search_for_initialization(A.kps4.data(), A.desc, A.n, B.kps4.data(), B.desc, B.n, B.bounds, pm.data(), windowSize, mfNNratio,
                                         mbCheckOrientation ? 1 : 0, m.data(), &nmatches), "orbm_search_for_initialization");
    for (int i = 0; i < A.n; i++) { vnMatches12[i] = m[i]; vbPrevMatched[i].x = pm[2 * i]; vbPrevMatched[i].y = pm[2 * i + 1]; }
    return nmatches;
  }

  // ---- LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:250), src/ORBmatcher.cc:582-722 (monocular) ---------------
  int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, const Matrix3d& F12, std::vector<std::pair<size_t, size_t> >& vMatchedPairs, const bool bOnlyStereo) {
    using namespace dropin;
    vMatchedPairs.clear();
    if (bOnlyStereo) return 0;                                   // no feature of a monocular keyframe has a right coordinate (":626-629")
    // epipole in the second image (":589-596")
    const P3 Cw = p3(pKF1->GetCameraCenter());
    const Matrix3d R2w = pKF2->GetRotation();
    const P3 C2 = add(rot(R2w, Cw), p3(pKF2->GetTranslation()));
    const float invz = 1.0f / C2.z;
    const float ex = pKF2->fx_ * C2.x * invz + pKF2->cx_;
    const float ey = pKF2->fy_ * C2.y * invz + pKF2->cy_;
    Flat A, B; flatten(*pKF1, &A); flatten(*pKF2, &B);
    std::vector<uint8_t> um1(A.n), um2(B.n);
    for (int i = 0; i < A.n; i++) um1[i] = pKF1->GetMapPoint(i) ? 0 : 1;
    for (int i = 0; i < B.n; i++) um2[i] = pKF2->GetMapPoint(i) ? 0 : 1;
    FlatFV f1, f2; flatten_fv(pKF1->feature_vector_, &f1); flatten_fv(pKF2->feature_vector_, &f2);
    double F[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) F[3 * r + c] = F12(r, c);
    std::vector<int32_t> m12(A.n ? A.n : 1, -1);
    int nmatches = 0;
    check(orbm_search_for_triangulation(A.kps4.data(), A.desc, um1.data(), A.n, B.kps4.data(), B.desc, um2.data(), B.n, f1.node.data(), f1.off.data(), f1.idx.data(),
                                        (int)f1.node.size(), f2.node.data(), f2.off.data(), f2.idx.data(), (int)f2.node.size(), F, ex, ey,
                                        pKF2->scale_factors_.data(), pKF2->level_sigma2s_.data(), mbCheckOrientation ? 1 : 0, m12.data(), &nmatches),
          "orbm_search_for_triangulation");
    vMatchedPairs.reserve(nmatches);
    for (int i = 0; i < A.n; i++) if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));
    return nmatches;
  }

  // ---- LoopClosing::ComputeSim3 (src/LoopClosing.cc:319), src/ORBmatcher.cc:956-1159 -----------------------------------
  int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const Matrix3d& R12, const Vector3d& t12, const float th) {
    using namespace dropin;
    const Matrix3d R1w = pKF1->GetRotation(), R2w = pKF2->GetRotation();
    const P3 t1w = p3(pKF1->GetTranslation()), t2w = p3(pKF2->GetTranslation());
    R33 sR12, sR21;                                              // sR12 = s12 * R12, sR21 = (1 / s12) * R12^T  (":973-975")
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { sR12.m[r][c] = s12 * R12(r, c); sR21.m[r][c] = (1.0 / s12) * R12(c, r); }
    const P3 t12p = p3(t12);
    const P3 t21r = rot(sR21, t12p); const P3 t21 = {-t21r.x, -t21r.y, -t21r.z};
    const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
    std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
    for (int i = 0; i < N1; i++) {
      MapPoint* pMP = vpMatches12[i];
      if (pMP) {
        vbAlreadyMatched1[i] = true;
        const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
        if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
      }
    }
    Flat A, B; flatten(*pKF1, &A); flatten(*pKF2, &B);
    Queries Q12(N1), Q21(N2);
    auto one_way = [&](const std::vector<MapPoint*>& pts, const std::vector<bool>& already, const Matrix3d& Raw, const P3& taw, const R33& sRba, const P3& tba,
                       KeyFrame* target, Queries& Q) {
      for (size_t i = 0; i < pts.size(); i++) {
        MapPoint* pMP = pts[i];
        if (!pMP || already[i]) continue;
        if (pMP->isBad()) continue;
        const P3 p3Dca = add(rot(Raw, p3(pMP->GetWorldPos())), taw);
        const P3 p3Dcb = add(rot(sRba, p3Dca), tba);
        if (p3Dcb.z < 0.0) continue;
        const float invz = 1.0 / p3Dcb.z;
        const float x = p3Dcb.x * invz, y = p3Dcb.y * invz;
        const float u = pKF1->fx_ * x + pKF1->cx_, v = pKF1->fy_ * y + pKF1->cy_;          // (both directions use keyframe 1's intrinsics, ":958-961")
        if (!target->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
        const float dist3D = norm(p3Dcb);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int nPredictedLevel = pMP->PredictScale(dist3D, target);
        Q.valid[i] = 1; Q.uv[2 * i] = u; Q.uv[2 * i + 1] = v;
        Q.radius[i] = th * target->scale_factors_[nPredictedLevel];
        Q.pred[i] = nPredictedLevel;
        Q.set_desc(i, pMP->GetDescriptor());
      }
    };
    one_way(vpMapPoints1, vbAlreadyMatched1, R1w, t1w, sR21, t21, pKF2, Q12);
    one_way(vpMapPoints2, vbAlreadyMatched2, R2w, t2w, sR12, t12p, pKF1, Q21);
    // the descriptor rows the C ABI searches with are the MAP POINTS' descriptors (":1036", ":1112"), not the keyframes' own
    std::vector<int32_t> m12(N1 ? N1 : 1, -1);
    int nFound = 0;
    check(orbm_search_by_sim3(A.kps4.data(), A.desc, N1, B.kps4.data(), B.desc, N2, A.bounds, B.bounds, Q12.uv.data(), Q12.radius.data(), Q12.pred.data(),
                              Q12.valid.data(), Q12.desc.data(), Q21.uv.data(), Q21.radius.data(), Q21.pred.data(), Q21.valid.data(), Q21.desc.data(),
                              m12.data(), &nFound), "orbm_search_by_sim3");
    for (int i1 = 0; i1 < N1; i1++) if (m12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[m12[i1]];
    return nFound;
  }

  // ---- LocalMapping::SearchInNeighbors (src/LocalMapping.cc:441,472), src/ORBmatcher.cc:724-842 --------------------------
  int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0) {
    using namespace dropin;
    Sim3Cam C;
    C.R = block33(pKF->GetRotation()); C.t = p3(pKF->GetTranslation()); C.Ow = p3(pKF->GetCameraCenter());
    const size_t nq = vpMapPoints.size();
    Queries Q(nq);
    // candidate selection does not depend on the map mutations below (no `taken` state, ":775-811"): it is computed for
    // every non-null point first; validity (isBad / IsInKeyFrame) is evaluated in the sequential pass, where it can change
    for (size_t i = 0; i < nq; i++) {
      MapPoint* pMP = vpMapPoints[i];
      if (!pMP) continue;
      float u, v, dist; int level;
      if (!project_with_gates(pKF, C, pMP, 0.0f, true, &u, &v, &dist, &level)) continue;
      Q.valid[i] = 1; Q.uv[2 * i] = u; Q.uv[2 * i + 1] = v;
      Q.radius[i] = th * pKF->scale_factors_[level];
      Q.pred[i] = level;
      Q.set_desc(i, pMP->GetDescriptor());
    }
    Flat T; flatten(*pKF, &T);
    std::vector<int32_t> match(nq ? nq : 1, -1);
    int n = 0;
    check(orbm_search_by_projection(T.kps4.data(), T.desc, T.n, T.bounds, Q.uv.data(), Q.radius.data(), nullptr, nullptr, Q.pred.data(), Q.desc.data(),
                                    Q.valid.data(), nullptr, (int)nq, pKF->inv_level_sigma2s_.data(), 5.99f, nullptr, 0, mfNNratio, TH_LOW, 0, match.data(), nullptr, &n),
          "orbm_search_by_projection");
    int nFused = 0;
    for (size_t i = 0; i < nq; i++) {
      MapPoint* pMP = vpMapPoints[i];
      if (!pMP) continue;
      if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
      if (match[i] < 0) continue;
      const int bestIdx = match[i];
      MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);
      if (pMPinKF) {
        if (!pMPinKF->isBad()) {
          if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
          else pMPinKF->Replace(pMP);
        }
      } else {
        pMP->AddObservation(pKF, bestIdx);
        pKF->AddMapPoint(pMP, bestIdx);
      }
      nFused++;
    }
    return nFused;
  }

  // ---- LocalMapping::SearchInNeighbors, the loop `for (target keyframes) matcher.Fuse(neighbor_keyframe, map_point_matches)`
  //      (src/LocalMapping.cc:437-442) with the candidate selection of ALL target keyframes in ONE call (orbl_fuse_batch, round 5).
  // The projection and its gates are evaluated per (keyframe, point) up front - they read only what no Fuse changes (poses,
  // positions, normals, distance bounds) - the validity tests (NULL / isBad / IsInKeyFrame) and the Replace / AddObservation
  // mutation run keyframe after keyframe in the reference's order.  What a mutation CAN change for a later keyframe is a point's
  // descriptor (MapPoint::Replace ends with ComputeDistinctiveDescriptors on the survivor, src/MapPoint.cc:230): a point whose
  // descriptor no longer equals the one the batch searched with is searched again, alone, for the keyframe at hand.
  // Returns the per-keyframe return values of Fuse.
  std::vector<int> Fuse(const std::vector<KeyFrame*>& targets, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0) {
    using namespace dropin;
    const size_t nt = targets.size(), nq = vpMapPoints.size();
    std::vector<int> ret(nt, 0);
    if (!nt || !nq) return ret;
    // orbl_fuse_batch takes ONE inv_level_sigma2 table; keyframes of differently configured extractors (the reference reads each
    // keyframe's own table, src/ORBmatcher.cc:789) keep the reference's loop of single calls
    for (size_t t = 1; t < nt; t++)
      if (targets[t]->inv_level_sigma2s_ != targets[0]->inv_level_sigma2s_) {
        for (size_t k = 0; k < nt; k++) ret[k] = Fuse(targets[k], vpMapPoints, th);
        return ret;
      }
    std::vector<float> uv(2 * nt * nq, 0.f), radius(nt * nq, 0.f); std::vector<int32_t> level(nt * nq, -1);
    std::vector<uint8_t> desc(32 * nq, 0);
    for (size_t i = 0; i < nq; i++) if (vpMapPoints[i]) std::memcpy(&desc[32 * i], vpMapPoints[i]->GetDescriptor().ptr(0), 32);
    std::vector<Flat> T(nt); std::vector<orbl_fuse_keyframe> kf(nt);
    for (size_t t = 0; t < nt; t++) {
      KeyFrame* pKF = targets[t];
      Sim3Cam C;
      C.R = block33(pKF->GetRotation()); C.t = p3(pKF->GetTranslation()); C.Ow = p3(pKF->GetCameraCenter());
      for (size_t i = 0; i < nq; i++) {
        MapPoint* pMP = vpMapPoints[i];
        if (!pMP) continue;
        float u, v, dist; int lv;
        if (!project_with_gates(pKF, C, pMP, 0.0f, true, &u, &v, &dist, &lv)) continue;
        const size_t e = t * nq + i;
        uv[2 * e] = u; uv[2 * e + 1] = v; radius[e] = th * pKF->scale_factors_[lv]; level[e] = lv;
      }
      flatten(*pKF, &T[t]);
      kf[t].kps = T[t].kps4.data(); kf[t].desc = T[t].desc; kf[t].n = T[t].n;
      for (int k = 0; k < 4; k++) kf[t].bounds[k] = T[t].bounds[k];
    }
    std::vector<int32_t> best_idx(nt * nq, -1), best_dist(nt * nq, 256);
    check(orbl_fuse_batch(kf.data(), (int)nt, uv.data(), radius.data(), level.data(), (int)nq, desc.data(), targets[0]->inv_level_sigma2s_.data(),
                          (int)targets[0]->inv_level_sigma2s_.size(), best_idx.data(), best_dist.data()), "orbl_fuse_batch");
    for (size_t t = 0; t < nt; t++) {
      KeyFrame* pKF = targets[t];
      int nFused = 0;
      for (size_t i = 0; i < nq; i++) {
        MapPoint* pMP = vpMapPoints[i];
        if (!pMP) continue;
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        const size_t e = t * nq + i;
        if (level[e] < 0) continue;
        int bestIdx = best_idx[e], bestDist = best_dist[e];
        if (std::memcmp(&desc[32 * i], pMP->GetDescriptor().ptr(0), 32) != 0) {      // the descriptor changed since the batch: this point, this keyframe, again
          Queries Q(1);
          Q.valid[0] = 1; Q.uv[0] = uv[2 * e]; Q.uv[1] = uv[2 * e + 1]; Q.radius[0] = radius[e]; Q.pred[0] = level[e];
          Q.set_desc(0, pMP->GetDescriptor());
          int32_t m1 = -1, d1 = 256; int n1 = 0;
          check(orbm_search_by_projection(T[t].kps4.data(), T[t].desc, T[t].n, T[t].bounds, Q.uv.data(), Q.radius.data(), nullptr, nullptr, Q.pred.data(), Q.desc.data(),
                                          Q.valid.data(), nullptr, 1, pKF->inv_level_sigma2s_.data(), 5.99f, nullptr, 0, mfNNratio, 256, 0, &m1, &d1, &n1),
                "orbm_search_by_projection");
          bestIdx = m1; bestDist = d1;
        }
        if (bestIdx < 0 || bestDist > TH_LOW) continue;
        MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);
        if (pMPinKF) {
          if (!pMPinKF->isBad()) {
            if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
            else pMPinKF->Replace(pMP);
          }
        } else {
          pMP->AddObservation(pKF, bestIdx);
          pKF->AddMapPoint(pMP, bestIdx);
        }
        nFused++;
      }
      ret[t] = nFused;
    }
    return ret;
  }

  // ---- LoopClosing::SearchAndFuse (src/LoopClosing.cc:611), src/ORBmatcher.cc:844-954 ------------------------------------
  int Fuse(KeyFrame* pKF, Matrix4d Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint) {
    using namespace dropin;
    Sim3Cam C = decompose(Scw);
    const std::set<MapPoint*> spAlreadyFound = pKF->GetMapPoints();
    const size_t nq = vpPoints.size();
    Queries Q(nq);
    for (size_t iMP = 0; iMP < nq; iMP++) {
      MapPoint* pMP = vpPoints[iMP];
      if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
      float u, v, dist; int level;
      if (!project_with_gates(pKF, C, pMP, 0.0f, true, &u, &v, &dist, &level)) continue;
      Q.valid[iMP] = 1; Q.uv[2 * iMP] = u; Q.uv[2 * iMP + 1] = v;
      Q.radius[iMP] = th * pKF->scale_factors_[level];
      Q.pred[iMP] = level;
      Q.set_desc(iMP, pMP->GetDescriptor());
    }
    Flat T; flatten(*pKF, &T);
    std::vector<int32_t> match(nq ? nq : 1, -1);
    int n = 0;
    check(orbm_search_by_projection(T.kps4.data(), T.desc, T.n, T.bounds, Q.uv.data(), Q.radius.data(), nullptr, nullptr, Q.pred.data(), Q.desc.data(),
                                    Q.valid.data(), nullptr, (int)nq, nullptr, 0.f, nullptr, 0, mfNNratio, TH_LOW, 0, match.data(), nullptr, &n),
          "orbm_search_by_projection");
    int nFused = 0;
    for (size_t iMP = 0; iMP < nq; iMP++) {
      if (match[iMP] < 0) continue;
      MapPoint* pMP = vpPoints[iMP];
      MapPoint* pMPinKF = pKF->GetMapPoint(match[iMP]);
      if (pMPinKF) {
        if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
      } else {
        pMP->AddObservation(pKF, match[iMP]);
        pKF->AddMapPoint(pMP, match[iMP]);
      }
      nFused++;
    }
    return nFused;
  }

  // ---- LoopClosing::SearchAndFuse, the whole loop (src/LoopClosing.cc:599-630):
  //        for (corrected keyframes) { matcher.Fuse(keyframe, Scw, loop_map_points_, 4, replace); lock map; for (i) if (replace[i]) replace[i]->Replace(loop_map_points_[i]); }
  //      with the candidate selection of ALL keyframes in ONE call (orbl_fuse_batch_sim3, round 6).  Projection and gates are evaluated per
  //      (keyframe, point) up front - they read what no Fuse / Replace changes (the corrected Sim(3), positions, normals, distance bounds) -;
  //      what a keyframe's turn CAN see of the turns before it is re-read when its turn comes: isBad (a loop point that was itself some
  //      keyframe's duplicate has been replaced), GetMapPoints() (Replace hands the replaced point's observations to the loop point: a
  //      later keyframe may already hold it), GetMapPoint(bestIdx), and the loop point's descriptor (Replace ends with
  //      ComputeDistinctiveDescriptors on the survivor, src/MapPoint.cc:230): a point whose descriptor no longer equals the one the
  //      batch searched with is searched again, alone, for the keyframe at hand.  `corrected` in the reference's iteration order
  //      (KeyFrameAndSim3 is a std::map ordered by keyframe pointer), Scw as Sophus::Sim3d::matrix().  Returns Fuse's value per keyframe.
  std::vector<int> SearchAndFuse(const std::vector<std::pair<KeyFrame*, Matrix4d>>& corrected, const std::vector<MapPoint*>& loop_map_points, Map* map, const float th = 4.0) {
    using namespace dropin;
    const size_t nt = corrected.size(), nq = loop_map_points.size();
    std::vector<int> ret(nt, 0);
    if (!nt || !nq) return ret;
    std::vector<float> uv(2 * nt * nq, 0.f), radius(nt * nq, 0.f); std::vector<int32_t> level(nt * nq, -1);
    std::vector<uint8_t> desc(32 * nq, 0);
    for (size_t i = 0; i < nq; i++) std::memcpy(&desc[32 * i], loop_map_points[i]->GetDescriptor().ptr(0), 32);
    std::vector<Flat> T(nt); std::vector<orbl_fuse_keyframe> kf(nt);
    int n_levels = 1;
    for (size_t t = 0; t < nt; t++) {
      KeyFrame* pKF = corrected[t].first;
      const Sim3Cam C = decompose(corrected[t].second);
      n_levels = std::max(n_levels, (int)pKF->scale_factors_.size());
      for (size_t i = 0; i < nq; i++) {
        float u, v, dist; int lv;
        if (!project_with_gates(pKF, C, loop_map_points[i], 0.0f, true, &u, &v, &dist, &lv)) continue;
        const size_t e = t * nq + i;
        uv[2 * e] = u; uv[2 * e + 1] = v; radius[e] = th * pKF->scale_factors_[lv]; level[e] = lv;
      }
      flatten(*pKF, &T[t]);
      kf[t].kps = T[t].kps4.data(); kf[t].desc = T[t].desc; kf[t].n = T[t].n;
      for (int k = 0; k < 4; k++) kf[t].bounds[k] = T[t].bounds[k];
    }
    std::vector<int32_t> best_idx(nt * nq, -1), best_dist(nt * nq, 256);
    check(orbl_fuse_batch_sim3(kf.data(), (int)nt, uv.data(), radius.data(), level.data(), (int)nq, desc.data(), n_levels, best_idx.data(), best_dist.data()),
          "orbl_fuse_batch_sim3");
    for (size_t t = 0; t < nt; t++) {
      KeyFrame* pKF = corrected[t].first;
      std::vector<MapPoint*> replace_map_points(nq, static_cast<MapPoint*>(nullptr));
      const std::set<MapPoint*> spAlreadyFound = pKF->GetMapPoints();
      int nFused = 0;
      for (size_t i = 0; i < nq; i++) {
        MapPoint* pMP = loop_map_points[i];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        const size_t e = t * nq + i;
        if (level[e] < 0) continue;
        int bestIdx = best_idx[e], bestDist = best_dist[e];
        if (std::memcmp(&desc[32 * i], pMP->GetDescriptor().ptr(0), 32) != 0) {      // the descriptor changed since the batch: this point, this keyframe, again
          Queries Q(1);
          Q.valid[0] = 1; Q.uv[0] = uv[2 * e]; Q.uv[1] = uv[2 * e + 1]; Q.radius[0] = radius[e]; Q.pred[0] = level[e];
          Q.set_desc(0, pMP->GetDescriptor());
          int32_t m1 = -1, d1 = 256; int n1 = 0;
          check(orbm_search_by_projection(T[t].kps4.data(), T[t].desc, T[t].n, T[t].bounds, Q.uv.data(), Q.radius.data(), nullptr, nullptr, Q.pred.data(), Q.desc.data(),
                                          Q.valid.data(), nullptr, 1, nullptr, 0.f, nullptr, 0, mfNNratio, 256, 0, &m1, &d1, &n1),
                "orbm_search_by_projection");
          bestIdx = m1; bestDist = d1;
        }
        if (bestIdx < 0 || bestDist > TH_LOW) continue;
        MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);
        if (pMPinKF) {
          if (!pMPinKF->isBad()) replace_map_points[i] = pMPinKF;
        } else {
          pMP->AddObservation(pKF, bestIdx);
          pKF->AddMapPoint(pMP, bestIdx);
        }
        nFused++;
      }
      ret[t] = nFused;
      std::unique_lock<std::mutex> lock(map->mutex_map_update_);     // "Get Map Mutex" (:617)
      for (size_t i = 0; i < nq; i++)
        if (replace_map_points[i]) replace_map_points[i]->Replace(loop_map_points[i]);
    }
    return ret;
  }

 protected:
  float RadiusByViewingCos(const float& viewCos) { return viewCos > 0.998 ? 2.5 : 4.0; }        // src/ORBmatcher.cc:121-126

  struct Sim3Cam { dropin::R33 R; dropin::P3 t, Ow; };
  // "Decompose Scw" (":269-274", ":854-859"): scw from the first row, Rcw = sRcw / scw, tcw = Scw.t / scw, Ow = -Rcw^T tcw
  static Sim3Cam decompose(const Matrix4d& Scw) {
    using namespace dropin;
    Sim3Cam C;
    const R33 sR = block33(Scw);
    const float scw = std::sqrt(sR.m[0][0] * sR.m[0][0] + sR.m[0][1] * sR.m[0][1] + sR.m[0][2] * sR.m[0][2]);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) C.R.m[r][c] = sR.m[r][c] / scw;
    const P3 st = col3(Scw);
    C.t = {st.x / scw, st.y / scw, st.z / scw};
    const P3 Rt = rot_t(C.R, C.t);
    C.Ow = {-Rt.x, -Rt.y, -Rt.z};
    return C;
  }
  // the gates shared by SearchByProjection(KeyFrame*, Scw, ...) and both Fuse overloads (":286-322", ":746-777", ":873-906"):
  // positive depth, inside the image, distance inside the scale-invariance region, viewing angle below 60 degrees
  static bool project_with_gates(KeyFrame* pKF, const Sim3Cam& C, MapPoint* pMP, double, bool viewing_gate, float* u_out, float* v_out, float* dist_out, int* level_out) {
    using namespace dropin;
    const P3 p3Dw = p3(pMP->GetWorldPos());
    const P3 p3Dc = add(rot(C.R, p3Dw), C.t);
    if (p3Dc.z < 0.0) return false;
    const float invz = 1 / p3Dc.z;
    const float x = p3Dc.x * invz, y = p3Dc.y * invz;
    const float u = pKF->fx_ * x + pKF->cx_, v = pKF->fy_ * y + pKF->cy_;
    if (!pKF->IsInImage(u, v)) return false;
    const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
    const P3 PO = sub(p3Dw, C.Ow);
    const float dist = norm(PO);
    if (dist < minDistance || dist > maxDistance) return false;
    if (viewing_gate && dot(PO, p3(pMP->GetNormal())) < 0.5 * dist) return false;
    *u_out = u; *v_out = v; *dist_out = dist;
    *level_out = pMP->PredictScale(dist, pKF);
    return true;
  }

  float mfNNratio;
  bool mbCheckOrientation;
};

// ============================================================================================== CeresOptimizer
template <class Types>
class CeresOptimizerT {
 public:
  typedef typename Types::Frame Frame;
  typedef typename Types::KeyFrame KeyFrame;
  typedef typename Types::MapPoint MapPoint;
  typedef typename Types::Map Map;
  typedef typename Types::Matrix3d Matrix3d;
  typedef typename Types::Matrix4d Matrix4d;
  typedef typename Types::Vector2d Vector2d;
  typedef typename Types::Vector3d Vector3d;
  typedef typename Types::Quaterniond Quaterniond;

  // MatEigenConverter::Matrix4dToMatrix_7_1 / Matrix_7_1_ToMatrix4d (src/MatEigenConverter.cc:66-85)
  static void Matrix4dToMatrix_7_1(const Matrix4d& pose, double out7[7]) {
    double T[16];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T[4 * r + c] = pose(r, c);
    dropin::check(ba_matrix4d_to_pose7(T, out7), "ba_matrix4d_to_pose7");
  }
  static Matrix4d Matrix_7_1_ToMatrix4d(const double in7[7]) {
    double T[16];
    dropin::check(ba_pose7_to_matrix4d(in7, T), "ba_pose7_to_matrix4d");
    Matrix4d pose;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) pose(r, c) = T[4 * r + c];
    return pose;
  }

  // src/CeresOptimizer.cc:227-241
  bool static CheckOutlier(Matrix3d K, Vector2d& observation, float inv_sigma, Vector3d& world_pose, Vector3d& tcw, Quaterniond& qcw, double thres) {
    const double K4[4] = {K(0, 0), K(1, 1), K(0, 2), K(1, 2)};
    const double pose7[7] = {tcw[0], tcw[1], tcw[2], qcw.x(), qcw.y(), qcw.z(), qcw.w()};
    const double X[3] = {world_pose[0], world_pose[1], world_pose[2]}, uv[2] = {observation[0], observation[1]};
    return ba_check_outlier(K4, pose7, X, uv, (double)inv_sigma, thres, nullptr) != 0;
  }

  // src/CeresOptimizer.cc:243-269
  int static CheckOutliers(Frame* frame, Vector3d& tcw, Quaterniond& qcw) {
    int n_bad = 0;
    const double K4[4] = {frame->fx_, frame->fy_, frame->cx_, frame->cy_};
    const double pose7[7] = {tcw[0], tcw[1], tcw[2], qcw.x(), qcw.y(), qcw.z(), qcw.w()};
    for (int i = 0; i < frame->N_; i++) {
      MapPoint* map_point = frame->map_points_[i];
      if (!map_point) continue;
      const Vector3d wp = map_point->GetWorldPos();
      const double X[3] = {wp[0], wp[1], wp[2]};
      const auto& kp = frame->undistort_keypoints_[i];
      const double uv[2] = {kp.pt.x, kp.pt.y};
      const float inv_sigma = frame->inv_level_sigma2s_[kp.octave];
      if (ba_check_outlier(K4, pose7, X, uv, (double)inv_sigma, 5.991, nullptr) != 0) { frame->is_outliers_[i] = true; n_bad++; }
      else frame->is_outliers_[i] = false;
    }
    return n_bad;
  }

  // ---- src/CeresOptimizer.cc:275-342 (Tracking.cc:587,646,684,1074,...) --------------------------------------------------
  int static PoseOptimization(Frame* frame) {
    const int N = frame->N_;
    double pose7[7];
    std::vector<double> Xw, uv; std::vector<float> isg; std::vector<int> slot;
    std::vector<uint8_t> outlier;
    int n_inliers = 0;
    {
      std::unique_lock<std::mutex> lock(MapPoint::global_mutex_);               // (":284")
      Matrix4dToMatrix_7_1(frame->Tcw_, pose7);                                 // frame_tcw, Eigen::Quaterniond(frame_R)
      const double K4[4] = {frame->fx_, frame->fy_, frame->cx_, frame->cy_};
      for (int i = 0; i < N; i++) {
        MapPoint* map_point = frame->map_points_[i];
        if (!map_point) continue;
        frame->is_outliers_[i] = false;
        const Vector3d p = map_point->GetWorldPos();
        const auto& kp = frame->undistort_keypoints_[i];
        Xw.push_back(p[0]); Xw.push_back(p[1]); Xw.push_back(p[2]);
        uv.push_back(kp.pt.x); uv.push_back(kp.pt.y);
        isg.push_back(frame->inv_level_sigma2s_[kp.octave]);
        slot.push_back(i);
      }
      const int n = (int)slot.size();
      if (n < 3) return 0;                                                      // (":330": pose untouched)
      outlier.assign(n, 0);
      dropin::check(ba_pose_optimization(K4, pose7, Xw.data(), uv.data(), isg.data(), n, outlier.data(), &n_inliers, nullptr), "ba_pose_optimization");
      for (int k = 0; k < n; k++) frame->is_outliers_[slot[k]] = outlier[k] != 0;       // CheckOutliers (":333")
    }
    frame->SetPose(Matrix_7_1_ToMatrix4d(pose7));                               // normalized q -> R (":336-340")
    return n_inliers;
  }

  // ---- src/CeresOptimizer.cc:49-57 (Tracking.cc:502, LoopClosing.cc:656) -------------------------------------------------
  void static GlobalBundleAdjustemnt(Map* map, int n_iterations = 200, bool* stop_flag = nullptr, const unsigned long n_loop_keyframe = 0,
                                     const bool is_robust = true) {
    std::vector<KeyFrame*> keyframes = map->GetAllKeyFrames();
    std::vector<MapPoint*> map_points = map->GetAllMapPoints();
    BundleAdjustment(keyframes, map_points, n_iterations, stop_flag, n_loop_keyframe, is_robust);
  }

  // ---- src/CeresOptimizer.cc:59-225 ---------------------------------------------------------------------------------------
  void static BundleAdjustment(const std::vector<KeyFrame*>& keyframes, const std::vector<MapPoint*>& map_points, int n_iterations = 200,
                               bool* stop_flag = nullptr, const unsigned long n_loop_keyframe = 0, const bool is_robust = true) {
    if (keyframes.empty()) return;
    unsigned long max_keyframe_id = 0;
    std::map<KeyFrame*, int> cam_of;                               // ided_keyframe_pose: non-bad keyframes
    std::vector<KeyFrame*> cams;
    std::vector<double> K4, poses7; std::vector<uint8_t> cam_fixed;
    // (cameras by keyframe id - Map::GetAllKeyFrames() hands over a std::set<KeyFrame*>'s pointer order, i.e. whatever the allocator did;
    // by id an odometry-like map gives a banded reduced system, which the library factors at the cost of the band: see LocalBundleAdjustment)
    std::vector<KeyFrame*> kf_by_id(keyframes);
    std::stable_sort(kf_by_id.begin(), kf_by_id.end(), [](const KeyFrame* a, const KeyFrame* b) { return a->id_ < b->id_; });
    for (size_t i = 0; i < kf_by_id.size(); i++) {
      KeyFrame* keyframe = kf_by_id[i];
      if (keyframe->isBad()) continue;
      if (cam_of.count(keyframe)) continue;
      double p7[7];
      Matrix4dToMatrix_7_1(keyframe->GetPose(), p7);
      cam_of[keyframe] = (int)cams.size(); cams.push_back(keyframe);
      poses7.insert(poses7.end(), p7, p7 + 7);
      const double k4[4] = {keyframe->fx_, keyframe->fy_, keyframe->cx_, keyframe->cy_};
      K4.insert(K4.end(), k4, k4 + 4);
      cam_fixed.push_back(keyframe->id_ == 0 ? 1 : 0);            // (":115-120")
      if (keyframe->id_ > max_keyframe_id) max_keyframe_id = keyframe->id_;
    }
    std::vector<int> pt_of(map_points.size(), -1);                 // -1: bad, or no edge (is_not_optimized_map_point)
    std::vector<double> pts3, obs_uv, obs_w; std::vector<int32_t> obs_cam, obs_pt; std::vector<uint8_t> obs_rob;
    for (size_t i = 0; i < map_points.size(); i++) {
      MapPoint* map_point = map_points[i];
      if (map_point->isBad()) continue;
      const Vector3d X = map_point->GetWorldPos();
      const std::map<KeyFrame*, size_t> observations = map_point->GetObservations();
      int n_edges = 0;
      const int pid = (int)(pts3.size() / 3);
      for (auto it = observations.begin(); it != observations.end(); ++it) {
        KeyFrame* keyframe = it->first;
        if (keyframe->isBad() || keyframe->id_ > max_keyframe_id) continue;
        auto c = cam_of.find(keyframe);
        if (c == cam_of.end()) continue;                           // (a keyframe outside `keyframes`: the reference would insert a zero pose here)
        n_edges++;
        const auto& kp = keyframe->undistort_keypoints_[it->second];
        obs_cam.push_back(c->second); obs_pt.push_back(pid);
        obs_uv.push_back(kp.pt.x); obs_uv.push_back(kp.pt.y);
        obs_w.push_back((double)keyframe->inv_level_sigma2s_[kp.octave]);        // sqrt_information = I * invSigma2 (F7)
        obs_rob.push_back(is_robust ? 1 : 0);
      }
      if (n_edges == 0) continue;                                  // (":170-172": RemoveParameterBlock of a never-added block; skipped)
      pt_of[i] = pid;
      pts3.push_back(X[0]); pts3.push_back(X[1]); pts3.push_back(X[2]);
    }
    ba_options o; o.max_iterations = n_iterations; o.huber_delta = std::sqrt(5.991); o.fix_points = 0;
    o.stop_flag = reinterpret_cast<const volatile uint8_t*>(stop_flag);
    dropin::check(ba_solve(K4.data(), poses7.data(), cam_fixed.data(), (int)cams.size(), pts3.data(), (int)(pts3.size() / 3), obs_cam.data(), obs_pt.data(),
                           obs_uv.data(), obs_w.data(), obs_rob.data(), (int)obs_cam.size(), &o, nullptr), "ba_solve");
    for (size_t c = 0; c < cams.size(); c++) {                     // (":194-209")
      KeyFrame* keyframe = cams[c];
      if (keyframe->isBad()) continue;
      const Matrix4d pose = Matrix_7_1_ToMatrix4d(&poses7[7 * c]);
      if (n_loop_keyframe == 0) keyframe->SetPose(pose);
      else { keyframe->global_BA_Tcw_ = pose; keyframe->n_BA_global_for_keyframe_ = n_loop_keyframe; }
    }
    for (size_t i = 0; i < map_points.size(); i++) {               // (":211-224")
      if (pt_of[i] < 0) continue;
      MapPoint* map_point = map_points[i];
      if (map_point->isBad()) continue;
      Vector3d X;
      for (int k = 0; k < 3; k++) X[k] = pts3[3 * (size_t)pt_of[i] + k];
      if (n_loop_keyframe == 0) { map_point->SetWorldPos(X); map_point->UpdateNormalAndDepth(); }
      else { map_point->global_BA_pose_ = X; map_point->n_BA_global_for_keyframe_ = n_loop_keyframe; }
    }
  }

  // ---- src/CeresOptimizer.cc:344-599 (LocalMapping.cc:89) -----------------------------------------------------------------
  void static LocalBundleAdjustment(KeyFrame* keyframe, bool* stop_flag, Map* map) {
    // local keyframes: the current one and its covisibles (":348-363")
    std::vector<KeyFrame*> local_kfs; std::map<KeyFrame*, int> cam_of;
    auto add_cam = [&](KeyFrame* kf, std::vector<KeyFrame*>& list) { if (!cam_of.count(kf)) { cam_of[kf] = -1; list.push_back(kf); } };
    add_cam(keyframe, local_kfs);
    keyframe->n_BA_local_for_keyframe_ = keyframe->id_;
    const std::vector<KeyFrame*> neighbor_keyframes = keyframe->GetVectorCovisibleKeyFrames();
    for (size_t i = 0; i < neighbor_keyframes.size(); i++) {
      KeyFrame* nb = neighbor_keyframes[i];
      nb->n_BA_local_for_keyframe_ = keyframe->id_;
      if (!nb->isBad()) add_cam(nb, local_kfs);
    }
    // local map points seen in local keyframes (":366-384"); std::map = the reference's container (and its iteration order)
    std::map<MapPoint*, int> pt_of;
    for (KeyFrame* kf : local_kfs) {
      const std::vector<MapPoint*> mps = kf->GetMapPointMatches();
      for (MapPoint* mp : mps)
        if (mp && !mp->isBad() && mp->n_BA_local_for_keyframe_ != keyframe->id_) { pt_of[mp] = -1; mp->n_BA_local_for_keyframe_ = keyframe->id_; }
    }
    // fixed keyframes: see local points, are not local (":388-406")
    std::map<KeyFrame*, int> fixed_set;
    for (auto it = pt_of.begin(); it != pt_of.end(); ++it) {
      const std::map<KeyFrame*, size_t> observations = it->first->GetObservations();
      for (auto ob = observations.begin(); ob != observations.end(); ++ob) {
        KeyFrame* kf = ob->first;
        if (kf->n_BA_local_for_keyframe_ != keyframe->id_ && kf->n_BA_fixed_for_keyframe_ != keyframe->id_) {
          kf->n_BA_fixed_for_keyframe_ = keyframe->id_;
          if (!kf->isBad()) fixed_set[kf] = -1;
        }
      }
    }
    // flatten: cameras = local (free unless id 0) then fixed; points in map order; one observation per (point, non-bad keyframe)
    std::vector<KeyFrame*> cams;
    std::vector<double> K4, poses7; std::vector<uint8_t> cam_fixed, cam_local;
    auto push_cam = [&](KeyFrame* kf, bool local) {
      double p7[7];
      Matrix4dToMatrix_7_1(kf->GetPose(), p7);
      cam_of[kf] = (int)cams.size(); cams.push_back(kf);
      poses7.insert(poses7.end(), p7, p7 + 7);
      const double k4[4] = {kf->fx_, kf->fy_, kf->cx_, kf->cy_};
      K4.insert(K4.end(), k4, k4 + 4);
      cam_local.push_back(local ? 1 : 0);
      cam_fixed.push_back((!local || kf->id_ == 0) ? 1 : 0);      // (":476-481", ":499-502")
    };
    // Camera ORDER (round 5).  The reference keeps the local keyframes in an unordered_map keyed by pointer (":348") and Ceres orders the
    // parameter blocks itself, so there is no order to reproduce; the library factors the reduced system in the order given and skips
    // the tiles outside its skyline (INTEGRATION.md "What the keyframe order means for the solver").  By keyframe id - the current
    // keyframe, which shares landmarks with every other one, is the newest and comes LAST - a window along an odometry chain is a band
    // with one dense block row; with the current keyframe first (rounds 1-4) the first block column was dense and so the whole envelope.
    // Deterministic whatever the allocator does; the results are the same to rounding.
    std::vector<KeyFrame*> by_id(local_kfs);
    std::stable_sort(by_id.begin(), by_id.end(), [](const KeyFrame* a, const KeyFrame* b) { return a->id_ < b->id_; });
    for (KeyFrame* kf : by_id) push_cam(kf, true);
    for (auto it = fixed_set.begin(); it != fixed_set.end(); ++it) push_cam(it->first, false);
    std::vector<MapPoint*> pts; std::vector<double> pts3, obs_uv; std::vector<float> obs_isg; std::vector<int32_t> obs_cam, obs_pt;
    std::vector<std::pair<KeyFrame*, MapPoint*>> obs_edge;
    for (auto it = pt_of.begin(); it != pt_of.end(); ++it) {
      MapPoint* mp = it->first;
      const Vector3d X = mp->GetWorldPos();
      it->second = (int)pts.size(); pts.push_back(mp);
      pts3.push_back(X[0]); pts3.push_back(X[1]); pts3.push_back(X[2]);
      const std::map<KeyFrame*, size_t> observations = mp->GetObservations();
      for (auto ob = observations.begin(); ob != observations.end(); ++ob) {
        KeyFrame* kf = ob->first;
        if (kf->isBad()) continue;
        auto c = cam_of.find(kf);
        if (c == cam_of.end() || c->second < 0) continue;          // neither local nor fixed (":465,:483": no residual is added)
        const auto& kp = kf->undistort_keypoints_[ob->second];
        obs_cam.push_back(c->second); obs_pt.push_back(it->second);
        obs_uv.push_back(kp.pt.x); obs_uv.push_back(kp.pt.y);
        obs_isg.push_back(kf->inv_level_sigma2s_[kp.octave]);
        obs_edge.push_back(std::make_pair(kf, mp));
      }
    }
    if (stop_flag && *stop_flag) return;                           // (":509-512")
    std::vector<uint8_t> erase(obs_cam.size() ? obs_cam.size() : 1, 0);
    int aborted = 0;
    dropin::check(ba_local_bundle_adjustment(K4.data(), poses7.data(), cam_fixed.data(), cam_local.data(), (int)cams.size(), pts3.data(), (int)pts.size(),
                                             obs_cam.data(), obs_pt.data(), obs_uv.data(), obs_isg.data(), (int)obs_cam.size(),
                                             reinterpret_cast<const volatile uint8_t*>(stop_flag), 1, erase.data(), &aborted, nullptr, nullptr),
                  "ba_local_bundle_adjustment");
    if (aborted) return;                                           // stop raised before pass 2: the reference returns without writing back
    std::unique_lock<std::mutex> lock(map->mutex_map_update_);     // (":573")
    for (size_t i = 0; i < obs_edge.size(); i++)
      if (erase[i]) { obs_edge[i].first->EraseMapPointMatch(obs_edge[i].second); obs_edge[i].second->EraseObservation(obs_edge[i].first); }
    for (KeyFrame* kf : local_kfs) kf->SetPose(Matrix_7_1_ToMatrix4d(&poses7[7 * (size_t)cam_of[kf]]));          // (":584-590"; the calls in the collection order, the poses from the cameras' places)
    for (size_t p = 0; p < pts.size(); p++) {                      // (":592-598")
      Vector3d X;
      for (int k = 0; k < 3; k++) X[k] = pts3[3 * p + k];
      pts[p]->SetWorldPos(X);
      pts[p]->UpdateNormalAndDepth();
    }
  }

  // ---- src/CeresOptimizer.cc:601-735 (LoopClosing.cc:324) ------------------------------------------------------------------
  // Sim3 = Sophus::Sim3d (or anything whose data() is its storage [qx, qy, qz, qw with |q|^2 = scale, tx, ty, tz]).
  typedef typename Types::Sim3 Sim3;
  typedef typename Types::KeyFrameAndSim3 KeyFrameAndSim3;

  int static OptimizeSim3(KeyFrame* keyframe_1, KeyFrame* keyframe_2, std::vector<MapPoint*>& matches12, Sim3& S12, const float th2, const bool bFixScale) {
    using namespace dropin;
    const double K1[4] = {keyframe_1->fx_, keyframe_1->fy_, keyframe_1->cx_, keyframe_1->cy_};      // keyframe->K_ (":607-608")
    const double K2[4] = {keyframe_2->fx_, keyframe_2->fy_, keyframe_2->cx_, keyframe_2->cy_};
    const Matrix3d R1cw = keyframe_1->GetRotation(), R2cw = keyframe_2->GetRotation();
    const P3 t1cw = p3(keyframe_1->GetTranslation()), t2cw = p3(keyframe_2->GetTranslation());
    const int N = (int)matches12.size();
    const std::vector<MapPoint*> map_points_1 = keyframe_1->GetMapPointMatches();
    std::vector<double> P3D2c, obs1, P3D1c, obs2; std::vector<float> w1, w2;
    for (int i = 0; i < N; i++) {                                      // (":633-692")
      if (!matches12[i]) continue;
      MapPoint* map_point_1 = map_points_1[i];
      MapPoint* map_point_2 = matches12[i];
      const int i2 = map_point_2->GetIndexInKeyFrame(keyframe_2);
      if (!map_point_1 || !map_point_2) continue;
      if (map_point_1->isBad() || map_point_2->isBad() || i2 < 0) continue;
      const auto& keypoint_1 = keyframe_1->undistort_keypoints_[i];
      const auto& keypoint_2 = keyframe_2->undistort_keypoints_[i2];
      const P3 c2 = add(rot(R2cw, p3(map_point_2->GetWorldPos())), t2cw);
      const P3 c1 = add(rot(R1cw, p3(map_point_1->GetWorldPos())), t1cw);
      P3D2c.push_back(c2.x); P3D2c.push_back(c2.y); P3D2c.push_back(c2.z); obs1.push_back(keypoint_1.pt.x); obs1.push_back(keypoint_1.pt.y);
      w1.push_back(keyframe_1->inv_level_sigma2s_[keypoint_1.octave]);
      P3D1c.push_back(c1.x); P3D1c.push_back(c1.y); P3D1c.push_back(c1.z); obs2.push_back(keypoint_2.pt.x); obs2.push_back(keypoint_2.pt.y);
      w2.push_back(keyframe_2->inv_level_sigma2s_[keypoint_2.octave]);
    }
    int n_inliers = 0;
    double s12[7];
    for (int k = 0; k < 7; k++) s12[k] = S12.data()[k];
    const double dummy[3] = {0, 0, 0}; const float fdummy[1] = {0};
    const int n = (int)w1.size();
    check(ba_optimize_sim3(K1, K2, s12, n ? P3D2c.data() : dummy, n ? obs1.data() : dummy, n ? w1.data() : fdummy, n ? P3D1c.data() : dummy, n ? obs2.data() : dummy,
                           n ? w2.data() : fdummy, n, (double)th2, bFixScale ? 1 : 0, nullptr, &n_inliers, nullptr), "ba_optimize_sim3");
    for (int k = 0; k < 7; k++) S12.data()[k] = s12[k];                // S12 = Sim3d::exp(sim12) (":696")
    return n_inliers;                                                  // 0 when fewer than 10 (":731")
  }

  // ---- src/CeresOptimizer.cc:737-957 (LoopClosing.cc:573) -----------------------------------------------------------------
  // The reference indexes its parameter blocks by keyframe id in arrays of max id + 1; here the non-bad keyframes of the map
  // are numbered densely in map order.  An edge to a keyframe without a block (bad, or not in the map: the reference would
  // read an uninitialised 7-vector there) is not added, and bad keyframes are not written back.
  void static OptimizeEssentialGraph(Map* map, KeyFrame* loop_keyframe, KeyFrame* current_keyframe, const KeyFrameAndSim3& keyframes_non_corrected_sim3,
                                     const KeyFrameAndSim3& keyframes_corrected_sim3, const std::map<KeyFrame*, std::set<KeyFrame*> >& loop_connections,
                                     const bool& is_fixed_scale) {
    (void)is_fixed_scale;                                              // (never read by the reference either)
    const int min_weight = 100;
    const std::vector<KeyFrame*> all_keyframes = map->GetAllKeyFrames();
    const std::vector<MapPoint*> all_map_points = map->GetAllMapPoints();
    std::unordered_map<KeyFrame*, int> vtx;
    std::vector<KeyFrame*> kfs;
    std::vector<double> lie; std::vector<uint8_t> fixed;
    for (size_t i = 0; i < all_keyframes.size(); i++) {                // (":769-793")
      KeyFrame* keyframe = all_keyframes[i];
      if (keyframe->isBad() || vtx.count(keyframe)) continue;
      double S[7], x[7];
      auto it = keyframes_corrected_sim3.find(keyframe);
      if (it != keyframes_corrected_sim3.end()) { for (int k = 0; k < 7; k++) S[k] = it->second.data()[k]; }
      else se3_as_sim3(keyframe->GetPose(), S);                        // Sim3d(RxSO3d(1.0, Rcw), tcw)
      dropin::check(ba_sim3_log(S, x), "ba_sim3_log");
      vtx[keyframe] = (int)kfs.size(); kfs.push_back(keyframe);
      lie.insert(lie.end(), x, x + 7);
      fixed.push_back(keyframe == loop_keyframe ? 1 : 0);
    }
    const int n_kf = (int)kfs.size();
    if (n_kf == 0) return;
    const std::vector<double> lie_orig = lie;
    std::vector<int32_t> edge_j, edge_i; std::vector<double> edge_S;
    auto exp_of = [&](int v, double* S) { dropin::check(ba_sim3_exp(&lie_orig[7 * (size_t)v], S), "ba_sim3_exp"); };
    auto add_edge = [&](int vj, int vi, const double* Sjw, const double* Swi) {
      double Sji[7];
      dropin::check(ba_sim3_mul(Sjw, Swi, Sji), "ba_sim3_mul");
      edge_j.push_back(vj); edge_i.push_back(vi); edge_S.insert(edge_S.end(), Sji, Sji + 7);
    };
    // the pose a keyframe enters "normal" edges with: its non-corrected Sim3 if it has one, else exp of its block (":826-833", ":842-847")
    auto world_pose = [&](KeyFrame* kf, int v, double* S) {
      auto it = keyframes_non_corrected_sim3.find(kf);
      if (it != keyframes_non_corrected_sim3.end()) { for (int k = 0; k < 7; k++) S[k] = it->second.data()[k]; }
      else exp_of(v, S);
    };
    std::set<std::pair<unsigned long, unsigned long> > inserted_edges;
    for (auto it = loop_connections.begin(); it != loop_connections.end(); ++it) {       // (":797-821")
      KeyFrame* keyframe = it->first;
      auto vi = vtx.find(keyframe);
      if (vi == vtx.end()) continue;
      const unsigned long id_i = keyframe->id_;
      double Siw[7], Swi[7];
      exp_of(vi->second, Siw);
      dropin::check(ba_sim3_inverse(Siw, Swi), "ba_sim3_inverse");
      for (auto jt = it->second.begin(); jt != it->second.end(); ++jt) {
        const unsigned long id_j = (*jt)->id_;
        if ((id_i != current_keyframe->id_ || id_j != loop_keyframe->id_) && keyframe->GetWeight(*jt) < min_weight) continue;
        auto vj = vtx.find(*jt);
        if (vj == vtx.end()) continue;
        double Sjw[7];
        exp_of(vj->second, Sjw);
        add_edge(vj->second, vi->second, Sjw, Swi);
        inserted_edges.insert(std::make_pair(std::min(id_i, id_j), std::max(id_i, id_j)));
      }
    }
    for (size_t i = 0; i < all_keyframes.size(); i++) {                // (":824-905")
      KeyFrame* keyframe = all_keyframes[i];
      auto vi = vtx.find(keyframe);
      if (vi == vtx.end() || kfs[vi->second] != keyframe) continue;
      double Siw[7], Swi[7];
      world_pose(keyframe, vi->second, Siw);
      dropin::check(ba_sim3_inverse(Siw, Swi), "ba_sim3_inverse");
      KeyFrame* parent_keyframe = keyframe->GetParent();
      if (parent_keyframe) {
        auto vj = vtx.find(parent_keyframe);
        if (vj != vtx.end()) { double Sjw[7]; world_pose(parent_keyframe, vj->second, Sjw); add_edge(vj->second, vi->second, Sjw, Swi); }
      }
      const std::set<KeyFrame*> loop_edges = keyframe->GetLoopEdges();
      for (auto lt = loop_edges.begin(); lt != loop_edges.end(); ++lt) {
        KeyFrame* local_loop_keyframe = *lt;
        if (local_loop_keyframe->id_ < keyframe->id_) {
          auto vl = vtx.find(local_loop_keyframe);
          if (vl == vtx.end()) continue;
          double Slw[7]; world_pose(local_loop_keyframe, vl->second, Slw); add_edge(vl->second, vi->second, Slw, Swi);
        }
      }
      const std::vector<KeyFrame*> connected_keyframes = keyframe->GetCovisiblesByWeight(min_weight);
      for (auto ct = connected_keyframes.begin(); ct != connected_keyframes.end(); ++ct) {
        KeyFrame* keyframe_n = *ct;
        if (keyframe_n && keyframe_n != parent_keyframe && !keyframe->hasChild(keyframe_n) && !loop_edges.count(keyframe_n)) {
          if (!keyframe_n->isBad() && keyframe_n->id_ < keyframe->id_) {
            if (inserted_edges.count(std::make_pair(std::min<unsigned long>(keyframe->id_, keyframe_n->id_), std::max<unsigned long>(keyframe->id_, keyframe_n->id_)))) continue;
            auto vn = vtx.find(keyframe_n);
            if (vn == vtx.end()) continue;
            double Snw[7]; world_pose(keyframe_n, vn->second, Snw); add_edge(vn->second, vi->second, Snw, Swi);
          }
        }
      }
    }
    const int32_t idummy[1] = {0}; const double ddummy[7] = {0, 0, 0, 1, 0, 0, 0};
    const int n_edges = (int)edge_j.size();
    dropin::check(ba_optimize_essential_graph(lie.data(), fixed.data(), n_kf, n_edges ? edge_j.data() : idummy, n_edges ? edge_i.data() : idummy,
                                              n_edges ? edge_S.data() : ddummy, n_edges, 100, nullptr, nullptr), "ba_optimize_essential_graph");
    // write-back (":908-956"): SE(3) recovery per keyframe, map points through their reference keyframe
    std::vector<MapPoint*> pts; std::vector<int32_t> pt_ref; std::vector<double> pts3;
    for (size_t i = 0; i < all_map_points.size(); i++) {
      MapPoint* map_point = all_map_points[i];
      if (map_point->isBad()) continue;
      KeyFrame* ref = nullptr;
      if (map_point->corrected_by_keyframe_ == current_keyframe->id_) {
        for (KeyFrame* k : kfs) if (k->id_ == (unsigned long)map_point->corrected_reference_) { ref = k; break; }
      } else ref = map_point->GetReferenceKeyFrame();
      auto vr = ref ? vtx.find(ref) : vtx.end();
      if (vr == vtx.end()) continue;                                   // (no block for the reference keyframe: left as it is)
      const Vector3d X = map_point->GetWorldPos();
      pts.push_back(map_point); pt_ref.push_back(vr->second); pts3.push_back(X[0]); pts3.push_back(X[1]); pts3.push_back(X[2]);
    }
    std::vector<double> Tiw(12 * (size_t)n_kf);
    dropin::check(ba_essential_graph_correct(lie_orig.data(), lie.data(), n_kf, Tiw.data(), pts.empty() ? idummy : pt_ref.data(), pts.empty() ? nullptr : pts3.data(), (int)pts.size()),
                  "ba_essential_graph_correct");
    std::unique_lock<std::mutex> lock(map->mutex_map_update_);         // (":911")
    for (int v = 0; v < n_kf; v++) {
      Matrix4d T;
      for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) T(r, c) = Tiw[12 * (size_t)v + 4 * r + c];
      T(3, 0) = 0; T(3, 1) = 0; T(3, 2) = 0; T(3, 3) = 1;
      kfs[v]->SetPose(T);
    }
    for (size_t p = 0; p < pts.size(); p++) {
      Vector3d X;
      for (int k = 0; k < 3; k++) X[k] = pts3[3 * p + k];
      pts[p]->SetWorldPos(X);
      pts[p]->UpdateNormalAndDepth();
    }
  }

  // Sophus::Sim3d(Sophus::RxSO3d(1.0, R), t) of a rigid pose, as 7 doubles (Eigen's matrix -> quaternion branches through the pose codec)
  static void se3_as_sim3(const Matrix4d& T, double S[7]) {
    double p7[7];
    Matrix4dToMatrix_7_1(T, p7);
    S[0] = p7[3]; S[1] = p7[4]; S[2] = p7[5]; S[3] = p7[6]; S[4] = p7[0]; S[5] = p7[1]; S[6] = p7[2];
  }
};

// ============================================================================================== Frame-side steps
// The bodies of the reference's member functions on either side of the matcher, over the same Types bundle: a maintainer
// replaces the body of Frame::ComputeBoW / KeyFrame::ComputeBoW / Frame::isInFrustum / Frame::GetFeaturesInArea /
// KeyFrame::GetFeaturesInArea by ONE call, and the per-match body of LocalMapping::CreateNewMapPoints by TriangulateMatches.
template <class Types>
struct FrameOpsT {
  typedef typename Types::Frame Frame;
  typedef typename Types::KeyFrame KeyFrame;
  typedef typename Types::MapPoint MapPoint;
  typedef typename Types::Matrix3d Matrix3d;
  typedef typename Types::Vector3d Vector3d;

  // Frame::ComputeBoW (src/Frame.cc:322-327: only when bow_vector_ is empty) and KeyFrame::ComputeBoW (src/KeyFrame.cc:107-117:
  // when either container is empty): orb_vocabulary_->transform(descriptors, bow_vector_, feature_vector_, 4)
  template <class F> static void ComputeBoW(F& f, orbv_ctx* orb_vocabulary, bool also_when_feature_vector_empty = false) {
    if (!(f.bow_vector_.empty() || (also_when_feature_vector_empty && f.feature_vector_.empty()))) return;
    const int n = (int)f.undistort_keypoints_.size();
    std::vector<uint8_t> desc(32 * (size_t)std::max(n, 1));
    for (int i = 0; i < n; i++) std::memcpy(&desc[32 * (size_t)i], f.descriptors_.ptr(i), 32);
    std::vector<uint32_t> bw(std::max(n, 1)), fn(std::max(n, 1)), fo(n + 2), fi(std::max(n, 1));
    std::vector<double> bv(std::max(n, 1));
    int nw = 0, nf = 0;
    dropin::check(orbv_transform(orb_vocabulary, desc.data(), n, 4, bw.data(), bv.data(), &nw, fn.data(), fo.data(), fi.data(), &nf), "orbv_transform");
    f.bow_vector_.clear(); f.feature_vector_.clear();
    for (int k = 0; k < nw; k++) f.bow_vector_.insert(f.bow_vector_.end(), std::make_pair(bw[k], bv[k]));
    for (int m = 0; m < nf; m++) {
      auto it = f.feature_vector_.insert(f.feature_vector_.end(), std::make_pair(fn[m], std::vector<unsigned int>()));
      it->second.assign(fi.begin() + fo[m], fi.begin() + fo[m + 1]);
    }
  }

  // Frame::isInFrustum (src/Frame.cc:191-241) for a LIST of map points - the loop of Tracking::SearchLocalPoints
  // (src/Tracking.cc:810-826) in one call: sets is_track_in_view_, track_proj_x_, track_proj_y_, track_scale_level_,
  // track_view_cos_ of every point exactly as the member function does and returns the per-point results.
  static std::vector<bool> isInFrustum(Frame& F, const std::vector<MapPoint*>& points, float viewingCosLimit) {
    using namespace dropin;
    const size_t n = points.size();
    std::vector<bool> res(n, false);
    if (!n) return res;
    double R[9], t[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[3 * r + c] = F.Tcw_(r, c); t[r] = F.Tcw_(r, 3); }
    const float K4[4] = {F.fx_, F.fy_, F.cx_, F.cy_}, bounds[4] = {(float)F.min_x_, (float)F.max_x_, (float)F.min_y_, (float)F.max_y_};
    std::vector<double> P(3 * n), Pn(3 * n); std::vector<float> mn(n), mx(n);
    for (size_t i = 0; i < n; i++) {
      const Vector3d p = points[i]->GetWorldPos(), nn = points[i]->GetNormal();
      for (int k = 0; k < 3; k++) { P[3 * i + k] = p[k]; Pn[3 * i + k] = nn[k]; }
      mn[i] = points[i]->GetMinDistanceInvariance(); mx[i] = points[i]->GetMaxDistanceInvariance();
    }
    std::vector<uint8_t> in_view(n); std::vector<float> uv(2 * n), vc(n), dist(n);
    check(orbm_is_in_frustum_gates(R, t, K4, bounds, P.data(), Pn.data(), mn.data(), mx.data(), (int)n, viewingCosLimit, in_view.data(), uv.data(), vc.data(), dist.data()),
          "orbm_is_in_frustum_gates");
    for (size_t i = 0; i < n; i++) {
      MapPoint* mp = points[i];
      mp->is_track_in_view_ = false;                                   // (":192")
      if (!in_view[i]) continue;
      const int nPredictedLevel = mp->PredictScale(dist[i], &F);      // (":231")
      mp->is_track_in_view_ = true; mp->track_proj_x_ = uv[2 * i]; mp->track_proj_y_ = uv[2 * i + 1];
      mp->track_scale_level_ = nPredictedLevel; mp->track_view_cos_ = vc[i];
      res[i] = true;
    }
    return res;
  }
  static bool isInFrustum(Frame& F, MapPoint* map_point, float viewingCosLimit) { return isInFrustum(F, std::vector<MapPoint*>(1, map_point), viewingCosLimit)[0]; }

  // Frame::GetFeaturesInArea (src/Frame.cc:243-307) and KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:575-622, levels -1 / -1)
  template <class F> static std::vector<size_t> GetFeaturesInArea(const F& f, const float& x, const float& y, const float& r, const int minLevel = -1, const int maxLevel = -1) {
    dropin::Flat T; dropin::flatten(f, &T);
    const float q[2] = {x, y};
    const int32_t lo = minLevel, hi = maxLevel;
    std::vector<uint32_t> off(2, 0), idx((size_t)std::max(T.n, 1));
    int total = 0;
    dropin::check(orbm_features_in_area(T.kps4.data(), T.n, T.bounds, q, &r, &lo, &hi, 1, off.data(), idx.data(), (int)idx.size(), &total), "orbm_features_in_area");
    return std::vector<size_t>(idx.begin(), idx.begin() + total);
  }

  // LocalMapping::CreateNewMapPoints, the body of "Triangulate each match" (src/LocalMapping.cc:267-378) for all matches of one
  // neighbour keyframe: x3D[k] / ok[k] for matched_indices[k]; the caller keeps the MapPoint construction (":380-395").
  static void TriangulateMatches(KeyFrame* current_keyframe, KeyFrame* neighbor_keyframe, const std::vector<std::pair<size_t, size_t> >& matched_indices,
                                 const float ratioFactor, std::vector<Vector3d>* x3D, std::vector<bool>* ok) {
    const size_t n = matched_indices.size();
    x3D->assign(n, Vector3d()); ok->assign(n, false);
    if (!n) return;
    double T1[12], T2[12];
    const Matrix3d R1 = current_keyframe->GetRotation(), R2 = neighbor_keyframe->GetRotation();
    const Vector3d t1 = current_keyframe->GetTranslation(), t2 = neighbor_keyframe->GetTranslation();
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) { T1[4 * r + c] = R1(r, c); T2[4 * r + c] = R2(r, c); } T1[4 * r + 3] = t1[r]; T2[4 * r + 3] = t2[r]; }
    const float K1[4] = {current_keyframe->fx_, current_keyframe->fy_, current_keyframe->cx_, current_keyframe->cy_};
    const float K2[4] = {neighbor_keyframe->fx_, neighbor_keyframe->fy_, neighbor_keyframe->cx_, neighbor_keyframe->cy_};
    std::vector<float> kp1(3 * n), kp2(3 * n);
    for (size_t k = 0; k < n; k++) {
      const auto& a = current_keyframe->undistort_keypoints_[matched_indices[k].first];
      const auto& b = neighbor_keyframe->undistort_keypoints_[matched_indices[k].second];
      kp1[3 * k] = a.pt.x; kp1[3 * k + 1] = a.pt.y; kp1[3 * k + 2] = (float)a.octave;
      kp2[3 * k] = b.pt.x; kp2[3 * k + 1] = b.pt.y; kp2[3 * k + 2] = (float)b.octave;
    }
    std::vector<double> X(3 * n); std::vector<uint8_t> good(n);
    dropin::check(orbm_triangulate_matches(T1, T2, K1, K2, kp1.data(), kp2.data(), (int)n, current_keyframe->level_sigma2s_.data(), current_keyframe->scale_factors_.data(),
                                           (int)current_keyframe->scale_factors_.size(), ratioFactor, X.data(), good.data()), "orbm_triangulate_matches");
    for (size_t k = 0; k < n; k++) { (*ok)[k] = good[k] != 0; for (int c = 0; c < 3; c++) (*x3D)[k][c] = X[3 * k + c]; }
  }

  // LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:196-396), everything between the baseline test and the MapPoint construction
  // for ALL neighbours in ONE call (orbl_create_new_map_points): per neighbour k the accepted triangulations {idx1, idx2, x3D} in
  // idx1 order - SearchForTriangulation with ORBmatcher(0.6, false) (:203, :250), the per-match body (:267-378), and the hand-over
  // between neighbours (a keypoint of the current keyframe that got a point is not searched again: AddMapPoint :383 /
  // src/ORBmatcher.cc:621-623).  `neighbours` = the keyframes that passed the baseline test (:231-244), F12s[k] =
  // ComputeF12(current_keyframe_, neighbours[k]) (:247); abort (nullable) = the flag CheckNewKeyFrames() reads (:227): looked at
  // before every neighbour after the first; *n_processed neighbours are complete.  The caller's loop over the result keeps the
  // reference's lines :380-393 (new MapPoint, AddObservation x 2, AddMapPoint x 2, ComputeDistinctiveDescriptors, UpdateNormalAndDepth).
  struct NewPoint { int idx1, idx2; Vector3d x3D; };
  static std::vector<std::vector<NewPoint> > CreateNewMapPoints(KeyFrame* current_keyframe, const std::vector<KeyFrame*>& neighbours, const std::vector<Matrix3d>& F12s,
                                                                const float ratioFactor, const volatile bool* abort = nullptr, int* n_processed = nullptr) {
    using namespace dropin;
    const size_t nn = neighbours.size();
    std::vector<std::vector<NewPoint> > out(nn);
    if (n_processed) *n_processed = 0;
    if (!nn) return out;
    Flat A; flatten(*current_keyframe, &A);
    FlatFV f1; flatten_fv(current_keyframe->feature_vector_, &f1);
    std::vector<uint8_t> um1((size_t)std::max(A.n, 1));
    for (int i = 0; i < A.n; i++) um1[i] = current_keyframe->GetMapPoint(i) ? 0 : 1;
    double T1[12];
    const Matrix3d R1 = current_keyframe->GetRotation(); const Vector3d t1 = current_keyframe->GetTranslation();
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T1[4 * r + c] = R1(r, c); T1[4 * r + 3] = t1[r]; }
    const float K1[4] = {current_keyframe->fx_, current_keyframe->fy_, current_keyframe->cx_, current_keyframe->cy_};
    const P3 Cw = p3(current_keyframe->GetCameraCenter());
    std::vector<Flat> B(nn); std::vector<FlatFV> f2(nn); std::vector<std::vector<uint8_t> > um2(nn);
    std::vector<orbl_keyframe> nb(nn);
    for (size_t k = 0; k < nn; k++) {
      KeyFrame* o = neighbours[k];
      flatten(*o, &B[k]); flatten_fv(o->feature_vector_, &f2[k]);
      um2[k].resize((size_t)std::max(B[k].n, 1));
      for (int i = 0; i < B[k].n; i++) um2[k][i] = o->GetMapPoint(i) ? 0 : 1;
      orbl_keyframe& q = nb[k];
      q.kps = B[k].kps4.data(); q.desc = B[k].desc; q.unmapped = um2[k].data(); q.n = B[k].n;
      q.fv_node = f2[k].node.data(); q.fv_off = f2[k].off.data(); q.fv_idx = f2[k].idx.data(); q.fv_n = (int)f2[k].node.size();
      const Matrix3d R2 = o->GetRotation(); const Vector3d t2 = o->GetTranslation();
      for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) { q.Tcw[4 * r + c] = R2(r, c); q.F12[3 * r + c] = F12s[k](r, c); } q.Tcw[4 * r + 3] = t2[r]; }
      q.K4[0] = o->fx_; q.K4[1] = o->fy_; q.K4[2] = o->cx_; q.K4[3] = o->cy_;
      // epipole in the neighbour's image (src/ORBmatcher.cc:589-596)
      const P3 C2 = add(rot(R2, Cw), p3(t2));
      const float invz = 1.0f / C2.z;
      q.ex = o->fx_ * C2.x * invz + o->cx_; q.ey = o->fy_ * C2.y * invz + o->cy_;
    }
    std::vector<int32_t> m12(nn * (size_t)std::max(A.n, 1)); std::vector<uint8_t> ok(nn * (size_t)std::max(A.n, 1)); std::vector<double> X(3 * nn * (size_t)std::max(A.n, 1));
    // (a bool the reference's thread sets is read as a byte: sizeof(bool) == 1 on every ABI this builds for)
    static_assert(sizeof(bool) == 1, "the abort flag is read as one byte");
    int npr = 0;
    check(orbl_create_new_map_points(A.kps4.data(), A.desc, um1.data(), A.n, f1.node.data(), f1.off.data(), f1.idx.data(), (int)f1.node.size(), T1, K1, nb.data(), (int)nn,
                                     current_keyframe->scale_factors_.data(), current_keyframe->level_sigma2s_.data(), (int)current_keyframe->scale_factors_.size(), ratioFactor,
                                     (const volatile uint8_t*)abort, m12.data(), ok.data(), X.data(), &npr), "orbl_create_new_map_points");
    if (n_processed) *n_processed = npr;
    for (size_t k = 0; k < (size_t)npr; k++)
      for (int i = 0; i < A.n; i++) {
        const size_t e = k * (size_t)A.n + i;
        if (!ok[e]) continue;
        NewPoint np; np.idx1 = i; np.idx2 = m12[e];
        for (int c = 0; c < 3; c++) np.x3D[c] = X[3 * e + c];
        out[k].push_back(np);
      }
    return out;
  }
};

}  // namespace ORB_SLAM2

#ifdef ORBSLAM_DROPIN_REFERENCE_TYPES
// Inside the reference tree (Frame.h, KeyFrame.h, MapPoint.h, Map.h, Eigen and OpenCV already included): the classes the
// call sites name.  src/ORBmatcher.cc and the three CeresOptimizer methods above drop out of the build.
namespace ORB_SLAM2 {
struct ReferenceTypes {
  typedef ORB_SLAM2::Frame Frame; typedef ORB_SLAM2::KeyFrame KeyFrame; typedef ORB_SLAM2::MapPoint MapPoint; typedef ORB_SLAM2::Map Map;
  typedef Eigen::Matrix3d Matrix3d; typedef Eigen::Matrix4d Matrix4d; typedef Eigen::Vector2d Vector2d; typedef Eigen::Vector3d Vector3d;
  typedef Eigen::Quaterniond Quaterniond; typedef cv::Mat Mat; typedef cv::Point2f Point2f;
  typedef Sophus::Sim3d Sim3; typedef LoopClosing::KeyFrameAndSim3 KeyFrameAndSim3;
};
typedef ORBmatcherT<ReferenceTypes> ORBmatcher;
typedef CeresOptimizerT<ReferenceTypes> CeresOptimizerHip;      // every static of include/CeresOptimizer.h:351-388
typedef FrameOpsT<ReferenceTypes> FrameOpsHip;
}  // namespace ORB_SLAM2
#endif
