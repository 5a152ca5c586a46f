// ============================================================================
// orbslam_compat.h -- header-only C++ shims with the REFERENCE's class names and call
// signatures over the C ABI of include/orbslam_hip.h, so the hot path drops into
// Tracking / LocalMapping unchanged:
//     ORB_SLAM2::ORBextractor   (reference include/ORBextractor.h:45-111)
//     ORB_SLAM2::ORBmatcher     (reference include/ORBmatcher.h:36-97; distance core +
//                                SearchForInitialization on flattened frames)
//     ORB_SLAM2::CeresOptimizer (reference include/CeresOptimizer.h:351-376, flattened)
// When OpenCV is available (it is not in this image) the extractor takes cv::InputArray /
// cv::OutputArray exactly like the reference; otherwise minimal PODs with cv::KeyPoint's layout
// are used.  Graph <-> array flattening for the optimizer needs the reference's own
// Frame/KeyFrame/MapPoint classes and is shown as a patch in INTEGRATION.md.
// Nothing here computes: every method forwards to liborbslam_hip.so and throws
// std::runtime_error on a non-zero return (the HIP path has no CPU fallback).
// ============================================================================
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <map>
#include <vector>

#include "../../../include/orbslam_hip.h"

#if defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>)
#include <opencv2/core/core.hpp>
#define ORBCOMPAT_HAVE_OPENCV 1
#endif
#endif

namespace ORB_SLAM2 {

#ifndef ORBCOMPAT_HAVE_OPENCV
namespace cvpod {
struct Point2f { float x = 0, y = 0; };
struct KeyPoint {             // field order and size of cv::KeyPoint (28 bytes)
  Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
};
struct Mat {                  // CV_8UC1 view / owner
  int rows = 0, cols = 0; size_t step = 0; uint8_t* data = nullptr; std::vector<uint8_t> store;
  Mat() {}
  Mat(int r, int c, uint8_t* d, size_t s) : rows(r), cols(c), step(s), data(d) {}
  void create(int r, int c) { rows = r; cols = c; step = (size_t)c; store.assign((size_t)r * c, 0); data = store.data(); }
  void release() { rows = cols = 0; step = 0; data = nullptr; store.clear(); }
  bool empty() const { return data == nullptr || rows <= 0 || cols <= 0; }
  uint8_t* ptr(int r) { return data + (size_t)r * step; }
  const uint8_t* ptr(int r) const { return data + (size_t)r * step; }
};
}  // namespace cvpod
typedef cvpod::KeyPoint KeyPointT;
typedef cvpod::Mat MatT;
typedef cvpod::Point2f Point2fT;
#else
typedef cv::KeyPoint KeyPointT;
typedef cv::Mat MatT;
typedef cv::Point2f Point2fT;
#endif

inline void orbcompat_check(int rc, const char* what) {
  if (rc != 0) throw std::runtime_error(std::string(what) + " failed: " + orbhip_last_error());
}

// ---------------------------------------------------------------------------- ORBextractor
class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0)
      : nlevels_(nlevels), scaleFactor_(scaleFactor) {
    orbcompat_check(orbx_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device, &ctx_), "orbx_create");
    mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    orbcompat_check(orbx_get_tables(ctx_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                                    mvInvLevelSigma2.data(), nullptr), "orbx_get_tables");
    mvImagePyramid.resize(nlevels);
  }
  ~ORBextractor() { orbx_destroy(ctx_); }
  ORBextractor(const ORBextractor&) = delete;
  ORBextractor& operator=(const ORBextractor&) = delete;

  // Compute the ORB features and descriptors on an image; mask is ignored (as in the reference).
#ifdef ORBCOMPAT_HAVE_OPENCV
  void operator()(cv::InputArray _image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray _descriptors) {
    if (_image.empty()) return;
    cv::Mat image = _image.getMat();
    CV_Assert(image.type() == CV_8UC1);
    std::vector<orbx_keypoint> k; std::vector<uint8_t> d; int n = 0;
    run(image.data, image.cols, image.rows, (int)image.step, k, d, n);
    if (n == 0) _descriptors.release();
    else { _descriptors.create(n, 32, CV_8U); std::memcpy(_descriptors.getMat().data, d.data(), (size_t)n * 32); }
    fill(keypoints, k, n);
    fetch_pyramid();
  }
#else
  void operator()(const MatT& image, const MatT& /*mask*/, std::vector<KeyPointT>& keypoints, MatT& descriptors) {
    if (image.empty()) return;                                    // reference: silent return (:1046)
    std::vector<orbx_keypoint> k; std::vector<uint8_t> d; int n = 0;
    run(image.data, image.cols, image.rows, (int)image.step, k, d, n);
    if (n == 0) descriptors.release();
    else { descriptors.create(n, 32); std::memcpy(descriptors.data, d.data(), (size_t)n * 32); }
    fill(keypoints, k, n);
    fetch_pyramid();
  }
#endif

  int inline GetLevels() { return nlevels_; }
  float inline GetScaleFactor() { return scaleFactor_; }
  std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
  std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
  std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
  std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

  std::vector<MatT> mvImagePyramid;      // filled after every call (level images without the 19-px border)
  bool fetch_pyramid_after_call = false; // copying the pyramid back costs PCIe time; opt in

  orbx_ctx* handle() { return ctx_; }

 protected:
  void run(const uint8_t* data, int w, int h, int stride, std::vector<orbx_keypoint>& k, std::vector<uint8_t>& d, int& n) {
    const int cap = orbx_max_keypoints(ctx_);
    k.resize(cap); d.resize((size_t)cap * 32);
    orbcompat_check(orbx_extract(ctx_, data, w, h, stride, k.data(), d.data(), cap, &n), "orbx_extract");
  }
  static void fill(std::vector<KeyPointT>& out, const std::vector<orbx_keypoint>& k, int n) {
    out.clear(); out.resize(n);
    for (int i = 0; i < n; i++) {
      out[i].pt.x = k[i].x; out[i].pt.y = k[i].y; out[i].size = k[i].size; out[i].angle = k[i].angle;
      out[i].response = k[i].response; out[i].octave = k[i].octave; out[i].class_id = k[i].class_id;
    }
  }
  void fetch_pyramid() {
    if (!fetch_pyramid_after_call) return;
    for (int l = 0; l < nlevels_; l++) {
      int w = 0, h = 0;
      orbcompat_check(orbx_get_level_image(ctx_, 0, l, 0, nullptr, &w, &h), "orbx_get_level_image");
#ifdef ORBCOMPAT_HAVE_OPENCV
      mvImagePyramid[l].create(h, w, CV_8UC1);
#else
      mvImagePyramid[l].create(h, w);
#endif
      orbcompat_check(orbx_get_level_image(ctx_, 0, l, 0, mvImagePyramid[l].data, &w, &h), "orbx_get_level_image");
    }
  }
  orbx_ctx* ctx_ = nullptr;
  int nlevels_; float scaleFactor_;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

// ---------------------------------------------------------------------------- ORBmatcher
// The reference's Frame is a host data model outside the hot path; FrameView carries exactly the
// members the flattened matcher entry points read (reference include/Frame.h).
struct FrameView {
  std::vector<KeyPointT> undistort_keypoints_;
  MatT descriptors_;                       // N x 32, CV_8U
  float min_x_ = 0, max_x_ = 0, min_y_ = 0, max_y_ = 0;     // Frame::ComputeImageBounds
};

// Frame-side steps (reference src/Frame.cc:158-173, 191-241, 243-355) on a FrameView: the batched forms the device wants
struct FrameOps {
  static std::vector<float> kps4(const FrameView& F) {
    std::vector<float> k(4 * F.undistort_keypoints_.size());
    for (size_t i = 0; i < F.undistort_keypoints_.size(); i++) {
      const KeyPointT& p = F.undistort_keypoints_[i];
      k[4 * i] = p.pt.x; k[4 * i + 1] = p.pt.y; k[4 * i + 2] = (float)p.octave; k[4 * i + 3] = p.angle;
    }
    return k;
  }
  // Frame::UndistortKeyPoints: keypoints -> undistort_keypoints_ (dist5 = k1, k2, p1, p2, k3)
  static void UndistortKeyPoints(const std::vector<KeyPointT>& keypoints, const float K4[4], const float dist5[5], FrameView& F) {
    const int n = (int)keypoints.size();
    std::vector<float> xy(2 * (size_t)n), out(2 * (size_t)n);
    for (int i = 0; i < n; i++) { xy[2 * i] = keypoints[i].pt.x; xy[2 * i + 1] = keypoints[i].pt.y; }
    orbcompat_check(orbm_undistort_keypoints(xy.data(), n, K4, dist5, out.data()), "orbm_undistort_keypoints");
    F.undistort_keypoints_ = keypoints;
    for (int i = 0; i < n; i++) { F.undistort_keypoints_[i].pt.x = out[2 * i]; F.undistort_keypoints_[i].pt.y = out[2 * i + 1]; }
  }
  // Frame::GetFeaturesInArea for a list of queries (x, y, r, minLevel, maxLevel): indices[q] in the reference's order
  static void GetFeaturesInArea(const FrameView& F, const std::vector<float>& q_xy, const std::vector<float>& q_r, const std::vector<int>& q_min_level,
                                const std::vector<int>& q_max_level, std::vector<std::vector<size_t>>& indices) {
    const int nq = (int)q_r.size(), n = (int)F.undistort_keypoints_.size();
    const std::vector<float> k = kps4(F);
    const float bounds[4] = {F.min_x_, F.max_x_, F.min_y_, F.max_y_};
    std::vector<uint32_t> off(nq + 1, 0), idx;
    int total = 0;
    orbcompat_check(orbm_features_in_area(k.data(), n, bounds, q_xy.data(), q_r.data(), q_min_level.empty() ? nullptr : q_min_level.data(),
                                          q_max_level.empty() ? nullptr : q_max_level.data(), nq, off.data(), nullptr, 0, &total), "orbm_features_in_area");
    idx.resize((size_t)std::max(total, 1));
    orbcompat_check(orbm_features_in_area(k.data(), n, bounds, q_xy.data(), q_r.data(), q_min_level.empty() ? nullptr : q_min_level.data(),
                                          q_max_level.empty() ? nullptr : q_max_level.data(), nq, off.data(), idx.data(), (int)idx.size(), &total), "orbm_features_in_area");
    indices.assign(nq, std::vector<size_t>());
    for (int q = 0; q < nq; q++) indices[q].assign(idx.begin() + off[q], idx.begin() + off[q + 1]);
  }
};

// ORBVocabulary::transform as Frame::ComputeBoW calls it (src/Frame.cc:322-327); BowVector / FeatureVector keep DBoW2's map types
typedef std::map<unsigned int, double> BowVector;
typedef std::map<unsigned int, std::vector<unsigned int>> FeatureVector;
class ORBVocabulary {
 public:
  // flattened tree (what loadFromTextFile builds in m_nodes), see include/orbslam_hip.h::orbv_create
  ORBVocabulary(const uint8_t* node_desc, const uint32_t* child_off, const uint32_t* children, const int32_t* word_id, const double* weight,
                int n_nodes, int L, int device = 0) {
    orbcompat_check(orbv_create(node_desc, child_off, children, word_id, weight, n_nodes, L, device, &ctx_), "orbv_create");
  }
  ~ORBVocabulary() { orbv_destroy(ctx_); }
  ORBVocabulary(const ORBVocabulary&) = delete;
  ORBVocabulary& operator=(const ORBVocabulary&) = delete;
  void transform(const MatT& descriptors, BowVector& v, FeatureVector& fv, int levelsup) const {
    const int n = descriptors.rows;
    std::vector<uint32_t> bw(std::max(n, 1)), fn(std::max(n, 1)), fo(n + 2), fi(std::max(n, 1));
    std::vector<double> bv(std::max(n, 1));
    int nw = 0, nf = 0;
    orbcompat_check(orbv_transform(ctx_, descriptors.data, n, levelsup, bw.data(), bv.data(), &nw, fn.data(), fo.data(), fi.data(), &nf), "orbv_transform");
    v.clear(); fv.clear();
    for (int k = 0; k < nw; k++) v[bw[k]] = bv[k];
    for (int m = 0; m < nf; m++) fv[fn[m]].assign(fi.begin() + fo[m], fi.begin() + fo[m + 1]);
  }
  double score(const BowVector& a, const BowVector& b) const {
    std::vector<uint32_t> wa, wb; std::vector<double> va, vb;
    for (auto& kv : a) { wa.push_back(kv.first); va.push_back(kv.second); }
    for (auto& kv : b) { wb.push_back(kv.first); vb.push_back(kv.second); }
    return orbv_score_l1(wa.data(), va.data(), (int)wa.size(), wb.data(), vb.data(), (int)wb.size());
  }
 private:
  orbv_ctx* ctx_ = nullptr;
};

class ORBmatcher {
 public:
  static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;      // src/ORBmatcher.cc:35-37
  ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

  // Computes the Hamming distance between two ORB descriptors (src/ORBmatcher.cc:1422-1437)
  static int DescriptorDistance(const MatT& a, const MatT& b) { return orbm_descriptor_distance(a.data, b.data); }

  // Matching for the Map Initialization (src/ORBmatcher.cc:363-468)
  int SearchForInitialization(FrameView& F1, FrameView& F2, std::vector<Point2fT>& vbPrevMatched,
                              std::vector<int>& vnMatches12, int windowSize = 10) {
    const int n1 = (int)F1.undistort_keypoints_.size(), n2 = (int)F2.undistort_keypoints_.size();
    std::vector<float> k1 = flatten(F1), k2 = flatten(F2), pm(2 * (size_t)n1);
    for (int i = 0; i < n1; i++) { pm[2 * i] = vbPrevMatched[i].x; pm[2 * i + 1] = vbPrevMatched[i].y; }
    const float bounds[4] = {F2.min_x_, F2.max_x_, F2.min_y_, F2.max_y_};
    vnMatches12.assign(n1, -1);
    int nm = 0;
    orbcompat_check(orbm_search_for_initialization(k1.data(), F1.descriptors_.data, n1, k2.data(), F2.descriptors_.data, n2,
                                                   bounds, pm.data(), windowSize, mfNNratio, mbCheckOrientation ? 1 : 0,
                                                   vnMatches12.data(), &nm), "orbm_search_for_initialization");
    for (int i = 0; i < n1; i++) { vbPrevMatched[i].x = pm[2 * i]; vbPrevMatched[i].y = pm[2 * i + 1]; }
    return nm;
  }

  // best / second-best distances of descriptor rows q against rows t (the inner loop of every Search* method)
  void HammingBest2(const MatT& q, const MatT& t, std::vector<int>& best_idx, std::vector<int>& best_d, std::vector<int>& second_d) {
    best_idx.assign(q.rows, -1); best_d.assign(q.rows, 256); second_d.assign(q.rows, 256);
    orbcompat_check(orbm_hamming_best2(q.data, q.rows, t.data, t.rows, nullptr, nullptr, best_idx.data(), best_d.data(), second_d.data()),
                    "orbm_hamming_best2");
  }

 protected:
  static std::vector<float> flatten(const FrameView& F) {
    std::vector<float> k(4 * F.undistort_keypoints_.size());
    for (size_t i = 0; i < F.undistort_keypoints_.size(); i++) {
      const KeyPointT& kp = F.undistort_keypoints_[i];
      k[4 * i] = kp.pt.x; k[4 * i + 1] = kp.pt.y; k[4 * i + 2] = (float)kp.octave; k[4 * i + 3] = kp.angle;
    }
    return k;
  }
  float mfNNratio;
  bool mbCheckOrientation;
};

// ---------------------------------------------------------------------------- CeresOptimizer (flattened)
struct PoseProblem {          // what PoseOptimization reads from a Frame (src/CeresOptimizer.cc:275-329)
  double K4[4];               // fx, fy, cx, cy
  double pose7[7];            // Tcw as [t, q_xyzw] (src/MatEigenConverter.cc:66-75)
  std::vector<double> Xw, uv; // 3 / 2 doubles per matched map point
  std::vector<float> inv_sigma2;
  std::vector<uint8_t> is_outliers_;      // out
};
struct BAProblem {            // what BundleAdjustment / LocalBundleAdjustment read from the map
  std::vector<double> K4, poses7, pts3, obs_uv;
  std::vector<uint8_t> cam_fixed, cam_local;
  std::vector<int32_t> obs_cam, obs_pt;
  std::vector<float> obs_inv_sigma2;
  std::vector<uint8_t> obs_erase;         // out (LocalBundleAdjustment: to_erase membership)
};
struct Sim3Problem {          // what OptimizeSim3 reads from the two keyframes and matches12 (src/CeresOptimizer.cc:601-692)
  double K1[4], K2[4];
  std::vector<double> P3D2c, obs1, P3D1c, obs2;   // 3 / 2 doubles per accepted correspondence, reference loop order
  std::vector<float> inv_sigma2_1, inv_sigma2_2;
  std::vector<uint8_t> is_outliers_;              // out: is_outlier_12 || is_outlier_21 (:694-726)
};

struct EssentialGraphProblem { // what OptimizeEssentialGraph builds from the map (src/CeresOptimizer.cc:765-905)
  std::vector<double> Scw_datas;            // n x 7 tangents (Scw.log()), in/out
  std::vector<uint8_t> kf_fixed;            // 1 for loop_keyframe
  std::vector<int32_t> edge_j, edge_i;      // AddResidualBlock(cost, nullptr, Scw_datas[id_j], Scw_datas[id_i])
  std::vector<double> edge_Sji;             // ne x 7, Sophus::Sim3d::data() of Sji
};

class CeresOptimizer {
 public:
  // the solve of OptimizeEssentialGraph(Map*, KeyFrame* loop, KeyFrame* current, ...) (:737-914)
  void static OptimizeEssentialGraph(EssentialGraphProblem* p) {
    orbcompat_check(ba_optimize_essential_graph(p->Scw_datas.data(), p->kf_fixed.data(), (int)p->kf_fixed.size(), p->edge_j.data(), p->edge_i.data(),
                                                p->edge_Sji.data(), (int)p->edge_j.size(), 100, nullptr, nullptr), "ba_optimize_essential_graph");
  }
  // int OptimizeSim3(KeyFrame*, KeyFrame*, vector<MapPoint*>& matches12, Sophus::Sim3d& S12, const float th2,
  //                  const bool bFixScale) (:601-735).  S12 = Sophus::Sim3d::data() (7 doubles: scaled q_xyzw, t).
  int static OptimizeSim3(Sim3Problem* p, double* S12, const float th2, const bool bFixScale) {
    const int n = (int)p->inv_sigma2_1.size();
    p->is_outliers_.assign(n, 0);
    int inl = 0;
    orbcompat_check(ba_optimize_sim3(p->K1, p->K2, S12, p->P3D2c.data(), p->obs1.data(), p->inv_sigma2_1.data(), p->P3D1c.data(),
                                     p->obs2.data(), p->inv_sigma2_2.data(), n, (double)th2, bFixScale ? 1 : 0,
                                     p->is_outliers_.data(), &inl, nullptr), "ba_optimize_sim3");
    return inl;
  }
  // int PoseOptimization(Frame*) (src/CeresOptimizer.cc:275-342): returns n_initial - n_bad
  int static PoseOptimization(PoseProblem* f) {
    const int n = (int)f->inv_sigma2.size();
    f->is_outliers_.assign(n, 0);
    int inl = 0;
    orbcompat_check(ba_pose_optimization(f->K4, f->pose7, f->Xw.data(), f->uv.data(), f->inv_sigma2.data(), n,
                                         f->is_outliers_.data(), &inl, nullptr), "ba_pose_optimization");
    return inl;
  }
  // void BundleAdjustment(keyframes, map_points, n_iterations, stop_flag, n_loop_keyframe, is_robust) (:59-225)
  void static BundleAdjustment(BAProblem* p, int n_iterations = 200, bool* stop_flag = nullptr, const bool is_robust = true) {
    const int nobs = (int)p->obs_cam.size(), ncam = (int)p->cam_fixed.size(), npts = (int)p->pts3.size() / 3;
    std::vector<double> w(nobs); std::vector<uint8_t> rob(nobs, is_robust ? 1 : 0);
    for (int i = 0; i < nobs; i++) w[i] = (double)p->obs_inv_sigma2[i];
    ba_options o; o.max_iterations = n_iterations; o.huber_delta = std::sqrt(5.991); o.fix_points = 0;
    o.stop_flag = reinterpret_cast<const volatile uint8_t*>(stop_flag);
    orbcompat_check(ba_solve(p->K4.data(), p->poses7.data(), p->cam_fixed.data(), ncam, p->pts3.data(), npts, p->obs_cam.data(),
                             p->obs_pt.data(), p->obs_uv.data(), w.data(), rob.data(), nobs, &o, nullptr), "ba_solve");
    for (int c = 0; c < ncam; c++) {                     // Matrix_7_1_ToMatrix4d normalises (:196-198)
      double* q = &p->poses7[7 * c + 3];
      const double nq = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      for (int k = 0; k < 4; k++) q[k] /= nq;
    }
  }
  void static GlobalBundleAdjustemnt(BAProblem* map, int n_iterations = 200, bool* stop_flag = nullptr,
                                     const unsigned long /*n_loop_keyframe*/ = 0, const bool is_robust = true) {
    BundleAdjustment(map, n_iterations, stop_flag, is_robust);       // (sic) reference spelling, :49-57
  }
  // void LocalBundleAdjustment(KeyFrame*, bool* stop_flag, Map*) (:344-599): returns silently when *stop_flag is set
  void static LocalBundleAdjustment(BAProblem* p, bool* stop_flag) {
    const int nobs = (int)p->obs_cam.size(), ncam = (int)p->cam_fixed.size(), npts = (int)p->pts3.size() / 3;
    p->obs_erase.assign(nobs, 0);
    int aborted = 0;
    orbcompat_check(ba_local_bundle_adjustment(p->K4.data(), p->poses7.data(), p->cam_fixed.data(), p->cam_local.data(), ncam,
                                               p->pts3.data(), npts, p->obs_cam.data(), p->obs_pt.data(), p->obs_uv.data(),
                                               p->obs_inv_sigma2.data(), nobs, reinterpret_cast<const volatile uint8_t*>(stop_flag),
                                               1, p->obs_erase.data(), &aborted, nullptr, nullptr), "ba_local_bundle_adjustment");
  }
  // bool CheckOutlier(K, observation, inv_sigma, world_pose, tcw, qcw, thres) (:227-241)
  bool static CheckOutlier(const double K4[4], const double uv[2], float inv_sigma, const double Xw[3], const double pose7[7], double thres) {
    return ba_check_outlier(K4, pose7, Xw, uv, (double)inv_sigma, thres, nullptr) != 0;
  }
};

}  // namespace ORB_SLAM2
