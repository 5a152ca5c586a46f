// Mock data model for the drop-in tests: structs that copy the member names (and the small accessor functions) of the
// reference's include/Frame.h, include/KeyFrame.h, include/MapPoint.h, include/Map.h, with tiny stand-ins for the Eigen and
// OpenCV types they use.  TEST INFRASTRUCTURE: the product header csrc/compat/orbslam_dropin.h is instantiated over these;
// tests/cpp/scene_io.h writes their state to disk and tests/dropin_checker.py replays every entry point on it in Python.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <climits>
#include <map>
#include <mutex>
#include <random>
#include <set>
#include <vector>

namespace mock {

struct Vector2d { double v[2] = {0, 0}; Vector2d() {} Vector2d(double a, double b) { v[0] = a; v[1] = b; } double& operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; } };
struct Vector3d {
  double v[3] = {0, 0, 0};
  Vector3d() {} Vector3d(double a, double b, double c) { v[0] = a; v[1] = b; v[2] = c; }
  double& operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; }
  Vector3d operator-(const Vector3d& o) const { return {v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]}; }
  Vector3d operator+(const Vector3d& o) const { return {v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]}; }
  double dot(const Vector3d& o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
  double norm() const { return std::sqrt(dot(*this)); }
};
struct Matrix3d {
  double m[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  double& operator()(int r, int c) { return m[r][c]; } double operator()(int r, int c) const { return m[r][c]; }
  Vector3d operator*(const Vector3d& p) const {
    return {m[0][0] * p[0] + m[0][1] * p[1] + m[0][2] * p[2], m[1][0] * p[0] + m[1][1] * p[1] + m[1][2] * p[2], m[2][0] * p[0] + m[2][1] * p[1] + m[2][2] * p[2]};
  }
  Matrix3d transpose() const { Matrix3d t; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) t.m[r][c] = m[c][r]; return t; }
};
struct Matrix4d {
  double m[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  double& operator()(int r, int c) { return m[r][c]; } double operator()(int r, int c) const { return m[r][c]; }
};
struct Quaterniond { double qx = 0, qy = 0, qz = 0, qw = 1; double x() const { return qx; } double y() const { return qy; } double z() const { return qz; } double w() const { return qw; } };
// Sophus::Sim3d as far as the boundary needs it: data() = [qx, qy, qz, qw (|q|^2 = scale), tx, ty, tz]
struct Sim3d { double d[7] = {0, 0, 0, 1, 0, 0, 0}; double* data() { return d; } const double* data() const { return d; } };

struct Point2f { float x = 0, y = 0; };
struct KeyPoint { Point2f pt; float size = 31, angle = 0, response = 0; int octave = 0, class_id = -1; };
struct Mat {                       // N x 32 CV_8U
  int rows = 0, cols = 32; std::vector<uint8_t> d;
  Mat() {} Mat(int r) : rows(r), d((size_t)r * 32, 0) {}
  uint8_t* ptr(int r) { return d.data() + (size_t)r * 32; } const uint8_t* ptr(int r) const { return d.data() + (size_t)r * 32; }
  Mat row(int r) const { Mat o(1); std::memcpy(o.d.data(), ptr(r), 32); return o; }
};
typedef std::map<unsigned int, std::vector<unsigned int> > FeatureVector;
typedef std::map<unsigned int, double> BowVector;

enum { FRAME_GRID_COLS = 64, FRAME_GRID_ROWS = 48 };
struct KeyFrame; struct Frame; struct Map;

struct MapPoint {
  unsigned long id_ = 0; static unsigned long next_id_;
  Vector3d world_pose_, normal_vector_; Mat descriptor_ = Mat(1);
  std::map<KeyFrame*, size_t> observations_; int n_observations_ = 0;
  bool is_bad_ = false; MapPoint* replaced_map_point_ = nullptr;
  float min_distance_ = 0, max_distance_ = 0;
  // tracking / BA bookkeeping fields the hot path touches
  float track_proj_x_ = 0, track_proj_y_ = 0, track_proj_x_r_ = 0, track_view_cos_ = 0; bool is_track_in_view_ = false; int track_scale_level_ = 0;
  unsigned long n_BA_local_for_keyframe_ = ~0ul, n_BA_global_for_keyframe_ = 0; Vector3d global_BA_pose_;
  int n_update_normal_calls_ = 0;
  unsigned long corrected_by_keyframe_ = ~0ul, corrected_reference_ = 0; KeyFrame* reference_keyframe_ = nullptr;     // loop closing (src/LoopClosing.cc:470-472)
  KeyFrame* GetReferenceKeyFrame() { return reference_keyframe_; }
  static std::mutex global_mutex_;

  Vector3d GetWorldPos() { return world_pose_; } void SetWorldPos(const Vector3d& p) { world_pose_ = p; }
  Vector3d GetNormal() { return normal_vector_; }
  std::map<KeyFrame*, size_t> GetObservations() { return observations_; } int Observations() { return n_observations_; }
  void AddObservation(KeyFrame* kf, size_t idx) { if (observations_.count(kf)) return; observations_[kf] = idx; n_observations_++; }
  void EraseObservation(KeyFrame* kf) { if (observations_.count(kf)) { observations_.erase(kf); n_observations_--; } }
  int GetIndexInKeyFrame(KeyFrame* kf) { auto it = observations_.find(kf); return it == observations_.end() ? -1 : (int)it->second; }
  bool IsInKeyFrame(KeyFrame* kf) { return observations_.count(kf) != 0; }
  bool isBad() { return is_bad_; }
  void Replace(MapPoint* pMP);
  void ComputeDistinctiveDescriptors();                      // src/MapPoint.cc:256-315
  Mat GetDescriptor() { return descriptor_; }
  void UpdateNormalAndDepth() { n_update_normal_calls_++; }
  float GetMinDistanceInvariance() { return 0.8f * min_distance_; } float GetMaxDistanceInvariance() { return 1.2f * max_distance_; }
  template <class F> int PredictScale(const float& current_dist, F* f) {        // src/MapPoint.cc:390-420 (KeyFrame* and Frame* overloads)
    const float ratio = max_distance_ / current_dist;
    int nScale = std::ceil(std::log(ratio) / f->log_scale_factor_);
    if (nScale < 0) nScale = 0; else if (nScale >= f->n_scale_levels_) nScale = f->n_scale_levels_ - 1;
    return nScale;
  }
};

struct GridOwner {                 // the members Frame and KeyFrame share
  int N_ = 0;
  std::vector<KeyPoint> keypoints_, undistort_keypoints_; Mat descriptors_;
  std::vector<MapPoint*> map_points_;
  std::vector<MapPoint*> true_owner_;           // (test only) the map point every feature was generated from, nullptr for noise
  FeatureVector feature_vector_; BowVector bow_vector_;
  int n_scale_levels_ = 8; float scale_factor_ = 1.2f, log_scale_factor_ = std::log(1.2f);
  std::vector<float> scale_factors_, level_sigma2s_, inv_level_sigma2s_;
  std::vector<size_t> grid_[FRAME_GRID_COLS][FRAME_GRID_ROWS];
  float gwinv_ = 0, ghinv_ = 0, gminx_ = 0, gminy_ = 0;
  void build_grid(float min_x, float max_x, float min_y, float max_y) {          // src/Frame.cc:138-141, :158-173, :309-320
    gminx_ = min_x; gminy_ = min_y;
    gwinv_ = static_cast<float>(FRAME_GRID_COLS) / (max_x - min_x); ghinv_ = static_cast<float>(FRAME_GRID_ROWS) / (max_y - min_y);
    for (auto& col : grid_) for (auto& c : col) c.clear();
    for (int i = 0; i < N_; i++) {
      const KeyPoint& kp = undistort_keypoints_[i];
      const int px = std::round((kp.pt.x - gminx_) * gwinv_), py = std::round((kp.pt.y - gminy_) * ghinv_);
      if (px < 0 || px >= FRAME_GRID_COLS || py < 0 || py >= FRAME_GRID_ROWS) continue;
      grid_[px][py].push_back(i);
    }
  }
  std::vector<size_t> features_in_area(const float& x, const float& y, const float& r, const int minLevel, const int maxLevel) const {   // src/Frame.cc:243-307
    std::vector<size_t> indices;
    const int min_cell_x = std::max(0, (int)std::floor((x - gminx_ - r) * gwinv_));
    if (min_cell_x >= FRAME_GRID_COLS) return indices;
    const int max_cell_x = std::min((int)FRAME_GRID_COLS - 1, (int)std::ceil((x - gminx_ + r) * gwinv_));
    if (max_cell_x < 0) return indices;
    const int min_cell_y = std::max(0, (int)std::floor((y - gminy_ - r) * ghinv_));
    if (min_cell_y >= FRAME_GRID_ROWS) return indices;
    const int max_cell_y = std::min((int)FRAME_GRID_ROWS - 1, (int)std::ceil((y - gminy_ + r) * ghinv_));
    if (max_cell_y < 0) return indices;
    const bool do_check_levels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = min_cell_x; ix <= max_cell_x; ix++)
      for (int iy = min_cell_y; iy <= max_cell_y; iy++)
        for (size_t j : grid_[ix][iy]) {
          const KeyPoint& kp = undistort_keypoints_[j];
          if (do_check_levels) {
            if (kp.octave < minLevel) continue;
            if (maxLevel >= 0 && kp.octave > maxLevel) continue;
          }
          const float distx = kp.pt.x - x, disty = kp.pt.y - y;
          if (std::fabs(distx) < r && std::fabs(disty) < r) indices.push_back(j);
        }
    return indices;
  }
};

struct Frame : GridOwner {
  static float fx_, fy_, cx_, cy_, min_x_, max_x_, min_y_, max_y_;
  Matrix4d Tcw_; std::vector<bool> is_outliers_;
  int n_set_pose_calls_ = 0;
  void SetPose(Matrix4d T) { Tcw_ = T; n_set_pose_calls_++; }
  std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, const int minLevel = -1, const int maxLevel = -1) const {
    return features_in_area(x, y, r, minLevel, maxLevel);
  }
};

struct KeyFrame : GridOwner {
  unsigned long id_ = 0; static unsigned long next_id_;
  float fx_ = 0, fy_ = 0, cx_ = 0, cy_ = 0; int min_x_ = 0, min_y_ = 0, max_x_ = 0, max_y_ = 0;
  Matrix4d Tcw_; Vector3d Ow_; bool is_bad_ = false;
  unsigned long n_BA_local_for_keyframe_ = ~0ul, n_BA_fixed_for_keyframe_ = ~0ul, n_BA_global_for_keyframe_ = 0; Matrix4d global_BA_Tcw_;
  std::vector<KeyFrame*> ordered_connected_keyframes_; std::vector<int> ordered_weights_; std::map<KeyFrame*, int> connected_keyframe_weights_;
  KeyFrame* parent_ = nullptr; std::set<KeyFrame*> children_, loop_edges_;                     // spanning tree and loop edges (include/KeyFrame.h)
  KeyFrame* GetParent() { return parent_; } bool hasChild(KeyFrame* kf) { return children_.count(kf) != 0; }
  std::set<KeyFrame*> GetLoopEdges() { return loop_edges_; }
  int GetWeight(KeyFrame* kf) { auto it = connected_keyframe_weights_.find(kf); return it == connected_keyframe_weights_.end() ? 0 : it->second; }
  std::vector<KeyFrame*> GetCovisiblesByWeight(const int& w) {                                 // src/KeyFrame.cc:218-235 (weights descending)
    size_t n = 0; while (n < ordered_weights_.size() && ordered_weights_[n] >= w) n++;
    if (ordered_connected_keyframes_.empty() || n == ordered_weights_.size()) return std::vector<KeyFrame*>();
    return std::vector<KeyFrame*>(ordered_connected_keyframes_.begin(), ordered_connected_keyframes_.begin() + n);
  }
  std::vector<KeyFrame*> GetBestCovisibilityKeyFrames(const int& N) {
    if ((int)ordered_connected_keyframes_.size() < N) return ordered_connected_keyframes_;
    return std::vector<KeyFrame*>(ordered_connected_keyframes_.begin(), ordered_connected_keyframes_.begin() + N);
  }
  int n_set_pose_calls_ = 0;
  void SetPose(const Matrix4d& T) {
    Tcw_ = T; n_set_pose_calls_++;
    Matrix3d R; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R(r, c) = T(r, c);
    const Vector3d t(T(0, 3), T(1, 3), T(2, 3)); const Vector3d o = R.transpose() * t;
    Ow_ = Vector3d(-o[0], -o[1], -o[2]);
  }
  Matrix4d GetPose() { return Tcw_; }
  Matrix3d GetRotation() { Matrix3d R; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R(r, c) = Tcw_(r, c); return R; }
  Vector3d GetTranslation() { return Vector3d(Tcw_(0, 3), Tcw_(1, 3), Tcw_(2, 3)); }
  Vector3d GetCameraCenter() { return Ow_; }
  std::vector<KeyFrame*> GetVectorCovisibleKeyFrames() { return ordered_connected_keyframes_; }
  std::vector<MapPoint*> GetMapPointMatches() { return map_points_; }
  std::set<MapPoint*> GetMapPoints() { std::set<MapPoint*> s; for (MapPoint* p : map_points_) if (p && !p->isBad()) s.insert(p); return s; }
  MapPoint* GetMapPoint(const size_t& i) { return map_points_[i]; }
  void AddMapPoint(MapPoint* p, const size_t& i) { map_points_[i] = p; }
  void EraseMapPointMatch(const size_t& i) { map_points_[i] = nullptr; }
  void EraseMapPointMatch(MapPoint* p) { int i = p->GetIndexInKeyFrame(this); if (i >= 0) map_points_[i] = nullptr; }
  void ReplaceMapPointMatch(const size_t& i, MapPoint* p) { map_points_[i] = p; }
  bool isBad() { return is_bad_; }
  bool IsInImage(const float& x, const float& y) const { return (x >= min_x_ && x < max_x_ && y >= min_y_ && y < max_y_); }
  std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r) const { return features_in_area(x, y, r, -1, -1); }   // src/KeyFrame.cc:575-622
  float ComputeSceneMedianDepth(const int q) {                                                  // src/KeyFrame.cc:624-655
    std::vector<float> depths;
    const double zc = Tcw_(2, 3);
    for (int i = 0; i < N_; i++) if (map_points_[i]) { const Vector3d X = map_points_[i]->GetWorldPos(); depths.push_back((float)(Tcw_(2, 0) * X[0] + Tcw_(2, 1) * X[1] + Tcw_(2, 2) * X[2] + zc)); }
    if (depths.empty()) return -1.0f;
    std::sort(depths.begin(), depths.end());
    return depths[(depths.size() - 1) / q];
  }
};

inline void MapPoint::Replace(MapPoint* pMP) {                // src/MapPoint.cc:185-222 (visible / found counters omitted)
  if (pMP->id_ == this->id_) return;
  std::map<KeyFrame*, size_t> obs = observations_;
  observations_.clear(); is_bad_ = true; replaced_map_point_ = pMP;
  for (auto& o : obs) {
    KeyFrame* kf = o.first;
    if (!pMP->IsInKeyFrame(kf)) { kf->ReplaceMapPointMatch(o.second, pMP); pMP->AddObservation(kf, o.second); }
    else kf->EraseMapPointMatch(o.second);
  }
  pMP->ComputeDistinctiveDescriptors();                       // (:230): the survivor's descriptor may change - later searches see the new one
}

inline void MapPoint::ComputeDistinctiveDescriptors() {       // src/MapPoint.cc:256-315: the observed descriptor with the least median distance to the others
  if (is_bad_ || observations_.empty()) return;
  std::vector<const uint8_t*> ds;
  for (auto& o : observations_) if (!o.first->isBad()) ds.push_back(o.first->descriptors_.ptr((int)o.second));
  if (ds.empty()) return;
  const size_t N = ds.size();
  std::vector<std::vector<int> > dist(N, std::vector<int>(N, 0));
  for (size_t i = 0; i < N; i++)
    for (size_t j = i + 1; j < N; j++) {
      int d = 0;
      for (int k = 0; k < 32; k++) d += __builtin_popcount((unsigned)(ds[i][k] ^ ds[j][k]));
      dist[i][j] = d; dist[j][i] = d;
    }
  int best_median = INT_MAX, best_index = 0;
  for (size_t i = 0; i < N; i++) {
    std::vector<int> v(dist[i]);
    std::sort(v.begin(), v.end());
    const int median = v[(size_t)(0.5 * (N - 1))];
    if (median < best_median) { best_median = median; best_index = (int)i; }
  }
  std::memcpy(descriptor_.ptr(0), ds[best_index], 32);
}

struct Map {
  std::mutex mutex_map_update_;
  std::vector<KeyFrame*> keyframes_; std::vector<MapPoint*> map_points_;
  std::vector<KeyFrame*> GetAllKeyFrames() { return keyframes_; }
  std::vector<MapPoint*> GetAllMapPoints() { return map_points_; }
  long unsigned int GetMaxKFid() { long unsigned int m = 0; for (KeyFrame* k : keyframes_) m = std::max<long unsigned int>(m, k->id_); return m; }
  void AddMapPoint(MapPoint* p) { map_points_.push_back(p); }
};

struct Types {
  typedef mock::Frame Frame; typedef mock::KeyFrame KeyFrame; typedef mock::MapPoint MapPoint; typedef mock::Map Map;
  typedef mock::Matrix3d Matrix3d; typedef mock::Matrix4d Matrix4d; typedef mock::Vector2d Vector2d; typedef mock::Vector3d Vector3d;
  typedef mock::Quaterniond Quaterniond; typedef mock::Mat Mat; typedef mock::Point2f Point2f;
  typedef mock::Sim3d Sim3; typedef std::map<mock::KeyFrame*, mock::Sim3d> KeyFrameAndSim3;
};

// ---------------------------------------------------------------------------------------------------------------- scene
// A small deterministic map: keyframes on a forward trajectory, map points in front of them, observations with pixel noise,
// octaves consistent with PredictScale, descriptors = the point's base descriptor with a few bits flipped per observation.
struct Scene {
  std::vector<KeyFrame> kfs; std::vector<MapPoint> mps; std::vector<Frame> frames; Map map;
  std::vector<float> scale, sigma2, inv_sigma2;
};
inline Matrix4d make_pose(double yaw, double pitch, const Vector3d& C) {       // Tcw from camera centre C and small rotations
  const double cy = std::cos(yaw), sy = std::sin(yaw), cp = std::cos(pitch), sp = std::sin(pitch);
  Matrix3d Ry, Rx; Ry(0, 0) = cy; Ry(0, 2) = sy; Ry(2, 0) = -sy; Ry(2, 2) = cy; Rx(1, 1) = cp; Rx(1, 2) = -sp; Rx(2, 1) = sp; Rx(2, 2) = cp;
  Matrix3d R;
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double s = 0; for (int k = 0; k < 3; k++) s += Rx(r, k) * Ry(k, c); R(r, c) = s; }
  const Vector3d t = R * C;
  Matrix4d T;
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T(r, c) = R(r, c); T(r, 3) = -t[r]; }
  return T;
}
inline void flip_bits(uint8_t* d, int nflip, std::mt19937& rng) { for (int f = 0; f < nflip; f++) d[rng() & 31] ^= (uint8_t)(1u << (rng() & 7)); }

inline void build_scene(Scene& S, unsigned seed, int n_kf = 6, int n_mp = 1500, int n_frames = 2) {
  std::mt19937 rng(seed);
  std::uniform_real_distribution<double> U(0, 1);
  std::normal_distribution<double> G(0, 1);
  const float fx = 718.856f, fy = 718.856f, cx = 607.1928f, cy = 185.2157f; const int W = 1241, H = 376;
  Frame::fx_ = fx; Frame::fy_ = fy; Frame::cx_ = cx; Frame::cy_ = cy; Frame::min_x_ = 0; Frame::max_x_ = W; Frame::min_y_ = 0; Frame::max_y_ = H;
  S.scale.resize(8); S.sigma2.resize(8); S.inv_sigma2.resize(8);
  S.scale[0] = 1.f; for (int l = 1; l < 8; l++) S.scale[l] = S.scale[l - 1] * 1.2f;
  for (int l = 0; l < 8; l++) { S.sigma2[l] = S.scale[l] * S.scale[l]; S.inv_sigma2[l] = 1.f / S.sigma2[l]; }
  MapPoint::next_id_ = 0; KeyFrame::next_id_ = 0;
  S.kfs.resize(n_kf); S.mps.resize(n_mp); S.frames.resize(n_frames);
  for (int p = 0; p < n_mp; p++) {
    MapPoint& mp = S.mps[p];
    mp.id_ = MapPoint::next_id_++;
    const double z = 6 + 45 * U(rng), u = 60 + (W - 120) * U(rng), v = 30 + (H - 60) * U(rng);
    mp.world_pose_ = Vector3d((u - cx) / fx * z + 0.3 * G(rng), (v - cy) / fy * z + 0.1 * G(rng), z + 1.5);
    for (int b = 0; b < 32; b++) mp.descriptor_.d[b] = (uint8_t)rng();
  }
  auto fill_owner = [&](GridOwner& o, const Matrix4d& T, double keep, int n_noise, void* self_kf) {
    o.scale_factors_ = S.scale; o.level_sigma2s_ = S.sigma2; o.inv_level_sigma2s_ = S.inv_sigma2;
    Matrix3d R; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R(r, c) = T(r, c);
    const Vector3d t(T(0, 3), T(1, 3), T(2, 3)); const Vector3d o3 = R.transpose() * t; const Vector3d Ow(-o3[0], -o3[1], -o3[2]);
    std::vector<KeyPoint> kps; std::vector<MapPoint*> owner; std::vector<std::vector<uint8_t> > descs;
    for (MapPoint& mp : S.mps) {
      const Vector3d pc = R * mp.world_pose_ + t;
      if (pc[2] < 1.0) continue;
      const float u = fx * pc[0] / pc[2] + cx, v = fy * pc[1] / pc[2] + cy;
      if (u < 20 || u > W - 20 || v < 20 || v > H - 20) continue;
      if (U(rng) > keep) continue;
      const float dist = (mp.world_pose_ - Ow).norm();
      if (mp.max_distance_ == 0) {                              // first observer defines the invariance region (UpdateNormalAndDepth)
        const int lvl = (int)(rng() % 4);
        mp.max_distance_ = dist * S.scale[lvl]; mp.min_distance_ = mp.max_distance_ / S.scale[7];
        const Vector3d n = mp.world_pose_ - Ow; const double nn = n.norm(); mp.normal_vector_ = Vector3d(n[0] / nn, n[1] / nn, n[2] / nn);
      }
      if (dist < mp.GetMinDistanceInvariance() || dist > mp.GetMaxDistanceInvariance()) continue;
      KeyPoint kp; kp.octave = mp.PredictScale(dist, &o);
      if (kp.octave > 0 && (rng() & 3) == 0) kp.octave--;       // (detected one level below the prediction now and then)
      kp.pt.x = u + (float)(0.6 * G(rng)) * S.scale[kp.octave]; kp.pt.y = v + (float)(0.6 * G(rng)) * S.scale[kp.octave];
      kp.angle = (float)std::fmod(37.0 * mp.id_ + 3.0 * G(rng) + 720.0, 360.0);
      std::vector<uint8_t> d(mp.descriptor_.d.begin(), mp.descriptor_.d.begin() + 32);
      flip_bits(d.data(), 6 + (int)(rng() % 10), rng);
      kps.push_back(kp); owner.push_back(&mp); descs.push_back(d);
    }
    for (int k = 0; k < n_noise; k++) {                          // features without a map point
      KeyPoint kp; kp.pt.x = (float)(20 + (W - 40) * U(rng)); kp.pt.y = (float)(20 + (H - 40) * U(rng)); kp.octave = (int)(rng() % 8); kp.angle = (float)(360 * U(rng));
      std::vector<uint8_t> d(32); for (auto& b : d) b = (uint8_t)rng();
      kps.push_back(kp); owner.push_back(nullptr); descs.push_back(d);
    }
    std::vector<int> perm(kps.size()); for (size_t i = 0; i < perm.size(); i++) perm[i] = (int)i;
    std::shuffle(perm.begin(), perm.end(), rng);
    o.N_ = (int)kps.size(); o.keypoints_.resize(o.N_); o.undistort_keypoints_.resize(o.N_); o.descriptors_ = Mat(o.N_); o.map_points_.assign(o.N_, nullptr);
    o.true_owner_.assign(o.N_, nullptr);
    for (int i = 0; i < o.N_; i++) {
      const int s = perm[i];
      o.keypoints_[i] = kps[s]; o.undistort_keypoints_[i] = kps[s]; std::memcpy(o.descriptors_.ptr(i), descs[s].data(), 32);
      o.true_owner_[i] = owner[s];
      if (owner[s] && self_kf) { o.map_points_[i] = owner[s]; owner[s]->AddObservation((KeyFrame*)self_kf, i); }
      const unsigned node = owner[s] ? (unsigned)(owner[s]->id_ % 61) * 3 + 5 : (unsigned)(rng() % 70) * 3 + 5;
      o.feature_vector_[node].push_back((unsigned)i);
    }
    o.build_grid(0, (float)W, 0, (float)H);
  };
  for (int k = 0; k < n_kf; k++) {
    KeyFrame& kf = S.kfs[k];
    kf.id_ = KeyFrame::next_id_++;
    kf.fx_ = fx; kf.fy_ = fy; kf.cx_ = cx; kf.cy_ = cy; kf.min_x_ = 0; kf.min_y_ = 0; kf.max_x_ = W; kf.max_y_ = H;
    kf.SetPose(make_pose(0.01 * k + 0.004 * G(rng), 0.003 * G(rng), Vector3d(0.05 * G(rng), 0.02 * G(rng), 0.8 * k)));
    kf.n_set_pose_calls_ = 0;
    fill_owner(kf, kf.Tcw_, 0.75, 250, &kf);
    S.map.keyframes_.push_back(&kf);
  }
  // covisibility (KeyFrame::UpdateConnections, src/KeyFrame.cc:293-377): weight = shared map points; the window is kept at
  // |j - k| <= 3 so that local BA sees fixed keyframes in a six-keyframe scene; neighbours ordered by weight, descending
  for (int k = 0; k < n_kf; k++) {
    std::vector<std::pair<int, int> > wj;
    for (int j = 0; j < n_kf; j++) {
      if (j == k || std::abs(j - k) > 3) continue;
      int w = 0;
      for (MapPoint* p : S.kfs[k].map_points_) if (p && p->IsInKeyFrame(&S.kfs[j])) w++;
      wj.push_back(std::make_pair(w, j));
    }
    std::stable_sort(wj.begin(), wj.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first > b.first; });
    for (auto& e : wj) { S.kfs[k].ordered_connected_keyframes_.push_back(&S.kfs[e.second]); S.kfs[k].ordered_weights_.push_back(e.first); S.kfs[k].connected_keyframe_weights_[&S.kfs[e.second]] = e.first; }
    if (k > 0) { S.kfs[k].parent_ = &S.kfs[k - 1]; S.kfs[k - 1].children_.insert(&S.kfs[k]); }      // spanning tree: a chain
  }
  for (MapPoint& mp : S.mps) if (!mp.observations_.empty()) mp.reference_keyframe_ = mp.observations_.begin()->first;
  for (MapPoint& mp : S.mps) if (mp.n_observations_ > 0) S.map.map_points_.push_back(&mp);
  for (int f = 0; f < n_frames; f++) {
    Frame& F = S.frames[f];
    F.Tcw_ = make_pose(0.01 * (n_kf - 1 + f) + 0.004 * G(rng), 0.003 * G(rng), Vector3d(0.05 * G(rng), 0.02 * G(rng), 0.8 * (n_kf - 1) + 0.4 * (f + 1)));
    fill_owner(F, F.Tcw_, 0.85, 300, nullptr);
    F.is_outliers_.assign(F.N_, false);
  }
}

}  // namespace mock
