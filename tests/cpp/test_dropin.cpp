// Drop-in test: the reference-signature classes of csrc/compat/orbslam_dropin.h (HIP library underneath), instantiated over
// the mock data model.  Every call below is written exactly as the reference's call site writes it (the file:line is given),
// so "the call sites compile unchanged" is checked by the compiler.  For every case the COMPLETE scene state is written to
// <outdir>/<case>.bin before and after the call together with the call's arguments and outputs (tests/cpp/scene_io.h);
// tests/test_gpu_compat_cpp.py replays each entry point on the "before" state with tests/dropin_checker.py - an independent
// Python restatement of the reference's semantics over the CPU oracle's flat functions - and compares the whole "after" state.
//   g++ -O1 -std=c++17 -I include -I tests/cpp tests/cpp/test_dropin.cpp -o /tmp/test_dropin -L ceres_mono_orb_slam2_amd/lib -lorbslam_hip
//   /tmp/test_dropin <outdir>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>

#include "../../ceres_mono_orb_slam2_amd/csrc/compat/orbslam_dropin.h"
#include "mock_orbslam.h"
#include "scene_io.h"

namespace mock {
unsigned long MapPoint::next_id_ = 0, KeyFrame::next_id_ = 0;
std::mutex MapPoint::global_mutex_;
float Frame::fx_, Frame::fy_, Frame::cx_, Frame::cy_, Frame::min_x_, Frame::max_x_, Frame::min_y_, Frame::max_y_;
}  // namespace mock
using namespace mock;
using sceneio::Writer;

namespace ORB_SLAM2 {      // the names the reference's call sites use
typedef ORBmatcherT<mock::Types> ORBmatcher;
typedef CeresOptimizerT<mock::Types> CeresOptimizer;
typedef FrameOpsT<mock::Types> FrameOps;
}  // namespace ORB_SLAM2
using ORB_SLAM2::CeresOptimizer;
using ORB_SLAM2::FrameOps;
using ORB_SLAM2::ORBmatcher;

static std::string g_out;
static int g_cases = 0;
static const void* g_spare_scene = nullptr;

static Matrix4d compose(const Matrix4d& A, const Matrix4d& B) { Matrix4d C; for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { double s = 0; for (int k = 0; k < 4; k++) s += A(r, k) * B(k, c); C(r, c) = s; } return C; }
static Matrix4d inverse_rt(const Matrix4d& T) { Matrix4d I; for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) I(r, c) = T(c, r); } for (int r = 0; r < 3; r++) { double s = 0; for (int k = 0; k < 3; k++) s += T(k, r) * T(k, 3); I(r, 3) = -s; } return I; }
static void perturb_map(Scene& S, unsigned seed, double rot, double trans, double pt_rel) {
  std::mt19937 rng(seed); std::normal_distribution<double> G(0, 1);
  for (size_t k = 1; k < S.kfs.size(); k++) {
    Matrix4d D = make_pose(rot * G(rng), rot * G(rng), Vector3d(trans * G(rng), trans * G(rng), trans * G(rng)));
    S.kfs[k].SetPose(compose(D, S.kfs[k].Tcw_)); S.kfs[k].n_set_pose_calls_ = 0;
  }
  for (MapPoint& mp : S.mps) { const double f = 1.0 + pt_rel * G(rng); mp.world_pose_ = Vector3d(mp.world_pose_[0] * f, mp.world_pose_[1] * f, mp.world_pose_[2] * f); }
}
// Sophus::Sim3d(Sophus::RxSO3d(s, R), t), operator*, inverse() for the mock Sim3d (its storage is Sophus's)
static Sim3d make_sim3(double s, const Matrix3d& R, const Vector3d& t) {
  double T[16] = {R(0, 0), R(0, 1), R(0, 2), t[0], R(1, 0), R(1, 1), R(1, 2), t[1], R(2, 0), R(2, 1), R(2, 2), t[2], 0, 0, 0, 1}, p7[7];
  ba_matrix4d_to_pose7(T, p7);
  const double n = std::sqrt(p7[3] * p7[3] + p7[4] * p7[4] + p7[5] * p7[5] + p7[6] * p7[6]), f = std::sqrt(s) / n;
  Sim3d S; S.d[0] = p7[3] * f; S.d[1] = p7[4] * f; S.d[2] = p7[5] * f; S.d[3] = p7[6] * f; S.d[4] = t[0]; S.d[5] = t[1]; S.d[6] = t[2];
  return S;
}
static Sim3d operator*(const Sim3d& a, const Sim3d& b) { Sim3d o; ba_sim3_mul(a.d, b.d, o.d); return o; }
static Matrix3d rot_of(const Matrix4d& T) { Matrix3d R; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R(r, c) = T(r, c); return R; }
static Vector3d trans_of(const Matrix4d& T) { return Vector3d(T(0, 3), T(1, 3), T(2, 3)); }
static std::vector<double> sim3_rows(const std::vector<Sim3d>& v) { std::vector<double> o; for (auto& s : v) o.insert(o.end(), s.d, s.d + 7); return o; }

// feature i of `kf` gets a NEW map point (a copy of the one it holds, seen by `kf` and n_extra_obs - 1 other keyframes): the
// situation Fuse resolves with Replace.  The spare MapPoint slots are taken from the end of the scene's vector, detached first.
static void make_duplicate(Scene& S, KeyFrame* kf, int i, int n_obs) {
  static size_t next_spare = 0;
  if (S.kfs.data() != g_spare_scene) { g_spare_scene = S.kfs.data(); next_spare = 0; }
  MapPoint* p = kf->map_points_[i];
  MapPoint& dup = S.mps[S.mps.size() - 1 - next_spare++];
  if (&dup == p) return;
  for (auto& ob : dup.observations_) ob.first->map_points_[ob.second] = nullptr;           // detach the spare from the map
  for (Frame& F : S.frames) for (MapPoint*& q : F.true_owner_) if (q == &dup) q = nullptr;
  const unsigned long id = 100000 + dup.id_;
  dup = *p; dup.id_ = id; dup.observations_.clear(); dup.n_observations_ = 0;
  p->EraseObservation(kf); dup.AddObservation(kf, i); kf->map_points_[i] = &dup;
  for (KeyFrame& o : S.kfs) {                                                             // extra observers: features of other keyframes that hold no point
    if (dup.n_observations_ >= n_obs) break;
    if (&o == kf || p->IsInKeyFrame(&o)) continue;
    for (int j = 0; j < o.N_; j++) if (!o.map_points_[j]) { o.map_points_[j] = &dup; dup.AddObservation(&o, j); break; }
  }
}

// one case: scene from `seed`, `setup` mutates it, the state is dumped, `call` runs the entry point under test (writing its
// arguments and outputs), the state is dumped again
template <class Setup, class Call> static void run_case(const std::string& name, unsigned seed, int n_kf, Setup setup, Call call) {
  std::unique_ptr<Scene> S(new Scene);
  build_scene(*S, seed, n_kf);
  setup(*S);
  Writer W(g_out + "/" + name + ".bin");
  sceneio::dump_scene(W, "before", *S);
  std::vector<MapPoint*> created;
  call(*S, W, created);
  sceneio::dump_scene(W, "after", *S, &created);
  for (MapPoint* p : created) delete p;
  g_cases++;
}

int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: test_dropin <outdir>\n"); return 2; }
  g_out = argv[1];
  setvbuf(stdout, nullptr, _IOLBF, 0);                           // (a crash must not take the log of the cases before it along)
  printf("device count %d\n", orbhip_device_count());
  if (orbhip_device_count() <= 0) { printf("no HIP device: the drop-in shims have no CPU fallback\n"); return 2; }

  // ---- Tracking::SearchLocalPoints (src/Tracking.cc:796-834): current_frame_.isInFrustum(map_point, 0.5) for every local
  //      point, then matcher.SearchByProjection(current_frame_, local_map_points_, th)
  for (int th : {1, 3}) run_case("search_local_points_th" + std::to_string(th), 11, 6, [&](Scene& S) {
    Frame& current_frame_ = S.frames[0];
    for (int i = 0; i < current_frame_.N_; i += 5) current_frame_.map_points_[i] = current_frame_.true_owner_[i];       // matches tracking already has
    S.mps[3].is_bad_ = true;
  }, [&](Scene& S, Writer& W, std::vector<MapPoint*>&) {
    const sceneio::Index I(S);
    Frame& current_frame_ = S.frames[0];
    std::vector<MapPoint*> local_map_points_;
    for (MapPoint& mp : S.mps) if (mp.n_observations_ > 0) local_map_points_.push_back(&mp);
    W.i32("arg.local_map_points", I.mps(local_map_points_)); W.iscalar("arg.th", th);
    // the loop of :813-826 (isBad points are skipped before the frustum test)
    std::vector<MapPoint*> to_project;
    for (MapPoint* map_point : local_map_points_) if (!map_point->isBad()) to_project.push_back(map_point);
    int n_to_match = 0;
    for (bool in : FrameOps::isInFrustum(current_frame_, to_project, 0.5)) n_to_match += in;
    int ret = 0;
    if (n_to_match > 0) {
      ORBmatcher matcher(0.8);
      ret = matcher.SearchByProjection(current_frame_, local_map_points_, th);
    }
    W.iscalar("ret", ret); W.iscalar("out.n_to_match", n_to_match);
    printf("SearchLocalPoints(th=%d): %d in view, %d matches\n", th, n_to_match, ret);
  });

  // ---- Tracking::TrackWithMotionModel: matcher.SearchByProjection(current_frame_, last_frame_, th)  (src/Tracking.cc:632,638)
  for (int th : {15, 30}) run_case("track_motion_model_th" + std::to_string(th), 12, 6, [&](Scene& S) {
    Frame &last_frame_ = S.frames[0], &current_frame_ = S.frames[1];
    for (int i = 0; i < last_frame_.N_; i++) { if (i % 7) last_frame_.map_points_[i] = last_frame_.true_owner_[i]; last_frame_.is_outliers_[i] = (i % 11) == 0; }
    for (int i = 0; i < current_frame_.N_; i += 9) current_frame_.map_points_[i] = current_frame_.true_owner_[i];
  }, [&](Scene& S, Writer& W, std::vector<MapPoint*>&) {
    Frame &last_frame_ = S.frames[0], &current_frame_ = S.frames[1];
    W.iscalar("arg.th", th);
    ORBmatcher matcher(0.9, true);
    const int nmatches = matcher.SearchByProjection(current_frame_, last_frame_, th);
    W.iscalar("ret", nmatches);
    printf("SearchByProjection(cur, last, th=%d): %d matches\n", th, nmatches);
  });

  // ---- Tracking::Relocalization: matcher2.SearchByProjection(current_frame_, candidate_keyframes[i], found, 10, 100)  (src/Tracking.cc:1085)
  run_case("relocalization_projection", 13, 6, [&](Scene& S) {
    Frame& current_frame_ = S.frames[0];
    for (int i = 0; i < current_frame_.N_; i += 6) if (current_frame_.true_owner_[i]) current_frame_.map_points_[i] = current_frame_.true_owner_[i];
  }, [&](Scene& S, Writer& W, std::vector<MapPoint*>&) {
    const sceneio::Index I(S);
    Frame& current_frame_ = S.frames[0]; KeyFrame* keyframe = &S.kfs[4];
    std::set<MapPoint*> found;
    for (int i = 0; i < current_frame_.N_; i++) if (current_frame_.map_points_[i]) found.insert(current_frame_.map_points_[i]);
    W.iscalar("arg.kf", 4); W.i32("arg.found", I.mps(std::vector<MapPoint*>(found.begin(), found.end())));
    ORBmatcher matcher2(0.9, true);
    const int nadditional = matcher2.SearchByProjection(current_frame_, keyframe, found, 10, 100);
    W.iscalar("ret", nadditional);
    printf("SearchByProjection(cur, KF, found, 10, 100): %d matches\n", nadditional);
  });

  // ---- LoopClosing::ComputeSim3: matcher.SearchByProjection(current_keyframe_, Scw_, loop_map_points_, current_matched_map_points_, 10)  (src/LoopClosing.cc:374)
  run_case("loop_projection", 14, 6, [&](Scene&) {}, [&](Scene& S, Writer& W, std::vector<MapPoint*>&) {
    const sceneio::Index I(S);
    KeyFrame* current_keyframe_ = &S.kfs[5];
    Matrix4d Scw_ = current_keyframe_->GetPose();
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) Scw_(r, c) *= 1.07;                        // [s R | s t]
    std::vector<MapPoint*> loop_map_points_;
    for (MapPoint* p : S.kfs[2].map_points_) if (p) loop_map_points_.push_back(p);
    for (MapPoint* p : S.kfs[3].map_points_) if (p && !p->IsInKeyFrame(&S.kfs[2])) loop_map_points_.push_back(p);
    std::vector<MapPoint*> current_matched_map_points_(current_keyframe_->N_, nullptr);
    for (int i = 0; i < current_keyframe_->N_; i += 8) current_matched_map_points_[i] = current_keyframe_->map_points_[i];
    W.iscalar("arg.kf", 5); W.mat4("arg.Scw", Scw_); W.i32("arg.points", I.mps(loop_map_points_)); W.i32("arg.matched", I.mps(current_matched_map_points_));
    ORBmatcher matcher(0.75, true);
    const int ret = matcher.SearchByProjection(current_keyframe_, Scw_, loop_map_points_, current_matched_map_points_, 10);
    W.iscalar("ret", ret); W.i32("out.matched", I.mps(current_matched_map_points_));
    printf("SearchByProjection(KF, Scw, points, matched, 10): %d matches\n", ret);
  });

  // ---- Tracking::TrackReferenceKeyFrame: matcher.SearchByBoW(reference_keyframe_, current_frame_, map_point_matches)  (src/Tracking.cc:576)
  run_case("bow_kf_frame", 15, 6, [&](Scene&) {}, [&](Scene& S, Writer& W, std::vector<MapPoint*>&) {
    const sceneio::Index I(S);
    KeyFrame* reference_keyframe_ = &S.kfs[5]; Frame& current_frame_ = S.frames[0];
    std::vector<MapPoint*> map_point_matches;
    ORBmatcher matcher(0.7, true);
    const int nmatches = matcher.SearchByBoW(reference_keyframe_, current_frame_, map_point_matches);
    W.iscalar("ret", nmatches); W.i32("out.matches", I.mps(map_point_matches));
    printf("SearchByBoW(KF, Frame): %d matches\n", nmatches);
  });

  // ---- LoopClosing::ComputeSim3: matcher.SearchByBoW(current_keyframe_, keyframe, map_point_matches[i])  (src/LoopClosing.cc:262)
  run_case("bow_kf_kf", 16, 6, [&](Scene& S) {
    KeyFrame* keyframe = &S.kfs[1];
    for (int i = 0; i < keyframe->N_; i += 13) if (keyframe->map_points_[i]) keyframe->map_points_[i]->is_bad_ = true;      // bad points on the way
  }, [&](Scene& S, Writer& W, std::vector<MapPoint*>&) {
    const sceneio::Index I(S);
    KeyFrame *current_keyframe_ = &S.kfs[5], *keyframe = &S.kfs[1];
    std::vector<MapPoint*> matches;
    ORBmatcher matcher(0.75, true);
    const int nmatches = matcher.SearchByBoW(current_keyframe_, keyframe, matches);
    W.iscalar("ret", nmatches); W.i32("out.matches", I.mps(matches));
    printf("SearchByBoW(KF, KF): %d matches\n", nmatches);
  });

  // ---- Tracking::MonocularInitialization: matcher.SearchForInitialization(init_frame_, current_frame_, pre_matched_keypoints_, init_matches_, 100)  (src/Tracking.cc:416)
  run_case("initialization", 17, 6, [&](Scene& S) {
    for (Frame* f : {&S.frames[0], &S.frames[1]}) for (int i = 0; i < f->N_; i++) if (i % 3) { f->undistort_keypoints_[i].octave = 0; f->keypoints_[i].octave = 0; }   // (only level-0 features take part, ":383-385")
  }, [&](Scene& S, Writer& W, std::vector<MapPoint*>&) {
    Frame &init_frame_ = S.frames[0], &current_frame_ = S.frames[1];
    std::vector<Point2f> pre_matched_keypoints_(init_frame_.N_);
    for (int i = 0; i < init_frame_.N_; i++) pre_matched_keypoints_[i] = init_frame_.undistort_keypoints_[i].pt;
    std::vector<int> init_matches_;
    ORBmatcher matcher(0.9, true);
    const int nmatches = matcher.SearchForInitialization(init_frame_, current_frame_, pre_matched_keypoints_, init_matches_, 100);
    std::vector<float> pm; for (auto& p : pre_matched_keypoints_) { pm.push_back(p.x); pm.push_back(p.y); }
    W.iscalar("ret", nmatches); W.i32("out.matches", std::vector<int32_t>(init_matches_.begin(), init_matches_.end())); W.f32("out.prev_matched", pm);
    printf("SearchForInitialization: %d matches\n", nmatches);
  });

  // LocalMapping::ComputeF12 (src/LocalMapping.cc:482-503): F12 = K1^-T [t12]x R12 K2^-1
  auto ComputeF12 = [](KeyFrame* kf1, KeyFrame* kf2) {
    const Matrix4d T12 = compose(kf1->GetPose(), inverse_rt(kf2->GetPose()));
    Matrix3d R12 = rot_of(T12), tx, Ki;
    const double t[3] = {T12(0, 3), T12(1, 3), T12(2, 3)};
    tx(0, 0) = 0; tx(0, 1) = -t[2]; tx(0, 2) = t[1]; tx(1, 0) = t[2]; tx(1, 1) = 0; tx(1, 2) = -t[0]; tx(2, 0) = -t[1]; tx(2, 1) = t[0]; tx(2, 2) = 0;
    Ki(0, 0) = 1.0 / kf1->fx_; Ki(1, 1) = 1.0 / kf1->fy_; Ki(0, 2) = -kf1->cx_ / kf1->fx_; Ki(1, 2) = -kf1->cy_ / kf1->fy_;
    auto mul = [](const Matrix3d& X, const Matrix3d& Y) { Matrix3d Z; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double s = 0; for (int k = 0; k < 3; k++) s += X(r, k) * Y(k, c); Z(r, c) = s; } return Z; };
    return mul(mul(mul(Ki.transpose(), tx), R12), Ki);
  };

  // ---- LocalMapping::CreateNewMapPoints: matcher.SearchForTriangulation(current_keyframe_, neighbor_keyframe, F12, matched_indices_, false)  (src/LocalMapping.cc:250)
  run_case("triangulation_search", 18, 6, [&](Scene& S) {
    for (KeyFrame* kf : {&S.kfs[5], &S.kfs[3]}) for (int i = 0; i < kf->N_; i++) if (i % 5 < 2 && kf->map_points_[i]) { kf->map_points_[i]->EraseObservation(kf); kf->map_points_[i] = nullptr; }     // not yet triangulated
  }, [&](Scene& S, Writer& W, std::vector<MapPoint*>&) {
    KeyFrame *current_keyframe_ = &S.kfs[5], *neighbor_keyframe = &S.kfs[3];
    const Matrix3d F12 = ComputeF12(current_keyframe_, neighbor_keyframe);
    std::vector<std::pair<size_t, size_t> > matched_indices_;
    ORBmatcher matcher(0.6, false);
    const int ret = matcher.SearchForTriangulation(current_keyframe_, neighbor_keyframe, F12, matched_indices_, false);
    std::vector<int32_t> mi; for (auto& m : matched_indices_) { mi.push_back((int32_t)m.first); mi.push_back((int32_t)m.second); }
    W.iscalar("arg.kf1", 5); W.iscalar("arg.kf2", 3); W.mat3("arg.F12", F12); W.iscalar("ret", ret); W.i32("out.pairs", mi);
    printf("SearchForTriangulation: %d pairs\n", ret);
  });

  // ---- LocalMapping::CreateNewMapPoints, the whole function (src/LocalMapping.cc:196-396, monocular): neighbours, baseline
  //      test, F12, SearchForTriangulation, the per-match triangulation + gates (FrameOps::TriangulateMatches), new MapPoints
  run_case("create_new_map_points", 26, 6, [&](Scene& S) {
    for (KeyFrame& kf : S.kfs) for (int i = 0; i < kf.N_; i++) if (i % 5 < 2 && kf.map_points_[i]) { kf.map_points_[i]->EraseObservation(&kf); kf.map_points_[i] = nullptr; }
    for (MapPoint& mp : S.mps) mp.reference_keyframe_ = mp.observations_.empty() ? nullptr : mp.observations_.begin()->first;
  }, [&](Scene& S, Writer& W, std::vector<MapPoint*>& created) {
    KeyFrame* current_keyframe_ = &S.kfs[5]; Map* map_ = &S.map;
    W.iscalar("arg.kf", 5);
    const std::vector<KeyFrame*> neighbor_keyframes = current_keyframe_->GetBestCovisibilityKeyFrames(20);
    ORBmatcher matcher(0.6, false);
    const Vector3d Ow1 = current_keyframe_->GetCameraCenter();
    const float ratioFactor = 1.5f * current_keyframe_->scale_factor_;
    int nnew = 0;
    for (size_t i = 0; i < neighbor_keyframes.size(); i++) {
      KeyFrame* neighbor_keyframe = neighbor_keyframes[i];
      const Vector3d Ow2 = neighbor_keyframe->GetCameraCenter();
      const Vector3d vBaseline = Ow2 - Ow1;
      const float baseline = vBaseline.norm();
      const float medianDepthKF2 = neighbor_keyframe->ComputeSceneMedianDepth(2);
      const float ratioBaselineDepth = baseline / medianDepthKF2;
      if (ratioBaselineDepth < 0.01) continue;
      const Matrix3d F12 = ComputeF12(current_keyframe_, neighbor_keyframe);
      std::vector<std::pair<size_t, size_t> > matched_indices_;
      matcher.SearchForTriangulation(current_keyframe_, neighbor_keyframe, F12, matched_indices_, false);
      // "Triangulate each match" (:267-378) in one call; what stays is the map-point construction (:380-395)
      std::vector<Vector3d> x3Ds; std::vector<bool> ok;
      FrameOps::TriangulateMatches(current_keyframe_, neighbor_keyframe, matched_indices_, ratioFactor, &x3Ds, &ok);
      const int nmatches = matched_indices_.size();
      for (int ikp = 0; ikp < nmatches; ikp++) {
        if (!ok[ikp]) continue;
        const int idx1 = matched_indices_[ikp].first, idx2 = matched_indices_[ikp].second;
        MapPoint* map_point = new MapPoint;                                   // MapPoint(x3D, current_keyframe_, map_)
        map_point->id_ = MapPoint::next_id_++; map_point->world_pose_ = x3Ds[ikp]; map_point->reference_keyframe_ = current_keyframe_;
        created.push_back(map_point);
        map_point->AddObservation(current_keyframe_, idx1);
        map_point->AddObservation(neighbor_keyframe, idx2);
        current_keyframe_->AddMapPoint(map_point, idx1);
        neighbor_keyframe->AddMapPoint(map_point, idx2);
        map_point->UpdateNormalAndDepth();
        map_->AddMapPoint(map_point);
        nnew++;
      }
    }
    W.iscalar("ret", nnew);
    printf("CreateNewMapPoints: %d new points from %zu neighbours\n", nnew, neighbor_keyframes.size());
  });

  // ---- the same function with everything between the baseline test and the MapPoint construction in ONE call for all neighbours
  //      (FrameOps::CreateNewMapPoints -> orbl_create_new_map_points, round 5): same scene, same replay on the Python side
  run_case("create_new_map_points_batched", 26, 6, [&](Scene& S) {
    for (KeyFrame& kf : S.kfs) for (int i = 0; i < kf.N_; i++) if (i % 5 < 2 && kf.map_points_[i]) { kf.map_points_[i]->EraseObservation(&kf); kf.map_points_[i] = nullptr; }
    for (MapPoint& mp : S.mps) mp.reference_keyframe_ = mp.observations_.empty() ? nullptr : mp.observations_.begin()->first;
  }, [&](Scene& S, Writer& W, std::vector<MapPoint*>& created) {
    KeyFrame* current_keyframe_ = &S.kfs[5]; Map* map_ = &S.map;
    W.iscalar("arg.kf", 5);
    const std::vector<KeyFrame*> neighbor_keyframes = current_keyframe_->GetBestCovisibilityKeyFrames(20);
    const Vector3d Ow1 = current_keyframe_->GetCameraCenter();
    const float ratioFactor = 1.5f * current_keyframe_->scale_factor_;
    std::vector<KeyFrame*> kept; std::vector<Matrix3d> F12s;
    for (size_t i = 0; i < neighbor_keyframes.size(); i++) {                 // the baseline test and F12 of every neighbour (:229-247)
      KeyFrame* neighbor_keyframe = neighbor_keyframes[i];
      const Vector3d vBaseline = neighbor_keyframe->GetCameraCenter() - Ow1;
      const float baseline = vBaseline.norm();
      const float medianDepthKF2 = neighbor_keyframe->ComputeSceneMedianDepth(2);
      if (baseline / medianDepthKF2 < 0.01) continue;
      kept.push_back(neighbor_keyframe); F12s.push_back(ComputeF12(current_keyframe_, neighbor_keyframe));
    }
    volatile bool abort_flag = false; int processed = 0;
    const auto res = FrameOps::CreateNewMapPoints(current_keyframe_, kept, F12s, ratioFactor, &abort_flag, &processed);
    int nnew = 0;
    for (size_t k = 0; k < res.size(); k++)
      for (const auto& e : res[k]) {                                           // (:380-393)
        MapPoint* map_point = new MapPoint;
        map_point->id_ = MapPoint::next_id_++; map_point->world_pose_ = e.x3D; map_point->reference_keyframe_ = current_keyframe_;
        created.push_back(map_point);
        map_point->AddObservation(current_keyframe_, e.idx1);
        map_point->AddObservation(kept[k], e.idx2);
        current_keyframe_->AddMapPoint(map_point, e.idx1);
        kept[k]->AddMapPoint(map_point, e.idx2);
        map_point->UpdateNormalAndDepth();
        map_->AddMapPoint(map_point);
        nnew++;
      }
    W.iscalar("ret", nnew);
    printf("CreateNewMapPoints (one call for %zu neighbours, %d processed): %d new points\n", kept.size(), processed, nnew);
  });

  // ---- LoopClosing::ComputeSim3: matcher.SearchBySim3(current_keyframe_, keyframe, map_point_matches, s, R, t, 7.5) followed by
  //      CeresOptimizer::OptimizeSim3(current_keyframe_, keyframe, map_point_matches, gScm, 10, is_fix_scale_)  (src/LoopClosing.cc:313-326)
  run_case("sim3_search_and_optimize", 19, 6, [&](Scene&) {}, [&](Scene& S, Writer& W, std::vector<MapPoint*>&) {
    const sceneio::Index I(S);
    KeyFrame *current_keyframe_ = &S.kfs[5], *keyframe = &S.kfs[2];
    const Matrix4d T12 = compose(current_keyframe_->GetPose(), inverse_rt(keyframe->GetPose()));
    const Matrix3d R = rot_of(T12);
    const Vector3d t(T12(0, 3) + 0.01, T12(1, 3), T12(2, 3) - 0.02); const double s = 1.02;
    std::vector<MapPoint*> map_point_matches(current_keyframe_->N_, nullptr);
    for (int i = 0; i < current_keyframe_->N_; i += 10) { MapPoint* p = current_keyframe_->map_points_[i]; if (p && p->IsInKeyFrame(keyframe)) map_point_matches[i] = p; }     // what SearchByBoW + RANSAC left
    W.iscalar("arg.kf1", 5); W.iscalar("arg.kf2", 2); W.scalar("arg.s", s); W.mat3("arg.R", R); W.vec3("arg.t", t); W.i32("arg.matches", I.mps(map_point_matches));
    const bool is_fix_scale_ = false;
    ORBmatcher matcher(0.75, true);
    const int nfound = matcher.SearchBySim3(current_keyframe_, keyframe, map_point_matches, s, R, t, 7.5);
    W.iscalar("out.nfound", nfound); W.i32("out.matches", I.mps(map_point_matches));
    Sim3d gScm = make_sim3(s, R, t);
    W.f64("arg.gScm", std::vector<double>(gScm.d, gScm.d + 7));
    const int n_inliers = CeresOptimizer::OptimizeSim3(current_keyframe_, keyframe, map_point_matches, gScm, 10, is_fix_scale_);
    W.iscalar("ret", n_inliers); W.f64("out.gScm", std::vector<double>(gScm.d, gScm.d + 7));
    printf("SearchBySim3: %d found; OptimizeSim3: %d inliers\n", nfound, n_inliers);
  });

  // ---- LocalMapping::SearchInNeighbors: matcher.Fuse(neighbor_keyframe, map_point_matches)  (src/LocalMapping.cc:441)
  run_case("fuse_kf", 20, 6, [&](Scene& S) {
    KeyFrame *current_keyframe_ = &S.kfs[5], *neighbor_keyframe = &S.kfs[3];
    // make the two keyframes disagree: the neighbour lost some matches and holds DUPLICATE points for others
    for (int i = 0; i < neighbor_keyframe->N_; i++) {
      MapPoint* p = neighbor_keyframe->map_points_[i];
      if (!p) continue;
      if (i % 4 == 0) { neighbor_keyframe->map_points_[i] = nullptr; p->EraseObservation(neighbor_keyframe); }
      else if (i % 4 == 1 && p->IsInKeyFrame(current_keyframe_)) make_duplicate(S, neighbor_keyframe, i, (i % 8 == 1) ? 1 : 4);
    }
  }, [&](Scene& S, Writer& W, std::vector<MapPoint*>&) {
    const sceneio::Index I(S);
    KeyFrame *current_keyframe_ = &S.kfs[5], *neighbor_keyframe = &S.kfs[3];
    std::vector<MapPoint*> map_point_matches = current_keyframe_->GetMapPointMatches();
    W.iscalar("arg.kf", 3); W.i32("arg.points", I.mps(map_point_matches));
    ORBmatcher matcher;
    const int ret = matcher.Fuse(neighbor_keyframe, map_point_matches);
    W.iscalar("ret", ret);
    printf("Fuse(KF, points): %d fused\n", ret);
  });

  // ---- LocalMapping::SearchInNeighbors, first loop: matcher.Fuse(neighbor_keyframe, map_point_matches) for every target keyframe
  //      (src/LocalMapping.cc:437-442) through the batched form (one orbl_fuse_batch call for all targets).  Every neighbour lost some
  //      matches and holds duplicate points: Replace runs, the survivors' descriptors are recomputed, and a later keyframe must be
  //      searched with the NEW descriptor
  run_case("fuse_many", 31, 6, [&](Scene& S) {
    KeyFrame* current_keyframe_ = &S.kfs[5];
    for (int k : {1, 2, 3, 4}) {
      KeyFrame* neighbor_keyframe = &S.kfs[k];
      for (int i = 0; i < neighbor_keyframe->N_; i++) {
        MapPoint* p = neighbor_keyframe->map_points_[i];
        if (!p) continue;
        if ((i + k) % 4 == 0) { neighbor_keyframe->map_points_[i] = nullptr; p->EraseObservation(neighbor_keyframe); }
        else if ((i + k) % 4 == 1 && p->IsInKeyFrame(current_keyframe_)) make_duplicate(S, neighbor_keyframe, i, ((i + k) % 8 == 1) ? 1 : 4);
      }
    }
  }, [&](Scene& S, Writer& W, std::vector<MapPoint*>&) {
    const sceneio::Index I(S);
    KeyFrame* current_keyframe_ = &S.kfs[5];
    std::vector<KeyFrame*> target_keyframes_ = {&S.kfs[3], &S.kfs[1], &S.kfs[4], &S.kfs[2]};
    std::vector<MapPoint*> map_point_matches = current_keyframe_->GetMapPointMatches();
    W.i32("arg.kfs", std::vector<int32_t>{3, 1, 4, 2}); W.i32("arg.points", I.mps(map_point_matches));
    ORBmatcher matcher;
    const std::vector<int> ret = matcher.Fuse(target_keyframes_, map_point_matches);
    W.i32("ret", std::vector<int32_t>(ret.begin(), ret.end()));
    printf("Fuse(%zu KFs, points): %d %d %d %d fused\n", ret.size(), ret[0], ret[1], ret[2], ret[3]);
  });

  // ---- LoopClosing::SearchAndFuse: matcher.Fuse(keyframe, eig_Scw, loop_map_points_, 4, replace_map_points)  (src/LoopClosing.cc:611)
  run_case("fuse_sim3", 21, 6, [&](Scene& S) {
    KeyFrame* keyframe = &S.kfs[4];
    for (int i = 0; i < keyframe->N_; i++) {
      MapPoint* p = keyframe->map_points_[i];
      if (!p) continue;
      if (i % 3 == 0) { keyframe->map_points_[i] = nullptr; p->EraseObservation(keyframe); }
      else if (i % 3 == 1 && p->n_observations_ > 1) make_duplicate(S, keyframe, i, 1);      // the loop side's point meets a duplicate: goes to replace_map_points
    }
  }, [&](Scene& S, Writer& W, std::vector<MapPoint*>&) {
    const sceneio::Index I(S);
    KeyFrame* keyframe = &S.kfs[4];
    Matrix4d eig_Scw = keyframe->GetPose();
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) eig_Scw(r, c) *= 0.96;
    std::vector<MapPoint*> loop_map_points_;
    for (MapPoint& mp : S.mps) if (mp.n_observations_ > 0 && (mp.id_ % 3) != 2) loop_map_points_.push_back(&mp);
    std::vector<MapPoint*> replace_map_points(loop_map_points_.size(), static_cast<MapPoint*>(nullptr));
    W.iscalar("arg.kf", 4); W.mat4("arg.Scw", eig_Scw); W.i32("arg.points", I.mps(loop_map_points_));
    ORBmatcher matcher(0.8);
    const int ret = matcher.Fuse(keyframe, eig_Scw, loop_map_points_, 4, replace_map_points);
    W.iscalar("ret", ret); W.i32("out.replace", I.mps(replace_map_points));
    printf("Fuse(KF, Scw, points, 4, replace): %d fused\n", ret);
  });

  // ---- LoopClosing::SearchAndFuse, the whole loop over the corrected keyframes (src/LoopClosing.cc:599-630) through the batched form (one
  //      orbl_fuse_batch_sim3 call): every keyframe lost some matches and holds duplicates of loop points, so Replace runs between the
  //      keyframes - the loop points take over observations in LATER keyframes (spAlreadyFound must be re-read) and their descriptors
  //      are recomputed (those points are searched again)
  run_case("search_and_fuse", 32, 6, [&](Scene& S) {
    int n_dup = 0;                                                 // (make_duplicate takes its spare points from the end of the scene's vector: a bounded number)
    for (int k : {1, 2, 3, 4}) {
      KeyFrame* keyframe = &S.kfs[k];
      for (int i = 0; i < keyframe->N_; i++) {
        MapPoint* p = keyframe->map_points_[i];
        if (!p) continue;
        if ((i + k) % 3 == 0) { keyframe->map_points_[i] = nullptr; p->EraseObservation(keyframe); }
        else if ((i + k) % 6 == 1 && p->n_observations_ > 1 && n_dup < 80) { make_duplicate(S, keyframe, i, ((i + k) % 12 == 1) ? 1 : 4); n_dup++; }
      }
    }
  }, [&](Scene& S, Writer& W, std::vector<MapPoint*>&) {
    const sceneio::Index I(S);
    Map* map_ = &S.map;
    std::vector<std::pair<KeyFrame*, Matrix4d>> CorrectedPosesMap;
    std::vector<int32_t> order = {2, 4, 1, 3};
    std::vector<double> scws;
    for (int k : order) {
      KeyFrame* keyframe = &S.kfs[k];
      Matrix4d eig_Scw = keyframe->GetPose();
      const double sc = 0.95 + 0.01 * k;
      for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) eig_Scw(r, c) *= sc;
      CorrectedPosesMap.push_back(std::make_pair(keyframe, eig_Scw));
      for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) scws.push_back(eig_Scw(r, c));
    }
    std::vector<MapPoint*> loop_map_points_;
    for (MapPoint& mp : S.mps) if (mp.n_observations_ > 0 && !mp.isBad() && (mp.id_ % 3) != 2) loop_map_points_.push_back(&mp);
    W.i32("arg.kfs", order); W.f64("arg.Scws", scws); W.i32("arg.points", I.mps(loop_map_points_));
    ORBmatcher matcher(0.8);
    const std::vector<int> ret = matcher.SearchAndFuse(CorrectedPosesMap, loop_map_points_, map_);
    W.i32("ret", std::vector<int32_t>(ret.begin(), ret.end()));
    printf("SearchAndFuse(%zu KFs, %zu loop points): %d %d %d %d fused\n", ret.size(), loop_map_points_.size(), ret[0], ret[1], ret[2], ret[3]);
  });

  // ---- Tracking: CeresOptimizer::PoseOptimization(&current_frame_)  (src/Tracking.cc:587,646,684,1074)
  run_case("pose_optimization", 22, 6, [&](Scene& S) {
    Frame& current_frame_ = S.frames[1];
    for (int i = 0; i < current_frame_.N_; i++) {
      current_frame_.map_points_[i] = current_frame_.true_owner_[i];
      if (i % 17 == 0) current_frame_.map_points_[i] = &S.mps[(i * 31) % S.mps.size()];               // wrong associations = outliers
    }
    current_frame_.Tcw_ = compose(make_pose(0.004, -0.003, Vector3d(0.04, -0.02, 0.05)), current_frame_.Tcw_);
  }, [&](Scene& S, Writer& W, std::vector<MapPoint*>&) {
    Frame& current_frame_ = S.frames[1];
    W.iscalar("arg.frame", 1);
    const int ret = CeresOptimizer::PoseOptimization(&current_frame_);
    W.iscalar("ret", ret);
    printf("PoseOptimization: %d inliers\n", ret);
  });
  run_case("pose_optimization_degenerate", 23, 6, [&](Scene& S) {      // < 3 correspondences: returns 0, pose untouched (":330")
    Frame& f = S.frames[0]; f.map_points_[0] = f.true_owner_[0] ? f.true_owner_[0] : &S.mps[0]; f.map_points_[1] = f.true_owner_[1] ? f.true_owner_[1] : &S.mps[1];
  }, [&](Scene& S, Writer& W, std::vector<MapPoint*>&) {
    W.iscalar("arg.frame", 0);
    W.iscalar("ret", CeresOptimizer::PoseOptimization(&S.frames[0]));
  });

  // ---- Tracking::CreateInitialMapMonocular / LoopClosing: CeresOptimizer::GlobalBundleAdjustemnt(map_, 20) ; (map_, 10, &stop, nLoopKF, false)  (src/Tracking.cc:502, src/LoopClosing.cc:656)
  for (unsigned long nLoopKF : {0ul, 5ul}) run_case("global_ba_loopkf" + std::to_string(nLoopKF), 24, 6, [&](Scene& S) {
    perturb_map(S, 99, 0.003, 0.03, 0.01);
    S.kfs[4].is_bad_ = true;                                                        // a culled keyframe must be skipped
    S.mps[7].is_bad_ = true;
  }, [&](Scene& S, Writer& W, std::vector<MapPoint*>&) {
    Map* map_ = &S.map; bool stop = false;
    W.iscalar("arg.n_iterations", 10); W.iscalar("arg.n_loop_kf", (int64_t)nLoopKF); W.iscalar("arg.robust", nLoopKF == 0);
    if (nLoopKF == 0) CeresOptimizer::GlobalBundleAdjustemnt(map_, 10);
    else CeresOptimizer::GlobalBundleAdjustemnt(map_, 10, &stop, nLoopKF, false);
    printf("GlobalBundleAdjustemnt(nLoopKF=%lu)\n", nLoopKF);
  });

  // ---- LocalMapping::Run: CeresOptimizer::LocalBundleAdjustment(current_keyframe_, &is_abort_BA_, map_)  (src/LocalMapping.cc:89)
  for (int preset = 0; preset < 2; preset++) run_case("local_ba_abort" + std::to_string(preset), 25, 6, [&](Scene& S) {
    perturb_map(S, 77, 0.002, 0.02, 0.008);
    for (int i = 0; i < S.kfs[5].N_; i += 23) if (S.kfs[5].map_points_[i]) S.kfs[5].undistort_keypoints_[i].pt.x += 40;      // gross outlier observations: must be erased
  }, [&](Scene& S, Writer& W, std::vector<MapPoint*>&) {
    KeyFrame* current_keyframe_ = &S.kfs[5]; bool is_abort_BA_ = preset != 0; Map* map_ = &S.map;
    W.iscalar("arg.kf", 5); W.iscalar("arg.abort", preset);
    // keyframes 0 and 1 are outside the covisibility window of keyframe 5 -> fixed keyframes
    CeresOptimizer::LocalBundleAdjustment(current_keyframe_, &is_abort_BA_, map_);
    printf("LocalBundleAdjustment(stop preset %d)\n", preset);
  });

  // ---- LoopClosing::CorrectLoop (src/LoopClosing.cc:425-573): corrected / non-corrected Sim3 of the current keyframe's
  //      covisibility group, loop connections, CeresOptimizer::OptimizeEssentialGraph(map_, matched_keyframe_, current_keyframe_,
  //      non_corrected_sim3, corrected_sim3, loop_connections, is_fix_scale_)
  run_case("essential_graph", 27, 14, [&](Scene& S) {
    const int n = (int)S.kfs.size();
    // odometry drift: keyframe k is displaced by k * 3 cm and scaled a little, so that the loop closure has something to distribute
    for (int k = 1; k < n; k++) { Matrix4d T = S.kfs[k].Tcw_; T(0, 3) += 0.03 * k; T(2, 3) -= 0.02 * k; S.kfs[k].SetPose(T); S.kfs[k].n_set_pose_calls_ = 0; }
    // covisibility weights that straddle min_weight = 100 (KeyFrame::GetCovisiblesByWeight returns the prefix >= 100, and NOTHING when all are)
    for (int k = 0; k < n; k++) {
      KeyFrame& kf = S.kfs[k];
      kf.ordered_connected_keyframes_.clear(); kf.ordered_weights_.clear(); kf.connected_keyframe_weights_.clear();
      std::vector<std::pair<int, int> > wj;
      for (int j = 0; j < n; j++) if (j != k && std::abs(j - k) <= 3) wj.push_back(std::make_pair(300 - 70 * std::abs(j - k) - (j % 3), j));
      std::stable_sort(wj.begin(), wj.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first > b.first; });
      for (auto& e : wj) { kf.ordered_connected_keyframes_.push_back(&S.kfs[e.second]); kf.ordered_weights_.push_back(e.first); kf.connected_keyframe_weights_[&S.kfs[e.second]] = e.first; }
    }
    S.kfs[9].loop_edges_.insert(&S.kfs[2]); S.kfs[2].loop_edges_.insert(&S.kfs[9]);          // an older loop closure
    S.kfs[6].is_bad_ = true;                                                            // (culled, still in the containers)
    for (size_t p = 0; p < S.mps.size(); p += 9) if (S.mps[p].reference_keyframe_) { S.mps[p].corrected_by_keyframe_ = S.kfs[n - 1].id_; S.mps[p].corrected_reference_ = S.kfs[n - 2].id_; }
    // weights of the NEW links between the current group and the loop side (KeyFrame::UpdateConnections ran, src/LoopClosing.cc:556), some below min_weight
    KeyFrame *current_keyframe_ = &S.kfs[n - 1], *matched_keyframe_ = &S.kfs[0];
    current_keyframe_->connected_keyframe_weights_[&S.kfs[1]] = 140; current_keyframe_->connected_keyframe_weights_[&S.kfs[2]] = 60;
    current_keyframe_->connected_keyframe_weights_[matched_keyframe_] = 30;               // (the loop edge itself is exempt from the weight test)
    S.kfs[n - 2].connected_keyframe_weights_[matched_keyframe_] = 120; S.kfs[n - 2].connected_keyframe_weights_[&S.kfs[1]] = 99;
  }, [&](Scene& S, Writer& W, std::vector<MapPoint*>&) {
    const sceneio::Index I(S);
    const int n = (int)S.kfs.size();
    Map* map_ = &S.map; KeyFrame *current_keyframe_ = &S.kfs[n - 1], *matched_keyframe_ = &S.kfs[0];
    const bool is_fix_scale_ = false;
    // the Sim3 ComputeSim3 left behind: current camera in the loop side's frame, 4 % scale drift
    const Matrix4d Tcw = current_keyframe_->GetPose();
    Vector3d tc = trans_of(Tcw); tc = Vector3d(tc[0] - 0.03 * (n - 1) * 0.9, tc[1], tc[2] + 0.02 * (n - 1) * 0.9);
    const Sim3d sophus_sim3_Scw_ = make_sim3(1.04, rot_of(Tcw), Vector3d(1.04 * tc[0], 1.04 * tc[1], 1.04 * tc[2]));
    std::vector<KeyFrame*> current_connected_keyframes_ = current_keyframe_->GetVectorCovisibleKeyFrames();
    current_connected_keyframes_.push_back(current_keyframe_);
    Types::KeyFrameAndSim3 sophus_corrected_sim3, sophus_non_corrected_sim3;
    sophus_corrected_sim3[current_keyframe_] = sophus_sim3_Scw_;
    const Matrix4d Twc = inverse_rt(current_keyframe_->GetPose());
    for (KeyFrame* pKFi : current_connected_keyframes_) {                                // (:447-469)
      const Matrix4d Tiw = pKFi->GetPose();
      if (pKFi != current_keyframe_) {
        const Matrix4d Tic = compose(Tiw, Twc);
        const Sim3d sophus_Sic = make_sim3(1.0, rot_of(Tic), trans_of(Tic));
        sophus_corrected_sim3[pKFi] = sophus_Sic * sophus_sim3_Scw_;
      }
      sophus_non_corrected_sim3[pKFi] = make_sim3(1.0, rot_of(Tiw), trans_of(Tiw));
    }
    // loop connections (:548-569): the new links between the current group and the loop side
    std::map<KeyFrame*, std::set<KeyFrame*> > loop_connections;
    loop_connections[current_keyframe_] = {matched_keyframe_, &S.kfs[1], &S.kfs[2]};
    loop_connections[&S.kfs[n - 2]] = {matched_keyframe_, &S.kfs[1]};
    std::vector<int32_t> ck, nk, lc; std::vector<Sim3d> cs, ns;
    for (auto& e : sophus_corrected_sim3) { ck.push_back(I.kf(e.first)); cs.push_back(e.second); }
    for (auto& e : sophus_non_corrected_sim3) { nk.push_back(I.kf(e.first)); ns.push_back(e.second); }
    for (auto& e : loop_connections) for (KeyFrame* j : e.second) { lc.push_back(I.kf(e.first)); lc.push_back(I.kf(j)); }
    W.iscalar("arg.loop_kf", 0); W.iscalar("arg.cur_kf", n - 1);
    W.i32("arg.corrected_kf", ck); W.f64("arg.corrected_sim3", sim3_rows(cs)); W.i32("arg.non_corrected_kf", nk); W.f64("arg.non_corrected_sim3", sim3_rows(ns));
    W.i32("arg.loop_connections", lc);
    CeresOptimizer::OptimizeEssentialGraph(map_, matched_keyframe_, current_keyframe_, sophus_non_corrected_sim3, sophus_corrected_sim3, loop_connections, is_fix_scale_);
    printf("OptimizeEssentialGraph: %d keyframes\n", n);
  });

  // ---- Frame::ComputeBoW / KeyFrame::ComputeBoW (src/Frame.cc:322-327, src/KeyFrame.cc:107-117; Tracking.cc:566, LocalMapping.cc:141)
  run_case("compute_bow", 28, 6, [&](Scene& S) {
    S.frames[0].feature_vector_.clear(); S.kfs[2].feature_vector_.clear();           // (the scene's synthetic feature vectors)
    S.kfs[3].bow_vector_[7] = 0.5;                                                    // already computed: must stay untouched
  }, [&](Scene& S, Writer& W, std::vector<MapPoint*>&) {
    // a synthetic vocabulary (ORBvoc.txt is not shipped): k = 8, L = 5, children = parent with some bits flipped, level by level
    const int k = 8, L = 5;
    std::mt19937 rng(5);
    std::vector<uint8_t> node_desc(32); for (auto& b : node_desc) b = (uint8_t)rng();
    std::vector<uint32_t> child_off{0}, children; std::vector<int32_t> word_id; std::vector<double> weight;
    size_t level_begin = 0, level_end = 1; int next_word = 0;
    for (int lev = 0; lev <= L; lev++) {
      const size_t n_here = level_end - level_begin;
      for (size_t a = 0; a < n_here; a++) {
        const size_t node = level_begin + a;
        if (lev < L) {
          for (int c = 0; c < k; c++) {
            const size_t child = node_desc.size() / 32;
            children.push_back((uint32_t)child);
            node_desc.insert(node_desc.end(), node_desc.begin() + 32 * node, node_desc.begin() + 32 * node + 32);
            flip_bits(&node_desc[32 * child], 40 >> lev, rng);
          }
          word_id.push_back(-1); weight.push_back(0.0);
        } else { word_id.push_back(next_word++); weight.push_back((rng() % 50 == 0) ? 0.0 : 0.5 + (rng() % 1000) / 120.0); }
        child_off.push_back((uint32_t)children.size());
      }
      level_begin = level_end; level_end = node_desc.size() / 32;
    }
    const int n_nodes = (int)word_id.size();
    W.u8("arg.voc_node_desc", node_desc); W.i32("arg.voc_child_off", std::vector<int32_t>(child_off.begin(), child_off.end()));
    W.i32("arg.voc_children", std::vector<int32_t>(children.begin(), children.end())); W.i32("arg.voc_word_id", word_id); W.f64("arg.voc_weight", weight);
    W.iscalar("arg.voc_L", L);
    orbv_ctx* orb_vocabulary_ = nullptr;
    ORB_SLAM2::dropin::check(orbv_create(node_desc.data(), child_off.data(), children.data(), word_id.data(), weight.data(), n_nodes, L, orbhip_get_default_device(), &orb_vocabulary_), "orbv_create");
    FrameOps::ComputeBoW(S.frames[0], orb_vocabulary_);                                // current_frame_.ComputeBoW()
    FrameOps::ComputeBoW(S.kfs[2], orb_vocabulary_, true);                             // current_keyframe_->ComputeBoW()
    FrameOps::ComputeBoW(S.kfs[3], orb_vocabulary_, true);                             // both containers filled already: returns at once
    orbv_destroy(orb_vocabulary_);
    printf("ComputeBoW: %zu words / %zu nodes (frame), %zu / %zu (keyframe)\n", S.frames[0].bow_vector_.size(), S.frames[0].feature_vector_.size(), S.kfs[2].bow_vector_.size(), S.kfs[2].feature_vector_.size());
  });

  // ---- Frame::GetFeaturesInArea / KeyFrame::GetFeaturesInArea as member-function bodies (src/Frame.cc:243-307, src/KeyFrame.cc:575-622)
  run_case("features_in_area", 29, 6, [&](Scene&) {}, [&](Scene& S, Writer& W, std::vector<MapPoint*>&) {
    std::mt19937 rng(3);
    std::vector<float> q; std::vector<int32_t> lv, off{0}, idx;
    for (int k = 0; k < 40; k++) {
      const float x = (float)(rng() % 1400) - 80.f, y = (float)(rng() % 460) - 40.f, r = (k % 4 == 0) ? 250.f : 5.f + (float)(rng() % 60);
      const int lo = (int)(rng() % 5) - 1, hi = lo + (int)(rng() % 3) - 1;
      const std::vector<size_t> a = (k & 1) ? FrameOps::GetFeaturesInArea(S.frames[1], x, y, r, lo, hi) : FrameOps::GetFeaturesInArea(S.kfs[4], x, y, r);
      q.insert(q.end(), {x, y, r}); lv.push_back((k & 1) ? lo : -1); lv.push_back((k & 1) ? hi : -1);
      for (size_t v : a) idx.push_back((int32_t)v);
      off.push_back((int32_t)idx.size());
    }
    W.f32("arg.queries", q); W.i32("arg.levels", lv); W.i32("out.off", off); W.i32("out.idx", idx);
    printf("GetFeaturesInArea: 40 queries, %zu candidates\n", idx.size());
  });

  printf("%d cases written to %s\n", g_cases, g_out.c_str());
  return 0;
}
