// Scene serialisation for the drop-in tests (TEST INFRASTRUCTURE).  A dump is a flat sequence of named arrays
//   u32 name_len | name | u8 dtype ('b' u8, 'i' i32, 'q' i64, 'f' f32, 'd' f64) | u64 count | data
// read back by tests/dropin_checker.py (load_records).  dump_scene() writes the COMPLETE state of a mock::Scene - every field
// the hot path reads or writes - so that the Python checker can replay an entry point from the "before" state and compare the
// whole "after" state, not a hand-picked subset.
#pragma once
#include <cstdio>
#include <string>
#include <vector>

#include "mock_orbslam.h"

namespace sceneio {
using namespace mock;

struct Writer {
  FILE* f = nullptr;
  explicit Writer(const std::string& path) { f = std::fopen(path.c_str(), "wb"); if (!f) { std::perror(path.c_str()); std::exit(3); } }
  ~Writer() { if (f) std::fclose(f); }
  void raw(const std::string& name, char code, const void* data, size_t count, size_t elsize) {
    const uint32_t nl = (uint32_t)name.size(); const uint64_t n = count;
    std::fwrite(&nl, 4, 1, f); std::fwrite(name.data(), 1, nl, f); std::fwrite(&code, 1, 1, f); std::fwrite(&n, 8, 1, f);
    if (count) std::fwrite(data, elsize, count, f);
  }
  void u8(const std::string& n, const std::vector<uint8_t>& v) { raw(n, 'b', v.data(), v.size(), 1); }
  void i32(const std::string& n, const std::vector<int32_t>& v) { raw(n, 'i', v.data(), v.size(), 4); }
  void i64(const std::string& n, const std::vector<int64_t>& v) { raw(n, 'q', v.data(), v.size(), 8); }
  void f32(const std::string& n, const std::vector<float>& v) { raw(n, 'f', v.data(), v.size(), 4); }
  void f64(const std::string& n, const std::vector<double>& v) { raw(n, 'd', v.data(), v.size(), 8); }
  void scalar(const std::string& n, double v) { raw(n, 'd', &v, 1, 8); }
  void iscalar(const std::string& n, int64_t v) { raw(n, 'q', &v, 1, 8); }
  void mat4(const std::string& n, const Matrix4d& T) { std::vector<double> v; for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) v.push_back(T(r, c)); f64(n, v); }
  void mat3(const std::string& n, const Matrix3d& T) { std::vector<double> v; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) v.push_back(T(r, c)); f64(n, v); }
  void vec3(const std::string& n, const Vector3d& p) { f64(n, {p[0], p[1], p[2]}); }
};

struct Index {                     // pointer -> index inside the scene's own vectors (-1 for nullptr); points created by the call under test
  const Scene& S; const std::vector<MapPoint*>* extra;                        // (CreateNewMapPoints) are numbered behind the scene's
  Index(const Scene& s, const std::vector<MapPoint*>* e = nullptr) : S(s), extra(e) {}
  int mp(const MapPoint* p) const {
    if (!p) return -1;
    if (p >= S.mps.data() && p < S.mps.data() + S.mps.size()) return (int)(p - S.mps.data());
    if (extra) for (size_t k = 0; k < extra->size(); k++) if ((*extra)[k] == p) return (int)(S.mps.size() + k);
    return -2;
  }
  int kf(const KeyFrame* k) const { return k ? (int)(k - S.kfs.data()) : -1; }
  std::vector<int32_t> mps(const std::vector<MapPoint*>& v) const { std::vector<int32_t> o; for (auto p : v) o.push_back(mp(p)); return o; }
  std::vector<int32_t> kfs(const std::vector<KeyFrame*>& v) const { std::vector<int32_t> o; for (auto p : v) o.push_back(kf(p)); return o; }
};

inline void dump_owner(Writer& W, const std::string& p, const GridOwner& o, const Index& I) {
  std::vector<float> ku, kr; std::vector<int32_t> mp, own;
  for (int i = 0; i < o.N_; i++) {
    const KeyPoint &a = o.undistort_keypoints_[i], &b = o.keypoints_[i];
    ku.insert(ku.end(), {a.pt.x, a.pt.y, (float)a.octave, a.angle}); kr.insert(kr.end(), {b.pt.x, b.pt.y, (float)b.octave, b.angle});
  }
  W.f32(p + ".kpu", ku); W.f32(p + ".kp", kr); W.u8(p + ".desc", o.descriptors_.d);
  W.i32(p + ".mp", I.mps(o.map_points_)); W.i32(p + ".owner", I.mps(o.true_owner_));
  std::vector<int32_t> node, off{0}, idx;
  for (auto& e : o.feature_vector_) { node.push_back((int32_t)e.first); for (auto v : e.second) idx.push_back((int32_t)v); off.push_back((int32_t)idx.size()); }
  W.i32(p + ".fv_node", node); W.i32(p + ".fv_off", off); W.i32(p + ".fv_idx", idx);
  std::vector<int32_t> bw; std::vector<double> bv;
  for (auto& e : o.bow_vector_) { bw.push_back((int32_t)e.first); bv.push_back(e.second); }
  W.i32(p + ".bow_word", bw); W.f64(p + ".bow_value", bv);
}

inline void dump_mappoint_rows(Writer& W, const std::string& p, const std::vector<const MapPoint*>& pts, const Index& I) {
  std::vector<double> pos, nrm, gpos; std::vector<uint8_t> desc; std::vector<int64_t> meta; std::vector<float> fl; std::vector<int32_t> ooff{0}, okf, oidx;
  for (const MapPoint* mp : pts) {
    for (int k = 0; k < 3; k++) { pos.push_back(mp->world_pose_[k]); nrm.push_back(mp->normal_vector_[k]); gpos.push_back(mp->global_BA_pose_[k]); }
    desc.insert(desc.end(), mp->descriptor_.d.begin(), mp->descriptor_.d.begin() + 32);
    meta.insert(meta.end(), {(int64_t)mp->id_, (int64_t)mp->is_bad_, (int64_t)I.mp(mp->replaced_map_point_), (int64_t)mp->n_observations_, (int64_t)mp->n_BA_local_for_keyframe_,
                             (int64_t)mp->n_BA_global_for_keyframe_, (int64_t)mp->n_update_normal_calls_, (int64_t)mp->is_track_in_view_, (int64_t)mp->track_scale_level_,
                             (int64_t)mp->corrected_by_keyframe_, (int64_t)mp->corrected_reference_, (int64_t)I.kf(mp->reference_keyframe_)});
    fl.insert(fl.end(), {mp->min_distance_, mp->max_distance_, mp->track_proj_x_, mp->track_proj_y_, mp->track_view_cos_});
    for (auto& ob : mp->observations_) { okf.push_back(I.kf(ob.first)); oidx.push_back((int32_t)ob.second); }
    ooff.push_back((int32_t)okf.size());
  }
  W.f64(p + ".pos", pos); W.f64(p + ".normal", nrm); W.f64(p + ".gba_pos", gpos); W.u8(p + ".desc", desc); W.i64(p + ".meta", meta); W.f32(p + ".fl", fl);
  W.i32(p + ".obs_off", ooff); W.i32(p + ".obs_kf", okf); W.i32(p + ".obs_idx", oidx);
}

// the whole scene under the prefix `p` ("before" / "after"); `extra` = map points the call under test allocated itself
inline void dump_scene(Writer& W, const std::string& p, const Scene& S, const std::vector<MapPoint*>* extra = nullptr) {
  const Index I(S, extra);
  W.iscalar(p + ".n_kf", (int64_t)S.kfs.size()); W.iscalar(p + ".n_frames", (int64_t)S.frames.size());
  W.f32(p + ".scale", S.scale); W.f32(p + ".sigma2", S.sigma2); W.f32(p + ".inv_sigma2", S.inv_sigma2);
  W.f32(p + ".frame_K", {Frame::fx_, Frame::fy_, Frame::cx_, Frame::cy_}); W.f32(p + ".frame_bounds", {Frame::min_x_, Frame::max_x_, Frame::min_y_, Frame::max_y_});
  for (size_t k = 0; k < S.kfs.size(); k++) {
    const KeyFrame& kf = S.kfs[k]; const std::string q = p + ".kf" + std::to_string(k);
    dump_owner(W, q, kf, I);
    W.mat4(q + ".T", kf.Tcw_); W.mat4(q + ".gba_T", kf.global_BA_Tcw_); W.vec3(q + ".Ow", kf.Ow_);
    W.f32(q + ".K", {kf.fx_, kf.fy_, kf.cx_, kf.cy_}); W.f32(q + ".bounds", {(float)kf.min_x_, (float)kf.max_x_, (float)kf.min_y_, (float)kf.max_y_});
    W.i64(q + ".meta", {(int64_t)kf.id_, (int64_t)kf.is_bad_, (int64_t)kf.n_BA_local_for_keyframe_, (int64_t)kf.n_BA_fixed_for_keyframe_, (int64_t)kf.n_BA_global_for_keyframe_,
                        (int64_t)kf.n_set_pose_calls_, (int64_t)I.kf(kf.parent_)});
    W.i32(q + ".conn", I.kfs(kf.ordered_connected_keyframes_)); W.i32(q + ".conn_w", std::vector<int32_t>(kf.ordered_weights_.begin(), kf.ordered_weights_.end()));
    std::vector<int32_t> ch, le, wk, wv;
    for (KeyFrame* c : kf.children_) ch.push_back(I.kf(c));
    for (KeyFrame* c : kf.loop_edges_) le.push_back(I.kf(c));
    for (auto& e : kf.connected_keyframe_weights_) { wk.push_back(I.kf(e.first)); wv.push_back(e.second); }
    W.i32(q + ".children", ch); W.i32(q + ".loop_edges", le); W.i32(q + ".w_kf", wk); W.i32(q + ".w_val", wv);
  }
  for (size_t k = 0; k < S.frames.size(); k++) {
    const Frame& F = S.frames[k]; const std::string q = p + ".fr" + std::to_string(k);
    dump_owner(W, q, F, I);
    W.mat4(q + ".T", F.Tcw_);
    std::vector<uint8_t> o(F.is_outliers_.size()); for (size_t i = 0; i < o.size(); i++) o[i] = F.is_outliers_[i];
    W.u8(q + ".outl", o); W.iscalar(q + ".n_set_pose", F.n_set_pose_calls_);
  }
  std::vector<const MapPoint*> pts; for (const MapPoint& mp : S.mps) pts.push_back(&mp);
  if (extra) for (const MapPoint* mp : *extra) pts.push_back(mp);
  dump_mappoint_rows(W, p + ".mp", pts, I);
  W.i32(p + ".map_kfs", I.kfs(S.map.keyframes_)); W.i32(p + ".map_mps", I.mps(S.map.map_points_));
}

}  // namespace sceneio
