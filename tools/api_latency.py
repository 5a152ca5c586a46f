"""Wall latency of the single-call host-pointer entry points a real-time tracker uses once per frame."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceres_mono_orb_slam2_amd import synth, optimizer, ORBmatcher, ORBextractor
import bench
def lat(f, n=30):
    for _ in range(3): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
out = {}
frames = bench.make_frames(4, seed=0)
ex = ORBextractor(2000, 1.2, 8, 20, 7)
out["orbx_extract_ms"] = lat(lambda: ex(frames[0]))
k1, d1 = ex(frames[0]); k2, d2 = ex(frames[1])
p = synth.make_pose_problem(0, n=2000)
out["ba_pose_optimization_ms"] = lat(lambda: optimizer.pose_optimization(p["K4"], p["pose0"], p["Xw"], p["uv"], p["inv_sigma2"]))
m = ORBmatcher(0.9, True)
out["orbm_hamming_best2_2000x2000_ms"] = lat(lambda: m.hamming_best2(d1, d2))
kps4 = np.stack([k2["x"], k2["y"], k2["octave"].astype(np.float32), k2["angle"]], 1).astype(np.float32)
b = np.array([0, 1241, 0, 376], np.float32)
quv = np.stack([k1["x"], k1["y"]], 1).astype(np.float32) + 2
qr = np.full(len(k1), 15, np.float32)
out["orbm_search_by_projection_ms"] = lat(lambda: m.search_by_projection(kps4, d2, b, quv, qr, d1, q_angle=k1["angle"]))
# ---- LocalMapping steps (round 5): one call for all neighbours against one call per neighbour and stage
from tests.test_gpu_localmapping import make_scene, SF, LS, BOUNDS
from ceres_mono_orb_slam2_amd import localmapping, frame as frame_ops
cur, nbs = make_scene(2, n_nb=20)
ratio = np.float32(1.8)
lm = {}
lm["orbl_create_new_map_points_20_neighbours_ms"] = lat(lambda: localmapping.create_new_map_points(cur, nbs, SF, LS, ratio), 20)
lm["orbl_create_new_map_points_20_neighbours_c_call_only_ms"] = lat(localmapping.prepare_create_new_map_points(cur, nbs, SF, LS, ratio), 30)      # arguments marshalled once: what a C++ caller pays
mm = ORBmatcher(0.6, False)
def per_neighbour():
    mask = cur["unmapped"].copy()
    for q in nbs:
        n_, m12 = mm.SearchForTriangulation(cur["kps"], cur["desc"], mask, q["kps"], q["desc"], q["unmapped"], cur["fv"], q["fv"], q["F12"], q["epipole"], SF, LS)
        hit = np.nonzero(m12 >= 0)[0]
        if len(hit):
            X, ok = frame_ops.TriangulateMatches(cur["Tcw"], q["Tcw"], cur["K4"], q["K4"], cur["kps"][hit][:, :3], q["kps"][m12[hit]][:, :3], LS, SF, ratio)
            mask[hit[ok.astype(bool)]] = 0
lm["per_neighbour_search_for_triangulation_plus_triangulate_20_neighbours_ms"] = lat(per_neighbour, 10)
m_, ok_, X_, npr_ = localmapping.create_new_map_points(cur, nbs, SF, LS, ratio)
lm["new_points"] = int(ok_.sum()); lm["keypoints_current_keyframe"] = int(len(cur["kps"]))
rng = np.random.default_rng(1)
Mq = 1200
kfs = [dict(kps=q["kps"], desc=q["desc"], bounds=BOUNDS) for q in nbs[:12]]
pick = [rng.integers(0, len(q["kps"]), Mq) for q in kfs]
uv = np.array([q["kps"][pk, :2] + rng.normal(0, 1.0, (Mq, 2)).astype(np.float32) for q, pk in zip(kfs, pick)], np.float32)
lvl = np.array([q["kps"][pk, 2].astype(np.int32) for q, pk in zip(kfs, pick)], np.int32)
rad = (3.0 * SF[lvl]).astype(np.float32)
mpd = cur["desc"][rng.choice(len(cur["desc"]), Mq, replace=False)]
ils = (1.0 / LS).astype(np.float32)
lm["orbl_fuse_batch_12_keyframes_x_1200_points_ms"] = lat(lambda: localmapping.fuse_batch(kfs, uv, rad, lvl, mpd, ils), 20)
mf = ORBmatcher(0.6, False)
def per_kf():
    for t, q in enumerate(kfs):
        mf.search_by_projection(q["kps"], q["desc"], q["bounds"], uv[t], rad[t], mpd, q_pred_level=lvl[t], inv_level_sigma2=ils, chi2_gate=5.99, th=50)
lm["per_keyframe_search_by_projection_12_keyframes_ms"] = lat(per_kf, 10)
out["localmapping"] = lm
print(json.dumps(out))
