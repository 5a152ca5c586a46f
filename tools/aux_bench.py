"""Throughput of the widened rows on one GPU next to the CPU oracle: BoW tree descent (k = 10, L = 6,
synthetic tree standing in for ORBvoc.txt), frustum test, grid + window candidates.  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ceres_mono_orb_slam2_amd import synth, frame
from ceres_mono_orb_slam2_amd.vocabulary import ORBVocabulary
from oracle import pyoracle as po

out = {}
voc = synth.make_vocabulary(0, k=10, L=6)
V = ORBVocabulary(*[voc[k] for k in ("node_desc", "child_off", "children", "word_id", "weight", "L")])
rng = np.random.default_rng(0)
leaves = np.nonzero(voc["word_id"] >= 0)[0]
nfr, nd = 256, 2000
d = voc["node_desc"][rng.choice(leaves, nfr * nd)] ^ (rng.integers(0, 256, (nfr * nd, 32), dtype=np.uint8) & rng.integers(0, 256, (nfr * nd, 32), dtype=np.uint8) & 0x21)
dd = torch.from_numpy(d).cuda()
for _ in range(3): V.descend_device(dd, 4)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): V.descend_device(dd, 4)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
out["bow_descend_frames_per_s"] = nfr / dt
out["bow_descend_ms_per_256_frames"] = dt * 1e3
out["bow_descend_hamming_per_s"] = nfr * nd * 60 / dt
t0 = time.perf_counter(); V.transform(d[:nd], 4); V.transform(d[nd:2 * nd], 4); out["bow_transform_single_frame_ms"] = (time.perf_counter() - t0) / 2 * 1e3
t0 = time.perf_counter()
for f in range(4): po.bow_transform(voc, d[f * nd:(f + 1) * nd], 4)
out["cpu_bow_transform_frames_per_s"] = 4 / (time.perf_counter() - t0)
# frustum
n = 200000
q = synth.quat_from_rotvec(rng.normal(0, 0.2, 3)); R = synth.quat_to_R(q); t = rng.normal(0, 1.0, 3)
P = np.stack([rng.normal(0, 15, n), rng.normal(0, 6, n), rng.uniform(-10, 80, n)], 1); Pn = rng.normal(0, 1, (n, 3)); Pn /= np.linalg.norm(Pn, axis=1)[:, None]
mx = rng.uniform(5, 100, n).astype(np.float32); mn = (mx / 3.58).astype(np.float32)
b = np.array([0, 1241, 0, 376], np.float32); K4 = synth.KITTI_K4.astype(np.float32)
a = (R, t, K4, b, P, Pn, mn, mx, 0.5, np.float32(np.log(np.float32(1.2))), 8)
frame.isInFrustum(*a); t0 = time.perf_counter(); frame.isInFrustum(*a); out["frustum_points_per_s_host_api"] = n / (time.perf_counter() - t0)
t0 = time.perf_counter(); po.is_in_frustum(*a); out["cpu_frustum_points_per_s"] = n / (time.perf_counter() - t0)
# grid + window candidates
k = np.stack([rng.uniform(0, 1241, 2000), rng.uniform(0, 376, 2000), rng.integers(0, 8, 2000), rng.uniform(0, 360, 2000)], 1).astype(np.float32)
qq = (k[:, :2] + rng.normal(0, 5, (2000, 2))).astype(np.float32); r = rng.uniform(10, 40, 2000).astype(np.float32)
frame.GetFeaturesInArea(k, b, qq, r); t0 = time.perf_counter(); off, idx = frame.GetFeaturesInArea(k, b, qq, r); out["area_queries_per_s_host_api"] = 2000 / (time.perf_counter() - t0)
out["area_candidates_per_query"] = float(off[-1]) / 2000
m1 = np.full(2000, -1, np.int32); t0 = time.perf_counter(); po.features_in_area(k, b, qq, r, m1, m1); out["cpu_area_queries_per_s"] = 2000 / (time.perf_counter() - t0)
# essential graph (N4): 500 keyframes, ~700 edges
from ceres_mono_orb_slam2_amd import optimizer
from tests.test_oracle_essential_graph import build_problem
g = synth.make_essential_graph(11, n=500, drift=0.001, n_corrected=8)
x0, ej, ei, Sji = build_problem(g)
optimizer.optimize_essential_graph(x0, g["fixed"], ej, ei, Sji)
t0 = time.perf_counter(); x, s = optimizer.optimize_essential_graph(x0, g["fixed"], ej, ei, Sji); dt = time.perf_counter() - t0
out["essential_graph_500kf_ms"] = dt * 1e3; out["essential_graph_500kf_iterations"] = s["iterations"]
t0 = time.perf_counter(); ox, os_ = po.optimize_essential_graph(x0, g["fixed"], ej, ei, Sji); out["cpu_essential_graph_500kf_ms"] = (time.perf_counter() - t0) * 1e3
# triangulation (N4)
from tests.test_oracle_tri import make_tri_problem
p = make_tri_problem(5, n=200000)
a = (p["T1"], p["T2"], p["K1"], p["K2"], p["kp1"], p["kp2"], p["ls"], p["sf"], p["ratio"])
frame.TriangulateMatches(*a); t0 = time.perf_counter(); frame.TriangulateMatches(*a); out["triangulate_matches_per_s_host_api"] = 200000 / (time.perf_counter() - t0)
t0 = time.perf_counter(); po.triangulate_matches(*a); out["cpu_triangulate_matches_per_s"] = 200000 / (time.perf_counter() - t0)
print(json.dumps(out))
