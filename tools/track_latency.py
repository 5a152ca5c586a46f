"""Per-frame latency of the device-resident Tracking step (orbt_track_with_motion_model) on a 1241 x 376 frame pair.
Under rocprofv3 --kernel-trace --stats the kernel breakdown of the chain is the by-product."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceres_mono_orb_slam2_amd import ORBextractor, tracking, synth
K4 = np.array([718.856, 718.856, 607.1928, 185.2157], np.float32)
B = np.array([0, 1241, 0, 376], np.float32)
seq, offs = synth.make_sequence(11, 1241, 376, 2, "blocks", max_shift=6)
ex = ORBextractor(2000, 1.2, 8, 20, 7)
k0, d0 = ex(seq[0])
n = len(k0); depth = 18.0
X = np.stack([(k0["x"] - K4[2]) / K4[0] * depth, (k0["y"] - K4[3]) / K4[1] * depth, np.full(n, depth)], 1).astype(np.float64)
sh = (offs[1] - offs[0]).astype(np.float64)
T = np.eye(4); T[0, 3] = -sh[0] * depth / K4[0] + 0.01; T[1, 3] = -sh[1] * depth / K4[1] - 0.01
a = (ex, seq[1], K4, B, T, X, d0, k0["octave"].astype(np.int32), k0["angle"].astype(np.float32), np.ones(n, np.uint8), 15.0, True)
for _ in range(10): r = tracking.track_with_motion_model(*a, copy=False)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
t0 = time.perf_counter()
for _ in range(N): r = tracking.track_with_motion_model(*a, copy=False)
ms = (time.perf_counter() - t0) / N * 1e3
t0 = time.perf_counter()
for _ in range(N): ex(seq[1])
ms_ex = (time.perf_counter() - t0) / N * 1e3
# second stage (TrackLocalMap) on the frame the last call left on the device: a local map of the last frame's points that stage 1
# did not match + 1500 further points around the trajectory, th = 1
r = tracking.track_with_motion_model(*a, copy=True)
rng = np.random.default_rng(5)
own = r["owner"].copy(); own[r["outlier"]] = -1
nk = len(r["kps"]); has = own >= 0
slot_state = np.zeros(nk, np.uint8); slot_state[has] = 1
slot_X = np.zeros((nk, 3)); slot_X[has] = X[own[has]]
Xr = np.stack([rng.uniform(-20, 20, 1500), rng.uniform(-6, 6, 1500), rng.uniform(8, 40, 1500)], 1)
MX = np.concatenate([X, Xr]); m = len(MX)
MD = np.concatenate([d0, rng.integers(0, 256, (1500, 32), dtype=np.uint8)])
oc = np.concatenate([k0["octave"].astype(int), rng.integers(0, 8, 1500)])
scale = 1.2 ** np.arange(8)
dist = np.linalg.norm(MX, axis=1)
maxd = (dist * scale[oc]).astype(np.float32); mind = (maxd / scale[7]).astype(np.float32)
Pn = MX / dist[:, None]
state = np.ones(m, np.uint8); state[own[has]] = 0
from oracle import pyoracle
Tm = pyoracle.pose7_to_matrix4d(r["pose7"])
b = (ex, K4, B, Tm, np.float32(np.log(np.float32(1.2))), MX, Pn, mind, maxd, MD, state, slot_X, slot_state, 1.0, 0.8)
for _ in range(10): r2 = tracking.track_local_map(*b)
t0 = time.perf_counter()
for _ in range(N): r2 = tracking.track_local_map(*b)
ms_lm = (time.perf_counter() - t0) / N * 1e3
print(json.dumps({"track_local_map_ms": round(ms_lm, 4), "local_map_points": m, "in_view": int(r2["n_in_view"]), "matched": r2["nmatches"], "correspondences": r2["n_correspondences"],
                  "inliers": r2["n_inliers"], "greedy_rounds": r2["greedy_rounds"]}))
# TrackReferenceKeyFrame on an ORBvoc-shaped tree (k = 10, L = 6: 100 nodes at levelsup 4), the last frame as the reference keyframe
from ceres_mono_orb_slam2_amd.vocabulary import ORBVocabulary
voc = synth.make_vocabulary(1, k=10, L=6)
V = ORBVocabulary(*[voc[x] for x in ("node_desc", "child_off", "children", "word_id", "weight", "L")])
kbw, kbv, kfv = V.transform(d0, 4)
c = (ex, V, seq[1], K4, B, T, d0, np.ones(n, np.uint8), k0["angle"].astype(np.float32), X, kfv, 0.7, True)
for _ in range(10): r3 = tracking.track_reference_keyframe(*c)
t0 = time.perf_counter()
for _ in range(N): r3 = tracking.track_reference_keyframe(*c)
ms_rk = (time.perf_counter() - t0) / N * 1e3
c2 = (ex, V, None) + c[3:]
t0 = time.perf_counter()
for _ in range(N): r4 = tracking.track_reference_keyframe(*c2)
ms_rk2 = (time.perf_counter() - t0) / N * 1e3
print(json.dumps({"track_reference_keyframe_ms": round(ms_rk, 4), "without_extraction_ms": round(ms_rk2, 4), "vocabulary": "k=10 L=6 synthetic", "keyframe_nodes": int(len(kfv[0])),
                  "matches": r3["nmatches"], "inliers": r3["n_inliers"]}))
print(json.dumps({"tracking_step_ms": round(ms, 4), "orbx_extract_alone_ms": round(ms_ex, 4), "keypoints": len(r["kps"]), "matches": r["nmatches"], "inliers": r["n_inliers"],
                  "greedy_rounds": r["greedy_rounds"]}))
