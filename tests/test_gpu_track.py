"""GPU parity of the device-resident Tracking step (csrc/orb_track.hip, include/orbslam_hip.h::orbt_track_with_motion_model)
against the CPU oracle's COMPOSITION of the same stages at 1241 x 376 (reference src/Tracking.cc:616-646): the oracle's
extractor, the projection of the last frame's map points restated here with explicit float32 steps, the oracle's
SearchByProjection family (M5: candidate order, `taken` state, points without observations, rotation histogram), the slot
ownership rules of src/ORBmatcher.cc:1232,1260-1264 and the oracle's PoseOptimization on the observations in feature order.
Keypoints, descriptors, matches, owners and outlier flags must be identical, the pose within 1e-7."""
import time

import numpy as np
import pytest

from ceres_mono_orb_slam2_amd import synth

pytestmark = pytest.mark.gpu
F32 = np.float32
K4 = np.array([718.856, 718.856, 607.1928, 185.2157], np.float32)            # configs/KITTI00-02.yaml:8-11
W_IMG, H_IMG = 1241, 376
BOUNDS = np.array([0, W_IMG, 0, H_IMG], np.float32)


def _scenario(oracle, seed, no_obs_frac=0.1, drop_frac=0.1, depth=18.0, pose_noise=2e-3):
    rng = np.random.default_rng(seed)
    seq, offs = synth.make_sequence(seed, W_IMG, H_IMG, 2, "blocks", max_shift=6)
    E = oracle.OracleExtractor(2000)
    k_last, d_last = E.extract(seq[0])
    n = len(k_last)
    shift = (offs[1] - offs[0]).astype(np.float64)                              # frame 1 = frame 0 moved by `shift`: u_cur = u_last - shift
    # the last frame's camera is the world frame; its map points at one depth, so that the image translation is a camera translation
    z = depth * (1.0 + 0.002 * rng.standard_normal(n))
    X = np.stack([(k_last["x"] - K4[2]) / K4[0] * z, (k_last["y"] - K4[3]) / K4[1] * z, z], 1).astype(np.float64)
    t_true = np.array([-shift[0] * depth / K4[0], -shift[1] * depth / K4[1], 0.0])
    rv = pose_noise * rng.standard_normal(3)
    q = synth.quat_from_rotvec(rv)
    pose7 = np.concatenate([t_true + 0.02 * rng.standard_normal(3), q])
    T = oracle.pose7_to_matrix4d(pose7)
    valid = np.ones(n, np.uint8)
    valid[rng.random(n) < no_obs_frac] = 3
    valid[rng.random(n) < drop_frac] = 0
    X[rng.random(n) < 0.03] *= 1.4                                              # wrong associations -> outliers of the pose optimisation
    return dict(img=seq[1], E=E, X=X, desc=d_last, octave=k_last["octave"].astype(np.int32), angle=k_last["angle"].astype(np.float32), valid=valid, T=T)


def _project(T, X, valid, octave, scale, th):
    """The loop head of src/ORBmatcher.cc:1185-1212: double camera coordinates, float u / v, bounds, radius."""
    n = len(X)
    uv = np.zeros((n, 2), np.float32); rad = np.zeros(n, np.float32); v = valid.copy()
    R = [[float(T[r, c]) for c in range(3)] for r in range(3)]; t = [float(T[r, 3]) for r in range(3)]
    fx, fy, cx, cy = [F32(k) for k in K4]
    for i in range(n):
        if not v[i]:
            continue
        P = [float(x) for x in X[i]]
        c = [(R[r][0] * P[0] + R[r][1] * P[1] + R[r][2] * P[2]) + t[r] for r in range(3)]
        xc, yc = F32(c[0]), F32(c[1])
        with np.errstate(divide="ignore"):
            invz = F32(np.float64(1.0) / np.float64(c[2]))
        if invz < 0:
            v[i] = 0; continue
        u = F32(F32(fx * xc) * invz) + cx
        w = F32(F32(fy * yc) * invz) + cy
        if u < BOUNDS[0] or u > BOUNDS[1] or w < BOUNDS[2] or w > BOUNDS[3]:
            v[i] = 0; continue
        uv[i] = (u, w); rad[i] = F32(th) * scale[octave[i]]
    return uv, rad, v


def _expected(oracle, S, th):
    E = S["E"]
    kps, desc = E.extract(S["img"])
    kps4 = np.stack([kps["x"], kps["y"], kps["octave"].astype(np.float32), kps["angle"]], 1).astype(np.float32)
    uv, rad, v = _project(S["T"], S["X"], S["valid"], S["octave"], E.scale, th)
    nm, m, _, _ = oracle.search_by_projection(kps4, desc, BOUNDS, uv, rad, S["desc"], q_min_level=S["octave"] - 1, q_max_level=S["octave"] + 1, q_valid=v,
                                              taken=np.zeros(len(kps4), np.uint8), q_angle=S["angle"], ratio=0.9, th=100, check_ori=True)
    owner = np.full(len(kps4), -1, np.int32)
    for q in range(len(m)):                                                     # assignments in query order (:1232) ...
        if m[q] >= 0: owner[m[q]] = q
        elif m[q] <= -2: owner[-2 - m[q]] = q
    for q in range(len(m)):                                                     # ... then the removed rotation bins empty their slots (:1260-1264)
        if m[q] <= -2: owner[-2 - m[q]] = -1
    feat = np.nonzero(owner >= 0)[0]
    pose0 = oracle.matrix4d_to_pose7(S["T"])
    outl = np.zeros(len(kps4), bool)
    if len(feat) >= 3:
        ninl, pose, out, _ = oracle.pose_optimization(K4.astype(np.float64), pose0, S["X"][owner[feat]], kps4[feat, :2].astype(np.float64), E.inv_sigma2[kps4[feat, 2].astype(int)])
        outl[feat] = out.astype(bool)
    else:
        ninl, pose = 0, pose0
    return dict(kps=kps, desc=desc, match=m, nmatches=nm, owner=owner, outlier=outl, pose7=pose, n_inliers=int(ninl), ncorr=len(feat))


@pytest.mark.parametrize("seed,th,kw", [(3, 15.0, {}), (4, 30.0, {}), (5, 15.0, dict(no_obs_frac=0.5)), (6, 7.0, dict(drop_frac=0.6)), (7, 15.0, dict(pose_noise=0.02))])
def test_tracking_step_vs_oracle_composition(oracle, seed, th, kw):
    from ceres_mono_orb_slam2_amd import ORBextractor, tracking
    S = _scenario(oracle, seed, **kw)
    ex = ORBextractor(2000, 1.2, 8, 20, 7)
    got = tracking.track_with_motion_model(ex, S["img"], K4, BOUNDS, S["T"], S["X"], S["desc"], S["octave"], S["angle"], S["valid"], th, True)
    exp = _expected(oracle, S, th)
    assert np.array_equal(got["kps"], exp["kps"]) and np.array_equal(got["desc"], exp["desc"])
    assert got["nmatches"] == exp["nmatches"] and np.array_equal(got["match"], exp["match"])
    assert np.array_equal(got["owner"], exp["owner"]) and got["n_correspondences"] == exp["ncorr"]
    assert got["n_inliers"] == exp["n_inliers"] and np.array_equal(got["outlier"], exp["outlier"])
    assert np.abs(got["pose7"] - exp["pose7"]).max() < 1e-7
    print("seed %d th %.0f: %d keypoints, %d matches (%d removed by rotation), %d inliers, %d greedy rounds" %
          (seed, th, len(got["kps"]), got["nmatches"], int((got["match"] <= -2).sum()), got["n_inliers"], got["greedy_rounds"]))
    if not kw.get("pose_noise"):
        assert got["nmatches"] > 200 and got["n_inliers"] > 150


def test_tracking_step_periodic_texture_many_lookalikes(oracle):
    """A frame tiled with ONE 20 x 20 patch: every corner has dozens of bit-identical look-alikes inside a th = 40 window, so
    queries hold more than 16 acceptable candidates (the full-list path of k_trk_windows / k_trk_greedy, the first-minimum rule
    among equal distances and long chains of contested targets).  Same frame as last and current one, identity motion."""
    from ceres_mono_orb_slam2_amd import ORBextractor, tracking
    rng = np.random.default_rng(42)
    patch = (rng.random((20, 20)) < 0.5).astype(np.uint8) * 170 + 40
    patch = np.kron(patch[::4, ::4], np.ones((4, 4), np.uint8))                 # 4-pixel blocks: strong corners
    img = np.tile(patch, (H_IMG // 20 + 1, W_IMG // 20 + 1))[:H_IMG, :W_IMG].copy()
    img[::37, ::41] ^= 0x55                                                     # a few irregularities so that not everything ties
    E = oracle.OracleExtractor(2000)
    k_last, d_last = E.extract(img)
    n = len(k_last)
    z = np.full(n, 18.0)
    X = np.stack([(k_last["x"] - K4[2]) / K4[0] * z, (k_last["y"] - K4[3]) / K4[1] * z, z], 1).astype(np.float64)
    valid = np.ones(n, np.uint8); valid[rng.random(n) < 0.2] = 3
    S = dict(img=img, E=E, X=X, desc=d_last, octave=k_last["octave"].astype(np.int32), angle=k_last["angle"].astype(np.float32), valid=valid, T=np.eye(4))
    ex = ORBextractor(2000, 1.2, 8, 20, 7)
    for th in (40.0, 12.0):
        got = tracking.track_with_motion_model(ex, img, K4, BOUNDS, S["T"], X, d_last, S["octave"], S["angle"], valid, th, True)
        exp = _expected(oracle, S, th)
        assert np.array_equal(got["kps"], exp["kps"])
        assert got["nmatches"] == exp["nmatches"] and np.array_equal(got["match"], exp["match"])
        assert np.array_equal(got["owner"], exp["owner"]) and got["n_correspondences"] == exp["ncorr"]
        assert got["n_inliers"] == exp["n_inliers"] and np.array_equal(got["outlier"], exp["outlier"])
        print("periodic texture th %.0f: %d keypoints, %d matches, %d greedy rounds" % (th, n, got["nmatches"], got["greedy_rounds"]))
    assert n > 500


def test_tracking_step_degenerate_inputs(oracle):
    """No last-frame features / fewer than 3 correspondences: the frame is still extracted, the pose stays the predicted one."""
    from ceres_mono_orb_slam2_amd import ORBextractor, tracking
    S = _scenario(oracle, 9)
    ex = ORBextractor(2000, 1.2, 8, 20, 7)
    kps, desc = S["E"].extract(S["img"])
    z = np.zeros
    got = tracking.track_with_motion_model(ex, S["img"], K4, BOUNDS, S["T"], z((0, 3)), z((0, 32), np.uint8), z(0, np.int32), z(0, np.float32), z(0, np.uint8))
    assert np.array_equal(got["kps"], kps) and got["nmatches"] == 0 and got["n_inliers"] == 0 and (got["owner"] == -1).all()
    assert np.abs(got["pose7"] - oracle.matrix4d_to_pose7(S["T"])).max() == 0
    v = np.zeros(len(S["X"]), np.uint8); v[:2] = 1
    got = tracking.track_with_motion_model(ex, S["img"], K4, BOUNDS, S["T"], S["X"], S["desc"], S["octave"], S["angle"], v)
    exp = _expected(oracle, dict(S, valid=v), 15.0)
    assert np.array_equal(got["match"], exp["match"]) and got["n_inliers"] == 0
    assert np.abs(got["pose7"] - oracle.matrix4d_to_pose7(S["T"])).max() == 0


def test_tracking_binding_buffer_semantics(oracle):
    """copy=True (the default) returns private arrays; copy=False returns READ-ONLY views of two alternating buffer sets kept with
    the extractor: a result stays intact through the next call and is overwritten by the one after."""
    from ceres_mono_orb_slam2_amd import ORBextractor, tracking
    S = _scenario(oracle, 12)
    S2 = _scenario(oracle, 13)
    ex = ORBextractor(2000, 1.2, 8, 20, 7)
    a1 = (ex, S["img"], K4, BOUNDS, S["T"], S["X"], S["desc"], S["octave"], S["angle"], S["valid"], 15.0, True)
    a2 = (ex, S2["img"], K4, BOUNDS, S2["T"], S2["X"], S2["desc"], S2["octave"], S2["angle"], S2["valid"], 15.0, True)
    V = dict(copy=False)
    tracking.track_with_motion_model(*a1, **V); tracking.track_with_motion_model(*a2, **V)          # (sizes the buffer sets for both scenarios)
    r1 = tracking.track_with_motion_model(*a1, **V)
    k1, m1 = r1["kps"].copy(), r1["match"].copy()
    assert not r1["kps"].flags.writeable and not r1["match"].flags.writeable
    r2 = tracking.track_with_motion_model(*a2, **V)
    assert np.array_equal(r1["kps"], k1) and np.array_equal(r1["match"], m1)      # still valid after ONE more call
    assert not np.shares_memory(r1["kps"], r2["kps"])
    r3 = tracking.track_with_motion_model(*a2, **V)                              # reuses r1's buffers
    assert np.shares_memory(r1["kps"], r3["kps"]) and np.array_equal(r3["kps"], r2["kps"])
    c = tracking.track_with_motion_model(*a1)                                    # the default: private, writeable copies
    assert not np.shares_memory(c["kps"], r2["kps"]) and not np.shares_memory(c["kps"], r3["kps"])
    assert c["kps"].flags.writeable and np.array_equal(c["kps"], k1) and np.array_equal(c["match"], m1)
    d = tracking.track_with_motion_model(*a2)
    assert np.array_equal(c["kps"], k1)                                          # untouched by later calls


def test_tracking_step_latency(oracle):
    """Per-frame latency of the chained call against the three separate host-pointer calls it replaces (printed; the bar is only
    that the chain is not slower)."""
    from ceres_mono_orb_slam2_amd import ORBextractor, ORBmatcher, optimizer, tracking
    S = _scenario(oracle, 11)
    ex = ORBextractor(2000, 1.2, 8, 20, 7)
    a = (ex, S["img"], K4, BOUNDS, S["T"], S["X"], S["desc"], S["octave"], S["angle"], S["valid"], 15.0, True)
    for _ in range(5): got = tracking.track_with_motion_model(*a, copy=False)
    t0 = time.perf_counter()
    for _ in range(50): got = tracking.track_with_motion_model(*a, copy=False)
    chained = (time.perf_counter() - t0) / 50 * 1e3
    # the separate calls: extract, SearchByProjection (projection on the host as the drop-in class does), PoseOptimization
    E = S["E"]
    def separate():
        kps, desc = ex(S["img"])
        kps4 = np.stack([kps["x"], kps["y"], kps["octave"].astype(np.float32), kps["angle"]], 1).astype(np.float32)
        uv, rad, v = UV
        nm, m, _, tk = ORBmatcher(0.9, True).search_by_projection(kps4, desc, BOUNDS, uv, rad, S["desc"], q_min_level=S["octave"] - 1, q_max_level=S["octave"] + 1,
                                                                 q_valid=v, q_angle=S["angle"], taken=np.zeros(len(kps4), np.uint8), th=100)
        f = np.nonzero(m >= 0)[0]
        return optimizer.pose_optimization(K4.astype(np.float64), oracle.matrix4d_to_pose7(S["T"]), S["X"][f], kps4[m[f], :2].astype(np.float64), E.inv_sigma2[kps4[m[f], 2].astype(int)])
    UV = _project(S["T"], S["X"], S["valid"], S["octave"], E.scale, 15.0)
    try:
        for _ in range(3): separate()
        t0 = time.perf_counter()
        for _ in range(20): separate()
        sep = (time.perf_counter() - t0) / 20 * 1e3
    except Exception as e:                                                      # (the comparison leg must not fail the parity suite)
        sep = float("nan"); print("separate-call leg failed:", repr(e))
    print("tracking step: chained %.3f ms per frame, separate host-pointer calls %.3f ms (numpy glue included)" % (chained, sep))
    assert chained < 1.0
