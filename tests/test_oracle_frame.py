"""Pins the oracle's restatement of the Frame-side steps (reference src/Frame.cc:158-173, 191-241, 243-355,
src/MapPoint.cc:406-420) with independent numpy formulations."""
import numpy as np

from oracle import pyoracle as po
from ceres_mono_orb_slam2_amd import synth

BOUNDS = np.array([0.0, 1241.0, 0.0, 376.0], np.float32)
TUM_K4 = np.array([520.908620, 521.007327, 325.141442, 249.701764], np.float32)      # configs/TUM2.yaml:8-17
TUM_DIST = np.array([0.231222, -0.784899, -0.003257, -0.000105, 0.917205], np.float32)


def test_undistort_inverts_the_distortion_model():
    """cv::undistortPoints solves distort(x) = observed by fixed-point iteration: re-distorting its output with the forward
    Brown model must give back the observed pixel (to the 5-iteration residual, < 0.02 px inside the image)."""
    rng = np.random.default_rng(0)
    xy = np.stack([rng.uniform(40, 600, 2000), rng.uniform(40, 440, 2000)], 1).astype(np.float32)
    und = po.undistort_keypoints(xy, TUM_K4, TUM_DIST).astype(np.float64)
    fx, fy, cx, cy = TUM_K4.astype(np.float64); k1, k2, p1, p2, k3 = TUM_DIST.astype(np.float64)
    x = (und[:, 0] - cx) / fx; y = (und[:, 1] - cy) / fy
    r2 = x * x + y * y
    rad = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x); yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    back = np.stack([xd * fx + cx, yd * fy + cy], 1)
    assert np.abs(back - xy).max() < 0.02
    assert np.array_equal(po.undistort_keypoints(xy, TUM_K4, np.zeros(5, np.float32)), xy)       # k1 == 0: copy (src/Frame.cc:330-333)


def test_grid_assignment_against_numpy():
    rng = np.random.default_rng(1)
    n = 3000
    k = np.stack([rng.uniform(-30, 1270, n), rng.uniform(-10, 390, n), rng.integers(0, 8, n), rng.uniform(0, 360, n)], 1).astype(np.float32)
    off, idx = po.assign_features_to_grid(k, BOUNDS)
    winv = np.float32(64) / (BOUNDS[1] - BOUNDS[0]); hinv = np.float32(48) / (BOUNDS[3] - BOUNDS[2])
    fx = (k[:, 0] - BOUNDS[0]) * winv; fy = (k[:, 1] - BOUNDS[2]) * hinv
    px = np.where(fx >= 0, np.floor(fx + np.float32(0.5)), np.ceil(fx - np.float32(0.5))).astype(int)          # round half away from zero
    py = np.where(fy >= 0, np.floor(fy + np.float32(0.5)), np.ceil(fy - np.float32(0.5))).astype(int)
    ok = (px >= 0) & (px < 64) & (py >= 0) & (py < 48)
    cell = px * 48 + py
    want = [np.nonzero(ok & (cell == c))[0] for c in range(64 * 48)]
    assert off[-1] == ok.sum() == len(idx)
    for c in (0, 47, 48, 1500, 3071, int(cell[ok][0])):
        assert np.array_equal(idx[off[c]:off[c + 1]], want[c])                          # ascending keypoint index = push_back order


def test_features_in_area_is_a_superset_filter():
    """every returned index passes |dx| < r, |dy| < r and the level window; every keypoint that passes and lies in a visited
    cell is returned exactly once."""
    rng = np.random.default_rng(2)
    n, nq = 2000, 300
    k = np.stack([rng.uniform(0, 1241, n), rng.uniform(0, 376, n), rng.integers(0, 8, n), rng.uniform(0, 360, n)], 1).astype(np.float32)
    q = np.stack([rng.uniform(0, 1241, nq), rng.uniform(0, 376, nq)], 1).astype(np.float32); r = rng.uniform(5, 60, nq).astype(np.float32)
    lv = rng.integers(1, 8, nq).astype(np.int32)
    off, idx = po.features_in_area(k, BOUNDS, q, r, lv - 1, lv)
    for i in range(nq):
        got = idx[off[i]:off[i + 1]]
        dx = np.abs(k[:, 0] - q[i, 0]); dy = np.abs(k[:, 1] - q[i, 1])
        want = np.nonzero((dx < r[i]) & (dy < r[i]) & (k[:, 2] >= lv[i] - 1) & (k[:, 2] <= lv[i]))[0]
        assert len(set(got)) == len(got) and set(got) <= set(want)
        # the window of cells covers the box except for keypoints rounded into a neighbouring cell at the box edge
        assert len(want) - len(got) <= 2


def test_frustum_against_double_numpy_away_from_boundaries():
    rng = np.random.default_rng(3)
    n = 4000
    q = synth.quat_from_rotvec(rng.normal(0, 0.2, 3)); R = synth.quat_to_R(q); t = rng.normal(0, 1.0, 3)
    P = np.stack([rng.normal(0, 15, n), rng.normal(0, 6, n), rng.uniform(-10, 80, n)], 1)
    Ow = -R.T @ t
    d = np.linalg.norm(P - Ow, axis=1)
    Pn = (P - Ow) / d[:, None] + rng.normal(0, 0.5, (n, 3)); Pn /= np.linalg.norm(Pn, axis=1)[:, None]
    maxd = (d * rng.uniform(0.5, 3.0, n)).astype(np.float32); mind = (maxd / np.float32(1.2 ** 7)).astype(np.float32)
    K4 = synth.KITTI_K4.astype(np.float32)
    iv, uv, lv, vc = po.is_in_frustum(R, t, K4, BOUNDS, P, Pn, mind, maxd, 0.5, np.float32(np.log(np.float32(1.2))), 8)
    Pc = P @ R.T + t
    u = K4[0] * Pc[:, 0] / Pc[:, 2] + K4[2]; v = K4[1] * Pc[:, 1] / Pc[:, 2] + K4[3]
    cosv = np.einsum("ij,ij->i", P - Ow, Pn) / d
    want = (Pc[:, 2] >= 0) & (u >= 0) & (u <= 1241) & (v >= 0) & (v <= 376) & (d >= 0.8 * mind) & (d <= 1.2 * maxd) & (cosv >= 0.5)
    margin = (np.abs(Pc[:, 2]) > 1e-3) & (np.abs(u) > 1e-2) & (np.abs(u - 1241) > 1e-2) & (np.abs(v) > 1e-2) & (np.abs(v - 376) > 1e-2) & \
             (np.abs(d - 0.8 * mind) > 1e-4 * d) & (np.abs(d - 1.2 * maxd) > 1e-4 * d) & (np.abs(cosv - 0.5) > 1e-5)
    assert np.array_equal(iv[margin].astype(bool), want[margin]) and margin.mean() > 0.98
    front = Pc[:, 2] > 0.1
    assert np.abs(uv[front, 0] - u[front]).max() < 1e-2 * max(1.0, np.abs(u[front]).max() / 1e3)
    lvl = np.clip(np.ceil(np.log(maxd.astype(np.float64) / d) / np.log(1.2)), 0, 7).astype(int)
    safe = np.abs(np.log(maxd.astype(np.float64) / d) / np.log(1.2) - np.round(np.log(maxd.astype(np.float64) / d) / np.log(1.2))) > 1e-4
    assert np.array_equal(lv[safe], lvl[safe])
    assert np.abs(vc - cosv).max() < 1e-5
