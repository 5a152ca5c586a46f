"""GPU parity (bit-exact) for the Frame-side steps around extract -> match: grid assignment, window candidates
in the reference's order, frustum test + scale prediction, keypoint undistortion -- HIP C ABI vs the CPU oracle."""
import numpy as np
import pytest

from ceres_mono_orb_slam2_amd import synth

pytestmark = pytest.mark.gpu
KITTI_BOUNDS = np.array([0.0, 1241.0, 0.0, 376.0], np.float32)


def _kps(seed, n, bounds=KITTI_BOUNDS, spill=0.0):
    rng = np.random.default_rng(seed)
    w, h = bounds[1] - bounds[0], bounds[3] - bounds[2]
    x = rng.uniform(bounds[0] - spill * w, bounds[1] + spill * w, n); y = rng.uniform(bounds[2] - spill * h, bounds[3] + spill * h, n)
    # a clump, so that some cells hold many keypoints
    x[: n // 8] = rng.normal(600, 4, n // 8); y[: n // 8] = rng.normal(200, 3, n // 8)
    return np.stack([x, y, rng.integers(0, 8, n), rng.uniform(0, 360, n)], 1).astype(np.float32)


@pytest.mark.parametrize("seed,n,spill", [(0, 2000, 0.0), (1, 4000, 0.05), (2, 37, 0.0), (3, 0, 0.0), (4, 9000, 0.02)])
def test_assign_features_to_grid(oracle, seed, n, spill):
    from ceres_mono_orb_slam2_amd import frame
    k = _kps(seed, n, spill=spill)
    off, idx = frame.AssignFeaturesToGrid(k, KITTI_BOUNDS)
    ooff, oidx = oracle.assign_features_to_grid(k, KITTI_BOUNDS)
    assert np.array_equal(off, ooff) and np.array_equal(idx, oidx)
    if spill:
        assert len(idx) < n                                          # keypoints outside the (undistorted) bounds are skipped


@pytest.mark.parametrize("seed,n,nq", [(0, 2000, 1500), (1, 4000, 3000), (2, 50, 20), (5, 2000, 1)])
def test_features_in_area_order_and_levels(oracle, seed, n, nq):
    from ceres_mono_orb_slam2_amd import frame
    rng = np.random.default_rng(100 + seed)
    k = _kps(seed, n, spill=0.02)
    q = np.stack([rng.uniform(-50, 1300, nq), rng.uniform(-30, 400, nq)], 1).astype(np.float32)
    q[: nq // 4] = k[rng.integers(0, n, nq // 4), :2] + rng.normal(0, 3, (nq // 4, 2)).astype(np.float32)
    r = rng.uniform(2, 90, nq).astype(np.float32)
    lv = rng.integers(0, 8, nq)
    for mn, mx in ((None, None), ((lv - 1).astype(np.int32), lv.astype(np.int32)), (np.full(nq, 2, np.int32), np.full(nq, -1, np.int32))):
        off, idx = frame.GetFeaturesInArea(k, KITTI_BOUNDS, q, r, mn, mx)
        ooff, oidx = oracle.features_in_area(k, KITTI_BOUNDS, q, r, np.full(nq, -1, np.int32) if mn is None else mn,
                                             np.full(nq, -1, np.int32) if mx is None else mx)
        assert np.array_equal(off, ooff) and np.array_equal(idx, oidx)  # same candidates in the same ORDER
    assert off[-1] > 0 or n < 100 or nq < 10


def test_is_in_frustum_and_predict_scale(oracle):
    from ceres_mono_orb_slam2_amd import frame
    rng = np.random.default_rng(7)
    n = 5000
    q = synth.quat_from_rotvec(rng.normal(0, 0.2, 3)); R = synth.quat_to_R(q); t = rng.normal(0, 1.0, 3)
    P = np.stack([rng.normal(0, 15, n), rng.normal(0, 6, n), rng.uniform(-10, 80, n)], 1)       # some behind the camera
    Ow = -R.T @ t
    d = np.linalg.norm(P - Ow, axis=1)
    Pn = (P - Ow) / d[:, None] + rng.normal(0, 0.5, (n, 3)); Pn /= np.linalg.norm(Pn, axis=1)[:, None]
    maxd = (d * rng.uniform(0.5, 3.0, n)).astype(np.float32); mind = (maxd / np.float32(1.2 ** 7)).astype(np.float32)
    K4 = synth.KITTI_K4.astype(np.float32)
    args = (R, t, K4, KITTI_BOUNDS, P, Pn, mind, maxd, 0.5, np.float32(np.log(np.float32(1.2))), 8)
    iv, uv, lv, vc = frame.isInFrustum(*args)
    oiv, ouv, olv, ovc = oracle.is_in_frustum(*args)
    assert np.array_equal(iv, oiv) and np.array_equal(lv, olv)
    assert np.array_equal(uv.view(np.uint32), ouv.view(np.uint32)) and np.array_equal(vc.view(np.uint32), ovc.view(np.uint32))   # bit-exact floats
    assert 0.05 < iv.mean() < 0.9 and len(np.unique(lv)) >= 6


def test_undistort_keypoints_tum(oracle):
    """TUM2 intrinsics / distortion of the reference's own config (configs/TUM2.yaml); k1 == 0 is the identity copy."""
    from ceres_mono_orb_slam2_amd import frame
    rng = np.random.default_rng(9)
    K4 = np.array([520.908620, 521.007327, 325.141442, 249.701764], np.float32)
    dist = np.array([0.231222, -0.784899, -0.003257, -0.000105, 0.917205], np.float32)
    xy = np.stack([rng.uniform(0, 640, 3000), rng.uniform(0, 480, 3000)], 1).astype(np.float32)
    out = frame.UndistortKeyPoints(xy, K4, dist)
    oout = oracle.undistort_keypoints(xy, K4, dist)
    assert np.array_equal(out.view(np.uint32), oout.view(np.uint32))
    assert np.abs(out - xy).max() > 1.0                               # the distortion actually moves border points
    assert np.array_equal(frame.UndistortKeyPoints(xy, K4, np.zeros(5, np.float32)), xy)
    corners = np.array([[0, 0], [640, 0], [0, 480], [640, 480]], np.float32)      # ComputeImageBounds (src/Frame.cc:357-385)
    assert np.array_equal(frame.UndistortKeyPoints(corners, K4, dist), oracle.undistort_keypoints(corners, K4, dist))
