"""Analytic cross-checks of the BA oracle (Ceres is absent: SURVEY.md 8(c))."""
import numpy as np
import pytest
from scipy.optimize import least_squares

from ceres_mono_orb_slam2_amd import synth


def _resid(oracle, K4, pose, X, uv, w):
    r, _, _, _ = oracle.ba_eval_obs(K4, pose, X, uv, w)
    return r


def test_jacobian_matches_central_differences(oracle):
    rng = np.random.default_rng(0)
    K4 = synth.KITTI_K4
    for _ in range(20):
        q = synth.quat_from_rotvec(rng.normal(0, 0.5, 3)); t = rng.normal(0, 1, 3)
        pose = np.concatenate([t, q])
        X = np.array([rng.normal(0, 3), rng.normal(0, 2), rng.uniform(5, 40)])
        X = synth.quat_to_R(q).T @ (X - t)
        uv = rng.uniform(0, 500, 2); w = rng.uniform(0.1, 1.0)
        r, Jc, Jp, rho = oracle.ba_eval_obs(K4, pose, X, uv, w)
        h = 1e-6
        num = np.zeros((2, 6)); nump = np.zeros((2, 3))
        for k in range(3):
            e = np.zeros(3); e[k] = h
            pp = pose.copy(); pp[:3] += e; pm = pose.copy(); pm[:3] -= e
            num[:, k] = (_resid(oracle, K4, pp, X, uv, w) - _resid(oracle, K4, pm, X, uv, w)) / (2 * h)
            pp = pose.copy(); pp[3:] = oracle.quat_plus(q, e); pm = pose.copy(); pm[3:] = oracle.quat_plus(q, -e)
            num[:, 3 + k] = (_resid(oracle, K4, pp, X, uv, w) - _resid(oracle, K4, pm, X, uv, w)) / (2 * h)
            nump[:, k] = (_resid(oracle, K4, pose, X + e, uv, w) - _resid(oracle, K4, pose, X - e, uv, w)) / (2 * h)
        assert np.allclose(Jc, num, rtol=1e-5, atol=1e-5 * max(1, np.abs(num).max()))
        assert np.allclose(Jp, nump, rtol=1e-5, atol=1e-5 * max(1, np.abs(nump).max()))
        assert np.isclose(rho, r @ r)


def test_quat_plus_is_left_multiplicative_half_angle(oracle):
    q = synth.quat_from_rotvec([0.3, -0.2, 0.5])
    d = np.array([0.01, -0.02, 0.03])
    qp = oracle.quat_plus(q, d)
    assert np.isclose(np.linalg.norm(qp), 1.0, atol=1e-15)
    exp = synth.quat_mul(synth.quat_from_rotvec(2 * d), q)          # delta is a HALF-angle vector (A4.2)
    assert np.allclose(qp, exp, atol=1e-15)
    assert np.array_equal(oracle.quat_plus(q, np.zeros(3)), q)
    v = np.array([1.0, 2.0, 3.0])
    assert np.allclose(oracle.quat_rotate(q, v), synth.quat_to_R(q) @ v, atol=1e-14)


def test_huber_corrector(oracle):
    K4 = synth.KITTI_K4
    pose = np.array([0, 0, 0, 0, 0, 0, 1.0]); X = np.array([0.0, 0, 10])
    a = np.sqrt(5.991)
    for off, w in [(1.0, 1.0), (30.0, 1.0), (30.0, 0.2)]:
        uv = np.array([K4[2] + off, K4[3]])
        r0, J0, _, rho0 = oracle.ba_eval_obs(K4, pose, X, uv, w, robust=False)
        r1, J1, _, rho1 = oracle.ba_eval_obs(K4, pose, X, uv, w, robust=True)
        s = (w * off) ** 2
        if s <= a * a:
            assert np.allclose(r0, r1) and np.isclose(rho1, s)
        else:
            k = np.sqrt(a / np.sqrt(s))
            assert np.allclose(r1, k * r0) and np.allclose(J1, k * J0) and np.isclose(rho1, 2 * a * np.sqrt(s) - a * a)


def test_zero_noise_ba_converges_to_ground_truth(oracle):
    g = synth.make_ba_graph(1, ncam=6, npts=80, nobs=320, outlier_frac=0.0, noise=0.0, n_fixed=2)
    nobs = len(g["obs_cam"])
    w = g["obs_inv_sigma2"].astype(np.float64)
    poses, pts, s = oracle.ba_solve(g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"],
                                    g["obs_uv"], w, np.zeros(nobs, np.uint8), 100)
    assert s["final_cost"] < 1e-12 * max(1.0, s["initial_cost"])
    assert np.allclose(pts, g["pts_gt"], atol=1e-5)
    for c in range(6):
        assert np.allclose(poses[c, :3], g["poses_gt"][c, :3], atol=1e-6)
        assert min(np.abs(poses[c, 3:] - g["poses_gt"][c, 3:]).max(), np.abs(poses[c, 3:] + g["poses_gt"][c, 3:]).max()) < 1e-7


def test_loss_free_optimum_matches_scipy(oracle):
    g = synth.make_ba_graph(2, ncam=5, npts=60, nobs=240, outlier_frac=0.0, noise=1.0, n_fixed=2)
    nobs = len(g["obs_cam"]); ncam = 5; npts = 60
    w = g["obs_inv_sigma2"].astype(np.float64)
    poses, pts, s = oracle.ba_solve(g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"],
                                    g["obs_uv"], w, np.zeros(nobs, np.uint8), 200)

    base = g["poses0"].copy()

    def unpack(x):
        P = base.copy()
        for i, c in enumerate(range(2, ncam)):
            P[c, :3] = x[6 * i:6 * i + 3]
            P[c, 3:] = synth.quat_mul(synth.quat_from_rotvec(x[6 * i + 3:6 * i + 6]), base[c, 3:])
        X = x[6 * (ncam - 2):].reshape(npts, 3)
        return P, X

    def fun(x):
        P, X = unpack(x)
        r = np.zeros((nobs, 2))
        for c in range(ncam):
            m = g["obs_cam"] == c
            uv, _ = synth.project(g["K4"][c], P[c], X[g["obs_pt"][m]])
            r[m] = (g["obs_uv"][m] - uv) * w[m, None]
        return r.ravel()

    x0 = np.concatenate([np.concatenate([base[c, :3], np.zeros(3)]) for c in range(2, ncam)] + [g["pts0"].ravel()])
    sol = least_squares(fun, x0, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=20000)
    assert np.isclose(s["final_cost"], sol.cost, rtol=1e-6)
    P, X = unpack(sol.x)
    # Ceres' function_tolerance (1e-6 relative cost change) stops before scipy's 1e-15 does, so weakly
    # constrained directions (depth) agree only to <1e-2 relative; the cost agrees to 1e-6.
    assert (np.linalg.norm(pts - X, axis=1) < 1e-2 * np.linalg.norm(X, axis=1)).all()
    assert np.allclose(poses[:, :3], P[:, :3], atol=2e-2)


def test_pose_optimization_semantics(oracle):
    p = synth.make_pose_problem(3, n=400)
    n, pose, out, s = oracle.pose_optimization(p["K4"], p["pose0"], p["Xw"], p["uv"], p["inv_sigma2"])
    assert n == 400 - int(out.sum())
    assert 0.85 * 400 < n <= 400
    assert np.isclose(np.linalg.norm(pose[3:]), 1.0, atol=1e-15)
    assert np.abs(pose[:3] - p["pose_gt"][:3]).max() < 0.05
    assert s["final_cost"] < s["initial_cost"] and s["iterations"] <= 100
    # < 3 correspondences: returns 0 and leaves the pose untouched (src/CeresOptimizer.cc:330)
    n2, pose2, _, _ = oracle.pose_optimization(p["K4"], p["pose0"], p["Xw"][:2], p["uv"][:2], p["inv_sigma2"][:2])
    assert n2 == 0 and np.array_equal(pose2, p["pose0"])


def test_local_ba_two_pass_and_stop_flag(oracle):
    g = synth.make_ba_graph(4, ncam=8, npts=150, nobs=700, outlier_frac=0.05, noise=1.0, n_fixed=2)
    ncam = 8
    local = np.ones(ncam, np.uint8)
    rc, poses, pts, er, s1, s2 = oracle.local_ba(g["K4"], g["poses0"], g["cam_fixed"], local, g["pts0"], g["obs_cam"],
                                                 g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
    assert rc == 0 and s1["iterations"] <= 5 and s2["iterations"] <= 10
    assert 0.02 * len(er) < er.sum() < 0.15 * len(er)
    assert np.allclose(poses[0], g["poses0"][0], rtol=0, atol=1e-15)                     # gauge camera untouched
    assert s1["final_cost"] < s1["initial_cost"] and s2["final_cost"] < s2["initial_cost"]
    # pass 2 = pass-1 problem + duplicated loss-free blocks (F6): its initial cost exceeds pass 1's final cost
    assert s2["initial_cost"] > s1["final_cost"]
    stop = np.array([1], np.uint8)
    rc, poses, pts, er, _, _ = oracle.local_ba(g["K4"], g["poses0"], g["cam_fixed"], local, g["pts0"], g["obs_cam"],
                                               g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"], stop=stop)
    assert rc == 1 and np.array_equal(poses, g["poses0"]) and np.array_equal(pts, g["pts0"])
    # F6 switch: without the duplicated blocks pass 2 is a different (plain least-squares) problem
    rc, poses_nd, _, _, _, s2nd = oracle.local_ba(g["K4"], g["poses0"], g["cam_fixed"], local, g["pts0"], g["obs_cam"],
                                                  g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"], duplicate_blocks=False)
    assert s2nd["initial_cost"] < s2["initial_cost"]


def test_thread_count_does_not_change_results(oracle):
    po = oracle
    """The 4-thread leg of the CPU baseline (reference: options.num_threads = 4, src/CeresOptimizer.cc:516) must be the
    same computation: bit-identical poses / points / flags for 1, 3 and 4 worker threads."""
    from ceres_mono_orb_slam2_amd import synth
    g = synth.make_ba_graph(5, ncam=12, npts=600, nobs=3000, n_fixed=1)
    a = (g["K4"], g["poses0"], g["cam_fixed"], np.ones(12, np.uint8), g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
    ref = po.local_ba(*a)
    try:
        for t in (3, 4):
            po.set_ba_threads(t)
            out = po.local_ba(*a)
            assert np.array_equal(out[1], ref[1]) and np.array_equal(out[2], ref[2]) and np.array_equal(out[3], ref[3])
            assert out[4] == ref[4] and out[5] == ref[5]
    finally:
        po.set_ba_threads(1)


def test_factored_jacobian_records_reproduce_the_oracles_blocks(oracle):
    """The HIP path never stores Jc (2x6) and Jp (2x3): an observation keeps {W = Q^T Q, r = 2 RX} and h = Q^T res
    (csrc/ba_solver.hip, the comment above ld_rec8) and every block of the normal equations is rebuilt from them and the camera's
    rotation.  This checks that algebra against the oracle's own Jacobians - including the Huber corrector's scale inside Q and a
    quaternion that is NOT normalised (Jc and Jp are built from the same RX and R, so the factorisation does not need it)."""
    rng = np.random.default_rng(5)
    K4 = synth.KITTI_K4
    skew = lambda v: np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    for k in range(40):
        q = synth.quat_from_rotvec(rng.normal(0, 0.5, 3)) * (1.0 if k % 2 else 1.0 + 3e-3 * rng.normal()); t = rng.normal(0, 1, 3)
        pose = np.concatenate([t, q])
        Xc = np.array([rng.normal(0, 3), rng.normal(0, 2), rng.uniform(5, 40)])
        X = synth.quat_to_R(q / np.linalg.norm(q)).T @ (Xc - t)
        uv = rng.uniform(0, 500, 2) if k % 3 else np.array([3000.0, -2000.0])          # (far off: the Huber branch)
        w = rng.uniform(0.1, 1.0)
        res, Jc, Jp, _ = oracle.ba_eval_obs(K4, pose, X, uv, w, robust=bool(k % 3 == 0))
        RX = oracle.quat_rotate(q, X)
        # R as ba_math.h's quat_to_R builds it (no normalisation)
        x, y, z, ww = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * ww), 2 * (x * z + y * ww)],
                      [2 * (x * y + z * ww), 1 - 2 * (x * x + z * z), 2 * (y * z - x * ww)],
                      [2 * (x * z - y * ww), 2 * (y * z + x * ww), 1 - 2 * (x * x + y * y)]])
        Q = Jc[:, :3]                                                   # Jc = Q [I | -2 [RX]x]
        W = Q.T @ Q; r = 2.0 * RX; h = Q.T @ res
        assert abs(W[0, 1]) <= 1e-300                                    # the record keeps five numbers of W
        tol = lambda ref: 1e-12 * max(1.0, np.abs(ref).max())
        assert np.abs(Q @ np.hstack([np.eye(3), -skew(r)]) - Jc).max() <= tol(Jc)
        assert np.abs(Q @ R - Jp).max() <= tol(Jp)
        G = np.vstack([W, skew(r) @ W])                                  # E = Jc^T Jp = [W; [r]x W] R
        assert np.abs(G @ R - Jc.T @ Jp).max() <= tol(Jc.T @ Jp)
        K = W @ skew(r)
        B = np.block([[W, -K], [-K.T, -skew(r) @ K]])                    # Jc^T Jc
        assert np.abs(B - Jc.T @ Jc).max() <= tol(Jc.T @ Jc)
        assert np.abs(np.concatenate([h, np.cross(r, h)]) - Jc.T @ res).max() <= tol(Jc.T @ res)
        assert np.abs(R.T @ W @ R - Jp.T @ Jp).max() <= tol(Jp.T @ Jp)
        assert np.abs(R.T @ h - Jp.T @ res).max() <= tol(Jp.T @ res)
        # the Schur cross term of two observations of one point: X' R_b^T G_b^T with X' = [Y; [r_a]x Y], Y = W_a (R_a N)
        N = np.diag(rng.uniform(0.5, 2.0, 3)); N = N + 0.1 * np.ones((3, 3))
        Y = W @ (R @ N)
        Xp = np.vstack([Y, skew(r) @ Y])
        assert np.abs(Xp @ R.T @ G.T - (Jc.T @ Jp) @ N @ (Jc.T @ Jp).T).max() <= tol((Jc.T @ Jp) @ N @ (Jc.T @ Jp).T)
