"""Pins oracle/tri_oracle.cpp (per-match triangulation of LocalMapping::CreateNewMapPoints, reference
src/LocalMapping.cc:267-378): the Jacobi null vector against LAPACK, the triangulated points against ground truth."""
import numpy as np

from oracle import pyoracle as po
from ceres_mono_orb_slam2_amd import synth


def make_tri_problem(seed, n=2000, noise=0.5, baseline=0.8):
    """two KITTI keyframes a baseline apart, n matched keypoints of 3-D points in front of both (+ outliers / low parallax)."""
    rng = np.random.default_rng(seed)
    K = synth.KITTI_K4.astype(np.float32)
    q1 = synth.quat_from_rotvec(rng.normal(0, 0.02, 3)); q2 = synth.quat_from_rotvec(rng.normal(0, 0.02, 3))
    R1 = synth.quat_to_R(q1); R2 = synth.quat_to_R(q2)
    t1 = rng.normal(0, 0.1, 3); C2 = -R1.T @ t1 + np.array([baseline, 0.05, 0.3]); t2 = -R2 @ C2
    T1 = np.hstack([R1, t1[:, None]]); T2 = np.hstack([R2, t2[:, None]])
    uv = np.stack([rng.uniform(50, 1190, n), rng.uniform(30, 340, n)], 1); z = rng.uniform(3, 60, n)
    z[: n // 10] = rng.uniform(300, 3000, n // 10)                    # very low parallax -> rejected (:296)
    Pc = np.stack([(uv[:, 0] - K[2]) / K[0] * z, (uv[:, 1] - K[3]) / K[1] * z, z], 1)
    X = (Pc - t1) @ R1                                                   # world
    P2 = X @ R2.T + t2
    uv2 = np.stack([K[0] * P2[:, 0] / P2[:, 2] + K[2], K[1] * P2[:, 1] / P2[:, 2] + K[3]], 1)
    o1 = rng.integers(0, 8, n); o2 = np.clip(o1 + rng.integers(-1, 2, n), 0, 7)
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32); ls = (sf * sf).astype(np.float32)
    kp1 = np.concatenate([uv + rng.normal(0, noise, (n, 2)) * sf[o1][:, None], o1[:, None]], 1).astype(np.float32)
    kp2 = np.concatenate([uv2 + rng.normal(0, noise, (n, 2)) * sf[o2][:, None], o2[:, None]], 1).astype(np.float32)
    bad = rng.choice(n, n // 12, replace=False); kp2[bad, :2] += rng.uniform(10, 40, (len(bad), 2)).astype(np.float32)   # wrong matches
    o2[: 40] = 7; o1[: 40] = 0; kp1[:40, 2] = 0; kp2[:40, 2] = 7        # scale-inconsistent pairs
    isbad = np.zeros(n, bool); isbad[bad] = True
    return dict(T1=T1, T2=T2, K1=K, K2=K, kp1=kp1, kp2=kp2, ls=ls, sf=sf, ratio=np.float32(1.5) * np.float32(1.2), X=X, bad=isbad)


def test_null_vector_against_lapack():
    rng = np.random.default_rng(0)
    for _ in range(200):
        A = rng.normal(0, 1, (4, 4)); A[3] = A[0] * rng.normal() + A[1] * rng.normal() + A[2] * rng.normal() + rng.normal(0, 1e-6, 4)
        x = po.null_vector4(A)
        v = np.linalg.svd(A)[2][3]
        x = x / x[np.argmax(np.abs(v))] * v[np.argmax(np.abs(v))]
        assert np.abs(x - v).max() < 1e-9 and abs(np.linalg.norm(x) - 1) < 1e-12


def test_triangulation_gates_and_accuracy():
    p = make_tri_problem(1)
    X, ok = po.triangulate_matches(p["T1"], p["T2"], p["K1"], p["K2"], p["kp1"], p["kp2"], p["ls"], p["sf"], p["ratio"])
    n = len(ok)
    assert not ok[: n // 10].any()                                      # low parallax rejected
    assert not ok[:40].any()                                            # octave ratio 1.2^-7 vs distance ratio ~1: scale gate
    good = ok.astype(bool)
    assert 0.3 < good.mean() < 0.9
    rel = np.linalg.norm(X[good] - p["X"][good], axis=1) / np.linalg.norm(p["X"][good], axis=1)
    assert np.median(rel) < 0.05
    # noise-free: exact recovery
    q = make_tri_problem(2, noise=0.0)
    X, ok = po.triangulate_matches(q["T1"], q["T2"], q["K1"], q["K2"], q["kp1"], q["kp2"], q["ls"], q["sf"], q["ratio"])
    g = ok.astype(bool) & ~q["bad"]
    assert g.sum() > 500 and np.abs(X[g] - q["X"][g]).max() < 1e-2 * np.abs(q["X"][g]).max()      # keypoints are float32 pixels
