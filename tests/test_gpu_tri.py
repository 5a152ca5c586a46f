"""GPU parity: per-match triangulation of LocalMapping::CreateNewMapPoints vs the CPU oracle.  The accept flags
must be identical and the points agree to 1e-9 relative (same one-sided Jacobi in double on both sides; libm sqrt only)."""
import numpy as np
import pytest

from tests.test_oracle_tri import make_tri_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,n,noise", [(1, 2000, 0.5), (2, 5000, 0.0), (3, 1, 0.5), (4, 20000, 1.0)])
def test_triangulate_vs_oracle(oracle, seed, n, noise):
    from ceres_mono_orb_slam2_amd import frame
    p = make_tri_problem(seed, n=max(n, 100), noise=noise)
    a = (p["T1"], p["T2"], p["K1"], p["K2"], p["kp1"][:n], p["kp2"][:n], p["ls"], p["sf"], p["ratio"])
    X, ok = frame.TriangulateMatches(*a)
    oX, ook = oracle.triangulate_matches(*a)
    assert np.array_equal(ok, ook)
    assert np.abs(X - oX).max() <= 1e-9 * max(1.0, np.abs(oX).max())
    if n >= 2000:
        assert 0.3 < ok.mean() < 0.95
