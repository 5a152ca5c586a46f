"""GPU parity: CeresOptimizer::OptimizeEssentialGraph through the C ABI vs the CPU oracle.  Same tolerances as the
bundle-adjustment tests: identical iteration counts / termination, costs to 1e-9 relative, tangents to 1e-7."""
import numpy as np
import pytest

from ceres_mono_orb_slam2_amd import synth
from tests.test_oracle_essential_graph import build_problem

pytestmark = pytest.mark.gpu


def _close(a, b, rtol):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() <= rtol * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("seed,n,drift,ncorr", [(3, 40, 0.004, 4), (5, 200, 0.002, 6), (6, 12, 0.01, 2), (7, 300, 0.001, 8)])
def test_essential_graph_vs_oracle(oracle, seed, n, drift, ncorr):
    from ceres_mono_orb_slam2_amd import optimizer
    g = synth.make_essential_graph(seed, n=n, drift=drift, n_corrected=ncorr)
    x0, ej, ei, Sji = build_problem(g)
    x, s = optimizer.optimize_essential_graph(x0, g["fixed"], ej, ei, Sji)
    ox, os_ = oracle.optimize_essential_graph(x0, g["fixed"], ej, ei, Sji)
    assert (s["iterations"], s["successful_steps"], s["termination"]) == (os_["iterations"], os_["successful_steps"], os_["termination"])
    assert _close(s["initial_cost"], os_["initial_cost"], 1e-9) and abs(s["final_cost"] - os_["final_cost"]) <= 1e-9 * max(os_["final_cost"], 1e-9) + 1e-15
    assert _close(x, ox, 1e-7)
    assert np.array_equal(x[0], x0[0]) and s["final_cost"] < 0.2 * s["initial_cost"]


def test_essential_graph_degenerate_and_write_back(oracle):
    from ceres_mono_orb_slam2_amd import optimizer
    g = synth.make_essential_graph(8, n=30)
    x0, ej, ei, Sji = build_problem(g)
    # no edges: nothing moves; every vertex constant: nothing moves
    x, s = optimizer.optimize_essential_graph(x0, g["fixed"], ej[:0], ei[:0], Sji[:0])
    assert np.array_equal(x, x0) and s["iterations"] == 0
    x, s = optimizer.optimize_essential_graph(x0, np.ones(30, np.uint8), ej, ei, Sji)
    assert np.array_equal(x, x0)
    x, _ = optimizer.optimize_essential_graph(x0, g["fixed"], ej, ei, Sji)
    rng = np.random.default_rng(2)
    pts = rng.normal(0, 20, (5000, 3)); ref = rng.integers(0, 30, 5000)
    T, P = optimizer.essential_graph_correct(x0, x, ref, pts)
    oT, oP = oracle.essential_graph_correct(x0, x, ref, pts)
    assert _close(T.reshape(30, 12), oT, 1e-12) and _close(P, oP, 1e-12)
