"""Pins the oracle's OptimizeEssentialGraph restatement (reference src/CeresOptimizer.cc:737-957, include/CeresOptimizer.h:266-330):
Sim3 adjoint against expm conjugation, edge Jacobians against finite differences under the right-multiplicative Plus, the solve
against scipy's least squares on the same residuals."""
import numpy as np
from scipy.linalg import expm, logm
from scipy.optimize import least_squares

from oracle import pyoracle as po
from ceres_mono_orb_slam2_amd import synth
from tests.test_oracle_sim3 import hat4, mat4


def build_problem(g):
    """tangents + edge measurements Sji = Sjw * Swi as the reference forms them"""
    x0 = np.stack([po.sim3_log(S) for S in g["S_init"]])
    ej, ei, Sji = [], [], []
    for (j, i, kind) in g["edges"]:
        src = g["S_true"] if kind == "corr" else g["S_est"]
        Sji.append(po.sim3_mul(src[j], po.sim3_inverse(src[i])))
        ej.append(j); ei.append(i)
    return x0, np.array(ej, np.int32), np.array(ei, np.int32), np.stack(Sji)


def test_adjoint_against_matrix_conjugation():
    rng = np.random.default_rng(0)
    for _ in range(20):
        S = po.sim3_exp(np.concatenate([rng.normal(0, 1, 3), rng.normal(0, 0.5, 3), rng.normal(0, 0.3, 1)]))
        xi = rng.normal(0, 1, 7)
        M = mat4(S)
        C = M @ hat4(xi) @ np.linalg.inv(M)                         # hat(Adj xi)
        want = np.array([C[0, 3], C[1, 3], C[2, 3], C[2, 1], C[0, 2], C[1, 0], C[0, 0]])
        assert np.allclose(po.sim3_adj(S) @ xi, want, rtol=1e-10, atol=1e-11)


def test_edge_residual_and_jacobians():
    rng = np.random.default_rng(1)
    for _ in range(10):
        li = np.concatenate([rng.normal(0, 1, 3), rng.normal(0, 0.4, 3), rng.normal(0, 0.2, 1)])
        lj = np.concatenate([rng.normal(0, 1, 3), rng.normal(0, 0.4, 3), rng.normal(0, 0.2, 1)])
        Si, Sj = po.sim3_exp(li), po.sim3_exp(lj)
        noise = po.sim3_exp(rng.normal(0, 0.02, 7))                 # small residual: the Jr series is accurate to 3rd order
        Sji = po.sim3_mul(po.sim3_mul(noise, Sj), po.sim3_inverse(Si))
        r, J = po.eg_eval_edge(lj, li, Sji)
        E = mat4(Sji) @ mat4(Si) @ np.linalg.inv(mat4(Sj))
        Lg = np.real(logm(E))
        want = np.array([Lg[0, 3], Lg[1, 3], Lg[2, 3], Lg[2, 1], Lg[0, 2], Lg[1, 0], Lg[0, 0]])
        assert np.allclose(r, want, atol=1e-9)
        h = 1e-6
        Jn_i = np.zeros((7, 7)); Jn_j = np.zeros((7, 7))
        for k in range(7):
            d = np.zeros(7); d[k] = h
            Jn_i[:, k] = (po.eg_eval_edge(lj, po.sim3_plus(li, d), Sji)[0] - po.eg_eval_edge(lj, po.sim3_plus(li, -d), Sji)[0]) / (2 * h)
            Jn_j[:, k] = (po.eg_eval_edge(po.sim3_plus(lj, d), li, Sji)[0] - po.eg_eval_edge(po.sim3_plus(lj, -d), li, Sji)[0]) / (2 * h)
        assert np.abs(J - Jn_i).max() < 2e-5 * max(1.0, np.abs(Jn_i).max())
        assert np.abs(-J - Jn_j).max() < 2e-5 * max(1.0, np.abs(Jn_j).max())


def test_solve_distributes_the_loop_error_and_matches_scipy():
    g = synth.make_essential_graph(3, n=40, drift=0.004, n_corrected=4)
    x0, ej, ei, Sji = build_problem(g)
    x, s = po.optimize_essential_graph(x0, g["fixed"], ej, ei, Sji)
    assert s["final_cost"] < 0.05 * s["initial_cost"] and s["iterations"] <= 100 and s["termination"] in (1, 2, 3)
    assert np.array_equal(x[0], x0[0])                              # the loop keyframe is constant

    free = np.nonzero(g["fixed"] == 0)[0]

    def res(z):
        xx = x0.copy(); xx[free] = z.reshape(-1, 7)
        return np.concatenate([po.eg_eval_edge(xx[j], xx[i], S)[0] for j, i, S in zip(ej, ei, Sji)])
    ref = least_squares(res, x0[free].ravel(), method="lm", xtol=1e-14, ftol=1e-14, gtol=1e-14, max_nfev=4000)
    assert s["final_cost"] <= ref.cost * (1 + 1e-3)                 # at least as good as scipy's (finite-difference) LM from the same start
    pol = least_squares(res, x[free].ravel(), method="lm", xtol=1e-14, ftol=1e-14, gtol=1e-14, max_nfev=2000)
    assert pol.cost >= s["final_cost"] * (1 - 1e-3)                 # and scipy cannot improve the oracle's solution: it is the local minimum
    # scales of the optimised vertices move towards the truth (scale 1) after the correction
    sc0 = np.exp(x0[:, 6]); sc = np.exp(x[:, 6])
    assert np.abs(np.log(sc[5:-4])).mean() < np.abs(np.log(sc0[5:-4])).mean()


def test_write_back_arithmetic():
    g = synth.make_essential_graph(4, n=30)
    x0, ej, ei, Sji = build_problem(g)
    x, _ = po.optimize_essential_graph(x0, g["fixed"], ej, ei, Sji)
    rng = np.random.default_rng(2)
    pts = rng.normal(0, 20, (200, 3)); ref = rng.integers(0, 30, 200)
    T, P = po.essential_graph_correct(x0, x, ref, pts)
    for p in (0, 17, 199):
        k = ref[p]
        M0 = mat4(po.sim3_exp(x0[k])); M1 = mat4(po.sim3_exp(x[k]))
        want = (np.linalg.inv(M1) @ (M0 @ np.append(pts[p], 1)))[:3]
        assert np.allclose(P[p], want, rtol=1e-10, atol=1e-10)
    S = po.sim3_exp(x[7]); s = S[:4] @ S[:4]
    assert np.allclose(T[7].reshape(3, 4)[:, 3], S[4:] / s) and np.allclose(T[7].reshape(3, 4)[:, :3] * s, mat4(S)[:3, :3])
