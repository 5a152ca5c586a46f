"""GPU parity: CeresOptimizer::OptimizeSim3 (reference src/CeresOptimizer.cc:601-735) through the C ABI vs the CPU oracle.

What can and cannot agree (tests/test_oracle_sim3.py pins both facts on the oracle alone):
  * The scale column of the header's Jacobian, J_camera * p, is analytically ZERO (a projection does not change when
    the camera-frame point is scaled), so H[6][6] and g[6] are pure rounding residue.  Ceres' minimum LM diagonal
    (1e-6 / radius) then turns that residue into a scale step of ~1e-3 per iteration: the reference's own sigma update
    is floating-point noise amplified ~1e10, different for any two builds of the reference itself.
  * With the reference's equal weights the inverse terms' sign-flipped Jacobian cancels the forward gradient and the
    LM usually rejects every step, so that noise never reaches the iterate.  Whenever the oracle accepts NO step the
    answer is build-independent and everything must agree: iteration count, termination, every outlier flag and the
    inlier count IDENTICAL, costs within 1e-9 relative, S12 within 1e-9 (measured on MI355X: 2e-15 / 3e-16).
  * As soon as a step is accepted the noise-driven sigma walk makes trajectories diverge (the oracle diverges from
    ITSELF under a 1e-13 input perturbation), so then only the initial cost is compared tightly; the final cost must
    agree to 1e-2 relative, R to 1e-3, t and scale to 5e-2 (t_z and the scale walk together), outlier counts to 10 %.
    "fwd" cases weight the inverse terms out to force this regime (~10 accepted steps, real convergence).
"""
import os

import numpy as np
import pytest

from ceres_mono_orb_slam2_amd import synth

pytestmark = pytest.mark.gpu
RTOL_COST, RTOL_X = 1e-9, 1e-9
ARGS = ("K1", "K2", "s12_0", "P3D2c", "obs1", "inv_sigma2_1", "P3D1c", "obs2", "inv_sigma2_2")


def _close(a, b, rtol):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() <= rtol * max(1.0, np.abs(b).max())


def _variant(pr, kind):
    pr = dict(pr)
    if kind == "fwd":
        pr["inv_sigma2_2"] = pr["inv_sigma2_2"] * np.float32(1e-4)
    elif kind == "inv":
        pr["inv_sigma2_1"] = pr["inv_sigma2_1"] * np.float32(1e-4)
    return pr


def _mat(S):
    q = S[:4]; n2 = q @ q
    x, y, z, w = q / np.sqrt(n2)
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return n2, R, S[4:]


def _compare(got, want, n_terms):
    """got / want = (n_inliers, S12, outlier, summary)."""
    n, S, out, s = got
    on, oS, oout, os_ = want
    assert _close(s["initial_cost"], os_["initial_cost"], 1e-12)
    assert (s["successful_steps"] == 0) == (os_["successful_steps"] == 0)
    if os_["successful_steps"] == 0:                                  # build-independent regime: everything agrees
        assert (s["iterations"], s["termination"]) == (os_["iterations"], os_["termination"])
        assert _close(s["final_cost"], os_["final_cost"], RTOL_COST) and _close(s["final_radius"], os_["final_radius"], 1e-9)
        assert _close(S, oS, RTOL_X)
        assert np.array_equal(out, oout) and n == on
        return "strict"
    assert s["final_cost"] <= s["initial_cost"] and abs(s["final_cost"] - os_["final_cost"]) <= 1e-2 * os_["final_cost"]
    (sc, R, t), (osc, oR, ot) = _mat(S), _mat(oS)
    assert np.abs(R - oR).max() < 1e-3 and np.abs(t - ot).max() < 5e-2 and abs(sc - osc) < 5e-2
    assert abs(int(out.sum()) - int(oout.sum())) <= max(3, 0.1 * len(out))     # the check is scale-sensitive (:695-709)
    good = len(out) - int(out.sum())
    assert n == (good if good >= 10 else 0)
    return "loose"


def _check(oracle, pr, th2=10.0, expect=None):
    from ceres_mono_orb_slam2_amd import optimizer
    a = [pr[k] for k in ARGS]
    got = optimizer.optimize_sim3(*a, th2=th2)
    want = oracle.optimize_sim3(*a, th2=th2)
    mode = _compare(got, want, len(pr["P3D2c"]))
    if expect:
        assert mode == expect
    return got


@pytest.mark.parametrize("seed,n,scale,kind", [(0, 120, 1.15, "both"), (1, 120, 1.0, "both"), (2, 300, 1.0, "both"), (3, 120, 1.05, "both"),
                                               (4, 60, 0.9, "both"), (5, 120, 1.0, "inv"), (6, 25, 1.0, "both"), (7, 1000, 1.02, "both"),
                                               (8, 2000, 1.0, "both"), (9, 120, 1.3, "inv")])
def test_optimize_sim3_vs_oracle(oracle, seed, n, scale, kind):
    for perturb in ((0.02, 0.1, 0.03), (0.002, 0.01, 0.003)):        # far start: ~all outliers; near start: mixed flags
        _check(oracle, _variant(synth.make_sim3_problem(seed, n=n, scale=scale, perturb=perturb), kind))


@pytest.mark.parametrize("seed", [44, 45, 47, 49, 50, 51, 53, 54, 55, 56, 59])
def test_optimize_sim3_mixed_flags_strict(oracle, seed):
    """Near starts where the oracle accepts no step: 30-80 % of the matches are inliers, every flag must agree."""
    got = _check(oracle, synth.make_sim3_problem(seed, n=150, scale=1.02, perturb=(0.002, 0.01, 0.003)), expect="strict")
    assert 40 <= got[0] <= 125


@pytest.mark.parametrize("seed,n,scale", [(2, 300, 1.0), (3, 120, 1.05), (4, 60, 0.9), (7, 1000, 1.02)])
def test_optimize_sim3_forward_dominated(oracle, seed, n, scale):
    got = _check(oracle, _variant(synth.make_sim3_problem(seed, n=n, scale=scale), "fwd"), expect="loose")
    assert got[3]["successful_steps"] >= 3 and got[3]["final_cost"] < 0.5 * got[3]["initial_cost"]


def test_optimize_sim3_other_threshold_and_fix_scale_ignored(oracle):
    from ceres_mono_orb_slam2_amd import optimizer
    pr = synth.make_sim3_problem(11, n=150, scale=1.0)
    _check(oracle, pr, th2=4.0)
    _check(oracle, pr, th2=25.0)
    a = [pr[k] for k in ARGS]
    r0 = optimizer.optimize_sim3(*a, fix_scale=False)
    r1 = optimizer.optimize_sim3(*a, fix_scale=True)                 # bFixScale is never read (:604)
    assert r0[0] == r1[0] and np.array_equal(r0[1], r1[1])


def test_optimize_sim3_degenerate(oracle):
    from ceres_mono_orb_slam2_amd import optimizer
    pr = synth.make_sim3_problem(9, n=8, outlier_frac=0.0, scale=1.0, noise=0.0, perturb=(0, 0, 0))
    n, S, out, s = _check(oracle, pr)
    assert n == 0 and out.sum() == 0                                  # 8 perfect matches: still < 10 inliers (:731)
    e3 = np.zeros((0, 3)); e2 = np.zeros((0, 2)); e1 = np.zeros(0, np.float32)
    n, S, out, s = optimizer.optimize_sim3(pr["K1"], pr["K2"], pr["s12_0"], e3, e2, e1, e3, e2, e1)
    on, oS, _, os_ = oracle.optimize_sim3(pr["K1"], pr["K2"], pr["s12_0"], e3, e2, e1, e3, e2, e1)
    assert n == 0 == on and s["iterations"] == 0 and _close(S, oS, 1e-14)


def test_optimize_sim3_batch_matches_single(oracle):
    import torch
    from ceres_mono_orb_slam2_amd import optimizer
    prs = [_variant(synth.make_sim3_problem(20 + i, n=40 + 37 * i, scale=1.0 + 0.03 * (i % 3)), "fwd" if i % 2 else "both") for i in range(9)]
    dev = torch.device("cuda:0")
    cat = lambda k, dt: torch.tensor(np.concatenate([np.asarray(p[k]).reshape(len(p["P3D2c"]), -1) for p in prs]), dtype=dt, device=dev).contiguous()
    stack = lambda k: torch.tensor(np.stack([p[k] for p in prs]), dtype=torch.float64, device=dev).contiguous()
    offs = np.concatenate([[0], np.cumsum([len(p["P3D2c"]) for p in prs])]).astype(np.int32)
    d_s12 = stack("s12_0")
    th2 = torch.full((len(prs),), 10.0, dtype=torch.float64, device=dev)
    outl, ninl, summ = optimizer.optimize_sim3_batch(stack("K1"), stack("K2"), d_s12, cat("P3D2c", torch.float64), cat("obs1", torch.float64),
                                                     cat("inv_sigma2_1", torch.float32).flatten(), cat("P3D1c", torch.float64),
                                                     cat("obs2", torch.float64), cat("inv_sigma2_2", torch.float32).flatten(),
                                                     torch.tensor(offs, device=dev), th2)
    torch.cuda.synchronize()
    outl = outl.cpu().numpy(); ninl = ninl.cpu().numpy(); S = d_s12.cpu().numpy()
    for i, p in enumerate(prs):
        n1, S1, o1, s1 = optimizer.optimize_sim3(*[p[k] for k in ARGS])
        assert ninl[i] == n1 and np.array_equal(outl[offs[i]:offs[i + 1]], o1) and np.array_equal(S[i], S1)


def test_golden_sim3_fixture():
    from ceres_mono_orb_slam2_amd import optimizer
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sim3_small.npz"))
    for i in range(3):
        n, S, out, s = optimizer.optimize_sim3(*[g["p%d_%s" % (i, k)] for k in ARGS])
        assert n == int(g["p%d_n_inliers" % i]) and np.array_equal(out, g["p%d_outlier" % i])
        assert _close(S, g["p%d_s12" % i], RTOL_X) and s["iterations"] == int(g["p%d_iters" % i]) and s["successful_steps"] == 0
        assert _close(s["final_cost"], g["p%d_cost" % i][1], RTOL_COST)
    assert _close(optimizer.sim3_exp(g["exp_in"]), g["exp_out"], 1e-14)
