"""GPU parity at BASELINE.json's full sizes: C4 (LocalBA 100 KF x 10 k pts x 50 k obs) and C5 (GlobalBA 500 KF x 50 k pts x
250 k obs, single and as 8 batched sub-maps) against the CPU oracle.

What "parity" can mean at these sizes (measured, tools/explore_ba_fullsize.py):

* SURVEY 8(d)'s synthetic graphs are ill-conditioned BY CONSTRUCTION: a 500-keyframe odometry chain whose gauge is pinned
  at one end (no loop closures) has weakly constrained bending modes, and tracks of 2-3 views 0.8 m apart at up to 60 m depth
  have sub-degree parallax.  The ORACLE ITSELF moves by 1e-4 (cost) / 5e-8 (poses) after 3 LM iterations and by 1e-2 / 0.4
  after 10 when its input points are perturbed by ONE ULP (2e-16 relative).  No two implementations with different summation
  orders -- including two builds of the reference -- can agree better than that cloud over a trajectory.
* So the tests split the claim in two:
    1. ONE LM ITERATION (the whole operator: Jacobians, Schur complement, 2994 x 2994 MFMA Cholesky, back-substitution, step
       acceptance) from states shared with the oracle -- the start and the oracle's own iterates after 3 and 10 iterations --
       must agree to the ordinary bars (cost 1e-9, poses 1e-7, points 1e-5 of their norm).  Measured: 2e-10 / 7e-9.
    2. TRAJECTORIES (3 and 10 iterations; LocalBA's 5 + 10) must have identical discrete outputs (iteration counts, accepted
       steps, termination, erase flags) and continuous outputs inside `CLOUD` x the oracle's own rounding cloud, which the
       test measures live by re-running the oracle on 1-ulp-perturbed inputs.  One exception, C5 at 10 iterations: once that
       cloud is wider than `DECORRELATED` (the oracle's perturbed runs end 1.4 ... 1.8 % apart in cost: the trajectories have
       separated by iteration 8) the number of ACCEPTED steps may differ by one - which step lands near the acceptance
       threshold is then a property of the rounding, not of the algorithm; iteration count and termination must still agree.
* With the gauge fixed at both ends of a short window (C4 with two fixed keyframes) the ordinary bars hold for the whole
  two-pass solve; that case is asserted flat.

Reference: src/CeresOptimizer.cc:59-225 (BundleAdjustment), :344-599 (LocalBundleAdjustment)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from ceres_mono_orb_slam2_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL_COST, RTOL_X, RTOL_PT = 1e-9, 1e-7, 1e-5
DECORRELATED = 1e-3   # relative width of the oracle's own 1-ulp cost cloud beyond which its trajectories no longer coincide step for step
CLOUD = 10.0          # a trajectory may differ from the oracle's by at most this multiple of the oracle's own 1-ulp rounding cloud


def _threads(oracle):
    try:
        oracle.set_ba_threads(min(len(os.sched_getaffinity(0)), 16))     # results do not depend on it (test_oracle_ba.py)
    except AttributeError:
        oracle.set_ba_threads(8)


def _pt_err(a, b):
    a = np.asarray(a).reshape(-1, 3); b = np.asarray(b).reshape(-1, 3)
    return float((np.linalg.norm(a - b, axis=1) / np.maximum(1.0, np.linalg.norm(b, axis=1))).max())


def _ulp_perturbed(x, seed):
    rng = np.random.default_rng(seed)
    return x * (1.0 + rng.uniform(-1, 1, x.shape) * 2e-16)


def _c5(seed=1000, n_fixed=1):
    g = synth.make_ba_graph(seed, ncam=500, npts=50000, nobs=250000, n_fixed=n_fixed)
    n = len(g["obs_cam"])
    w = np.asarray(g["obs_inv_sigma2"], np.float32).astype(np.float64)       # F7: weight = invSigma2
    return g, (g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], w, np.ones(n, np.uint8))


def _with(base, poses, pts):
    return (base[0], poses, base[2], pts) + tuple(base[4:])


def _discrete(s):
    return (s["iterations"], s["successful_steps"], s["termination"])


# ------------------------------------------------------------------------------------------------------- C5, one iteration
def test_globalba_c5_one_iteration_from_shared_states(oracle):
    """The full LM iteration at 500 KF / 2994 unknowns in the reduced system, from three states on the oracle's own
    trajectory.  Ordinary bars."""
    from ceres_mono_orb_slam2_amd import optimizer
    _threads(oracle)
    g, base = _c5()
    states = [(g["poses0"], g["pts0"])]
    for k in (3, 10):
        p, x, _ = oracle.ba_solve(*base, k)
        states.append((p, x))
    for poses0, pts0 in states:
        a = _with(base, poses0, pts0)
        poses, pts, s = optimizer.bundle_adjustment(*a, 1)
        oposes, opts, os_ = oracle.ba_solve(*a, 1)
        assert _discrete(s) == _discrete(os_)
        assert abs(s["initial_cost"] - os_["initial_cost"]) <= RTOL_COST * os_["initial_cost"]
        assert abs(s["final_cost"] - os_["final_cost"]) <= RTOL_COST * os_["final_cost"]
        assert np.abs(poses - oposes).max() <= RTOL_X * max(1.0, np.abs(oposes).max())
        assert _pt_err(pts, opts) <= RTOL_PT


# ------------------------------------------------------------------------------------------------------- C5, trajectories
@pytest.mark.parametrize("iters", [3, 10])
def test_globalba_c5_trajectory_inside_the_oracles_rounding_cloud(oracle, iters):
    from ceres_mono_orb_slam2_amd import optimizer
    _threads(oracle)
    g, base = _c5()
    poses, pts, s = optimizer.bundle_adjustment(*base, iters)
    oposes, opts, os_ = oracle.ba_solve(*base, iters)
    assert abs(s["initial_cost"] - os_["initial_cost"]) <= RTOL_COST * os_["initial_cost"]
    cloud_cost, cloud_pose = 0.0, 0.0
    for seed in (1, 2):
        pp, _, ps = oracle.ba_solve(*_with(base, g["poses0"], _ulp_perturbed(g["pts0"], seed)), iters)
        cloud_cost = max(cloud_cost, abs(ps["final_cost"] - os_["final_cost"]) / os_["final_cost"])
        cloud_pose = max(cloud_pose, np.abs(pp - oposes).max())
    d_cost = abs(s["final_cost"] - os_["final_cost"]) / os_["final_cost"]
    d_pose = np.abs(poses - oposes).max()
    print("C5 %d iterations: GPU-oracle cost %.2e pose %.2e | oracle 1-ulp cloud cost %.2e pose %.2e" % (iters, d_cost, d_pose, cloud_cost, cloud_pose))
    # The accept / reject sequence is compared while the oracle's own trajectories still coincide.  Once a 1-ulp change of the input
    # moves the oracle's final cost by more than DECORRELATED (C5 after ~8 iterations: 2e-4 at iteration 7, 2.5e-2 at 8, measured
    # with both arithmetic orders the HIP path has had), a step near the acceptance threshold legitimately flips: the iteration
    # count and the termination must still agree, the number of accepted steps may differ by one.
    if cloud_cost <= DECORRELATED:
        assert _discrete(s) == _discrete(os_)
    else:
        assert (s["iterations"], s["termination"]) == (os_["iterations"], os_["termination"])
        assert abs(s["successful_steps"] - os_["successful_steps"]) <= 1
    assert d_cost <= max(RTOL_COST, CLOUD * cloud_cost)
    assert d_pose <= max(RTOL_X, CLOUD * cloud_pose)
    assert s["final_cost"] < 0.7 * s["initial_cost"]                          # and it is a descent, not a stall


_LA_SCRIPT = r"""
import sys, json, numpy as np
sys.path.insert(0, sys.argv[1])
from ceres_mono_orb_slam2_amd import synth, optimizer
g = synth.make_ba_graph(1000, ncam=500, npts=50000, nobs=250000, n_fixed=1)
n = len(g["obs_cam"]); w = np.asarray(g["obs_inv_sigma2"], np.float32).astype(np.float64)
out = {}
for it in (1, 3):
    poses, pts, s = optimizer.bundle_adjustment(g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], w, np.ones(n, np.uint8), it)
    np.save(sys.argv[2] + "_%d.npy" % it, poses)
    out[str(it)] = s
print("RESULT " + json.dumps(out))
"""


def test_globalba_c5_both_factorisation_schemes(oracle, tmp_path):
    """C5's reduced system (94 block rows, a band of 3) takes the look-ahead family since round 5 (one persistent launch that walks the
    skyline); ORBHIP_BA_LOOKAHEAD=0 (read once per process, hence the subprocess) keeps it on the two-level blocked Cholesky, which
    still serves the large systems with a wide skyline.  One iteration: ordinary bars against the oracle for BOTH; three iterations:
    both inside the rounding cloud (see the module docstring)."""
    from ceres_mono_orb_slam2_amd import optimizer
    _threads(oracle)
    g, base = _c5()
    stem = str(tmp_path / "la")
    r = subprocess.run([sys.executable, "-c", _LA_SCRIPT, ROOT, stem], env=dict(os.environ, ORBHIP_BA_LOOKAHEAD="0"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    la = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
    for it in (1, 3):
        oposes, opts, os_ = oracle.ba_solve(*base, it)
        poses, pts, s = optimizer.bundle_adjustment(*base, it)
        cloud_cost, cloud_pose = 0.0, 0.0
        if it > 1:
            pp, _, ps = oracle.ba_solve(*_with(base, g["poses0"], _ulp_perturbed(g["pts0"], 1)), it)
            cloud_cost = abs(ps["final_cost"] - os_["final_cost"]) / os_["final_cost"]; cloud_pose = np.abs(pp - oposes).max()
        for name, pz, sz in (("look-ahead", poses, s), ("two-level", np.load(stem + "_%d.npy" % it), la[str(it)])):
            assert _discrete(sz) == _discrete(os_), name
            assert abs(sz["final_cost"] - os_["final_cost"]) / os_["final_cost"] <= max(RTOL_COST, CLOUD * cloud_cost), (name, it)
            assert np.abs(pz - oposes).max() <= max(RTOL_X, CLOUD * cloud_pose), (name, it)


def test_globalba_c5_eight_submaps_batched_equal_single_calls():
    """BASELINE configs[4]: 8 independent 500-KF sub-maps.  Through ba_solve_batch (one lockstep launch sequence) every
    sub-map must come out BIT-IDENTICAL to its own single ba_solve call (same kernels, same per-problem order)."""
    from ceres_mono_orb_slam2_amd import optimizer
    probs = []
    for q in range(8):
        g, base = _c5(seed=2000 + q)
        probs.append(base)
    batched = optimizer.bundle_adjustment_batch(probs, n_iterations=3)
    for q in (0, 3, 7):
        poses, pts, s = optimizer.bundle_adjustment(*probs[q], 3)
        bposes, bpts, bs = batched[q]
        assert np.array_equal(poses, bposes) and np.array_equal(pts, bpts), q
        assert bs == s, q
    assert all(b[2]["iterations"] == 3 and b[2]["final_cost"] < b[2]["initial_cost"] for b in batched)


# ------------------------------------------------------------------------------------------------------- C4 LocalBA
def _c4(n_fixed):
    g = synth.make_ba_graph(0, ncam=100, npts=10000, nobs=50000, n_fixed=n_fixed)
    return g, (g["K4"], g["poses0"], g["cam_fixed"], np.ones(100, np.uint8), g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])


@pytest.mark.parametrize("dup", [True, False])
def test_localba_c4_two_fixed_keyframes_ordinary_bars(oracle, dup):
    """C4 size with the gauge held by two keyframes (what LocalBundleAdjustment normally has: its fixed keyframes are the
    covisible ones outside the local window, src/CeresOptimizer.cc:388-406): erase flags and iteration counts identical,
    cost 1e-9, poses 1e-7 over the whole 5 + 10 iteration two-pass solve."""
    from ceres_mono_orb_slam2_amd import optimizer
    _threads(oracle)
    g, a = _c4(2)
    ab, poses, pts, er, s1, s2 = optimizer.local_bundle_adjustment(*a, duplicate_blocks=dup)
    rc, oposes, opts, oer, o1, o2 = oracle.local_ba(*a, duplicate_blocks=dup)
    assert ab == 0 and rc == 0
    assert np.array_equal(er, oer) and er.sum() > 1000
    assert _discrete(s1) == _discrete(o1) and _discrete(s2) == _discrete(o2) and s1["iterations"] == 5 and s2["iterations"] == 10
    for sa, sb in ((s1, o1), (s2, o2)):
        assert abs(sa["final_cost"] - sb["final_cost"]) <= RTOL_COST * sb["final_cost"]
    assert np.abs(poses - oposes).max() <= RTOL_X * max(1.0, np.abs(oposes).max())


@pytest.mark.parametrize("dup", [True, False])
def test_localba_c4_gauge_keyframe_only(oracle, dup):
    """SURVEY 8(d)'s C4: every keyframe free except the gauge keyframe 0, so the monocular scale is held only by the LM
    damping.  Discrete outputs (erase flags, iterations, termination) identical; pass 1 (5 iterations) at 1e-8 on cost; the
    final state inside the oracle's own 1-ulp rounding cloud."""
    from ceres_mono_orb_slam2_amd import optimizer
    _threads(oracle)
    g, a = _c4(1)
    ab, poses, pts, er, s1, s2 = optimizer.local_bundle_adjustment(*a, duplicate_blocks=dup)
    rc, oposes, opts, oer, o1, o2 = oracle.local_ba(*a, duplicate_blocks=dup)
    assert ab == 0 and rc == 0
    assert np.array_equal(er, oer) and er.sum() > 1000
    assert _discrete(s1) == _discrete(o1) and _discrete(s2) == _discrete(o2) and s1["iterations"] == 5 and s2["iterations"] == 10
    assert abs(s1["final_cost"] - o1["final_cost"]) <= 1e-8 * o1["final_cost"]
    cloud_cost, cloud_pose = 0.0, 0.0
    for seed in (1, 2, 3):
        ap = a[:4] + (_ulp_perturbed(g["pts0"], seed),) + a[5:]
        _, pp, _, per, _, p2 = oracle.local_ba(*ap, duplicate_blocks=dup)
        cloud_cost = max(cloud_cost, abs(p2["final_cost"] - o2["final_cost"]) / o2["final_cost"])
        cloud_pose = max(cloud_pose, np.abs(pp - oposes).max())
    d_cost = abs(s2["final_cost"] - o2["final_cost"]) / o2["final_cost"]
    d_pose = np.abs(poses - oposes).max()
    print("C4 dup=%s: GPU-oracle cost %.2e pose %.2e | oracle 1-ulp cloud cost %.2e pose %.2e" % (dup, d_cost, d_pose, cloud_cost, cloud_pose))
    assert d_cost <= max(RTOL_COST, CLOUD * cloud_cost)
    assert d_pose <= max(RTOL_X, CLOUD * cloud_pose)
