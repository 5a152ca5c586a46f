"""GPU parity at C4 / C5 size on reduced camera systems with the STRUCTURE a map of the reference has, beside
SURVEY 8(d)'s odometry band of tests/test_gpu_ba_full_size.py:

  covis  tracks over windows of 15-30 keyframes with gaps + a current keyframe (the last one) that shares >= 15 landmarks with every
         local keyframe (src/CeresOptimizer.cc:353-363: the local window IS the current keyframe's covisibility list): a band of
         ~6 tiles under a dense last block row;
  dense  every keyframe pair shares landmarks: a full reduced system (every tile update of the factorisation runs);
  loop   a 500-keyframe chain whose ends share landmarks (GlobalBundleAdjustemnt after a loop closure, src/LoopClosing.cc:656): the
         last block rows reach back to column 0.

Per structure: ONE LM iteration (Jacobians, Schur complement, the whole factorisation, back-substitution, step acceptance) from
the start and from the oracle's own iterate must agree with the oracle at cost 1e-9 / poses 1e-7 / points 1e-5; the two-pass
LocalBA must give identical discrete outputs (erase flags, iteration counts, accepted steps, termination) and a final state
inside the oracle's own 1-ulp rounding cloud; a lockstep batch must equal the single calls bit for bit (k_chol_wg against
k_chol_persist on the same skyline).  The form the factorisation took is asserted from ba_get_last_plan."""
import os

import numpy as np
import pytest

from ceres_mono_orb_slam2_amd import synth

pytestmark = pytest.mark.gpu
RTOL_COST, RTOL_X, RTOL_PT = 1e-9, 1e-7, 1e-5
CLOUD = 10.0


def _threads(oracle):
    try:
        oracle.set_ba_threads(min(len(os.sched_getaffinity(0)), 16))
    except AttributeError:
        oracle.set_ba_threads(8)


def _pt_err(a, b):
    a = np.asarray(a).reshape(-1, 3); b = np.asarray(b).reshape(-1, 3)
    return float((np.linalg.norm(a - b, axis=1) / np.maximum(1.0, np.linalg.norm(b, axis=1))).max())


def _discrete(s):
    return (s["iterations"], s["successful_steps"], s["termination"])


def _ulp(x, seed):
    return x * (1.0 + np.random.default_rng(seed).uniform(-1, 1, x.shape) * 2e-16)


def _ba_args(g):
    n = len(g["obs_cam"])
    w = np.asarray(g["obs_inv_sigma2"], np.float32).astype(np.float64)
    return (g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], w, np.ones(n, np.uint8))


def _local_args(g):
    return (g["K4"], g["poses0"], g["cam_fixed"], np.ones(len(g["cam_fixed"]), np.uint8), g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])


def _one_iteration(optimizer, oracle, base, states):
    for poses0, pts0 in states:
        a = (base[0], poses0, base[2], pts0) + tuple(base[4:])
        poses, pts, s = optimizer.bundle_adjustment(*a, 1)
        oposes, opts, os_ = oracle.ba_solve(*a, 1)
        assert _discrete(s) == _discrete(os_)
        assert abs(s["initial_cost"] - os_["initial_cost"]) <= RTOL_COST * os_["initial_cost"]
        assert abs(s["final_cost"] - os_["final_cost"]) <= RTOL_COST * os_["final_cost"]
        assert np.abs(poses - oposes).max() <= RTOL_X * max(1.0, np.abs(oposes).max())
        assert _pt_err(pts, opts) <= RTOL_PT


@pytest.mark.parametrize("structure", ["covis", "dense"])
def test_c4_structures_one_iteration_vs_oracle(oracle, structure):
    from ceres_mono_orb_slam2_amd import optimizer
    _threads(oracle)
    g = synth.make_ba_graph_covis(100, ncam=100, npts=10000, nobs=50000, structure=structure)
    base = _ba_args(g)
    p3, x3, _ = oracle.ba_solve(*base, 3)
    _one_iteration(optimizer, oracle, base, [(g["poses0"], g["pts0"]), (p3, x3)])
    plan = optimizer.get_last_plan()
    assert plan["lookahead_form"] == "k_chol_persist" and plan["two_level_form"] == "none"      # 19 block rows: one persistent launch walks the skyline
    assert plan["band_tiles"] >= (17 if structure == "dense" else 5)


@pytest.mark.parametrize("structure", ["covis", "dense"])
def test_c4_structures_local_ba_vs_oracle_and_batch_equals_single(oracle, structure):
    """The reference's two-pass LocalBundleAdjustment (src/CeresOptimizer.cc:344-599) on a covisibility-structured / dense local map:
    discrete outputs identical, pass 1 at 1e-8, the final state inside the oracle's 1-ulp cloud; then the same map inside a
    lockstep batch of 33 (k_chol_wg: one workgroup per system, list of active rows per column) bit for bit."""
    from ceres_mono_orb_slam2_amd import optimizer
    _threads(oracle)
    g = synth.make_ba_graph_covis(101, ncam=100, npts=10000, nobs=50000, structure=structure)
    a = _local_args(g)
    ab, poses, pts, er, s1, s2 = optimizer.local_bundle_adjustment(*a)
    rc, oposes, opts, oer, o1, o2 = oracle.local_ba(*a)
    assert ab == 0 and rc == 0
    assert np.array_equal(er, oer) and er.sum() > 1000
    assert _discrete(s1) == _discrete(o1) and _discrete(s2) == _discrete(o2) and s1["iterations"] == 5 and s2["iterations"] == 10
    assert abs(s1["final_cost"] - o1["final_cost"]) <= 1e-8 * o1["final_cost"]
    cloud_cost, cloud_pose = 0.0, 0.0
    for seed in (1, 2):
        ap = a[:4] + (_ulp(g["pts0"], seed),) + a[5:]
        _, pp, _, _, _, p2 = oracle.local_ba(*ap)
        cloud_cost = max(cloud_cost, abs(p2["final_cost"] - o2["final_cost"]) / o2["final_cost"])
        cloud_pose = max(cloud_pose, np.abs(pp - oposes).max())
    d_cost = abs(s2["final_cost"] - o2["final_cost"]) / o2["final_cost"]
    d_pose = np.abs(poses - oposes).max()
    print("C4 %s: GPU-oracle cost %.2e pose %.2e | oracle 1-ulp cloud cost %.2e pose %.2e" % (structure, d_cost, d_pose, cloud_cost, cloud_pose))
    assert d_cost <= max(RTOL_COST, CLOUD * cloud_cost)
    assert d_pose <= max(RTOL_X, CLOUD * cloud_pose)
    # lockstep batch of 33 = 3 distinct maps x 11
    gs = [g] + [synth.make_ba_graph_covis(102 + k, ncam=100, npts=10000, nobs=50000, structure=structure) for k in range(2)]
    probs = [_local_args(gs[k % 3]) for k in range(33)]
    bab, res = optimizer.local_bundle_adjustment_batch(probs)
    plan = optimizer.get_last_plan()
    assert bab == 0 and plan["lookahead_form"] == "k_chol_wg" and plan["problems"] == 33
    for k in (0, 3, 30):                                         # copies of `g`
        bposes, bpts, ber, b1, b2 = res[k]
        assert np.array_equal(bposes, poses) and np.array_equal(bpts, pts) and np.array_equal(ber, er) and b1 == s1 and b2 == s2
    sab, sposes, spts, ser, ss1, ss2 = optimizer.local_bundle_adjustment(*probs[32])
    assert np.array_equal(res[32][0], sposes) and np.array_equal(res[32][1], spts) and np.array_equal(res[32][2], ser) and res[32][3] == ss1 and res[32][4] == ss2


def test_c5_loop_closed_one_iteration_vs_oracle(oracle):
    """GlobalBA on a loop-closed 500-keyframe map (94 block rows; the windows wrap around, so the last ~6 block rows are dense):
    one LM iteration from the start and from the oracle's third iterate, ordinary bars; whichever family factors it."""
    from ceres_mono_orb_slam2_amd import optimizer
    _threads(oracle)
    g = synth.make_ba_graph_covis(3000, ncam=500, npts=50000, nobs=250000, structure="loop")
    base = _ba_args(g)
    p3, x3, _ = oracle.ba_solve(*base, 3)
    _one_iteration(optimizer, oracle, base, [(g["poses0"], g["pts0"]), (p3, x3)])
    plan = optimizer.get_last_plan()
    print("loop-closed C5 plan:", plan)
    assert plan["lookahead_form"] != "none" or plan["two_level_form"] != "none"
    # three iterations: discrete outputs, and a descent
    poses, pts, s = optimizer.bundle_adjustment(*base, 3)
    oposes, opts, os_ = oracle.ba_solve(*base, 3)
    assert _discrete(s) == _discrete(os_)
    cloud = 0.0
    pp, _, ps = oracle.ba_solve(*((base[0], g["poses0"], base[2], _ulp(g["pts0"], 1)) + tuple(base[4:])), 3)
    cloud = abs(ps["final_cost"] - os_["final_cost"]) / os_["final_cost"]
    assert abs(s["final_cost"] - os_["final_cost"]) / os_["final_cost"] <= max(RTOL_COST, CLOUD * cloud)
    assert s["final_cost"] < 0.8 * s["initial_cost"]


_ORDER_SCRIPT = r"""
import sys, json, numpy as np
sys.path.insert(0, sys.argv[1])
from ceres_mono_orb_slam2_amd import optimizer, synth
g = synth.shuffle_keyframes(synth.make_ba_graph(31, ncam=60, npts=3000, nobs=15000, n_fixed=1, max_depth=25.0, min_len=4), 9)
a = (g["K4"], g["poses0"], g["cam_fixed"], np.ones(60, np.uint8), g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
ab, poses, pts, er, s1, s2 = optimizer.local_bundle_adjustment(*a)
print("RESULT " + json.dumps({"poses": poses.tolist(), "pts": pts.tolist(), "erase": er.tolist(), "s1": s1, "s2": s2, "plan": optimizer.get_last_plan()}))
"""


def test_rcm_order_narrows_the_skyline_of_a_shuffled_map(oracle):
    """ORBHIP_BA_ORDER=rcm (round 6; read once per process: two subprocesses): the free keyframes are eliminated in reverse Cuthill - McKee
    order of their covisibility graph instead of the caller's.  On SURVEY 8(d)'s odometry graph with its keyframes SHUFFLED the caller's
    order fills the reduced system's triangle (a band of >= 8 tiles of 12), the reordered one is the band again (<= 4: the one-launch
    backward substitution).  Only the order of elimination changes: both runs reproduce the oracle's two-pass LocalBA - erase flags,
    iteration counts, termination identical, cost 1e-9, poses 1e-7, landmarks 1e-7 of their depth - on this well-conditioned graph."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for v in ("id", "rcm"):
        r = subprocess.run([sys.executable, "-c", _ORDER_SCRIPT, root], env=dict(os.environ, ORBHIP_BA_ORDER=v), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[v] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
    assert res["id"]["plan"]["band_tiles"] >= 8, res["id"]["plan"]
    assert res["rcm"]["plan"]["band_tiles"] <= 4, res["rcm"]["plan"]
    assert res["rcm"]["plan"]["backward_substitution"] == "k_chol_bsolve_sky"
    g = synth.shuffle_keyframes(synth.make_ba_graph(31, ncam=60, npts=3000, nobs=15000, n_fixed=1, max_depth=25.0, min_len=4), 9)
    _threads(oracle)
    oab, oposes, opts, oer, o1, o2 = oracle.local_ba(g["K4"], g["poses0"], g["cam_fixed"], np.ones(60, np.uint8), g["pts0"], g["obs_cam"], g["obs_pt"],
                                                     g["obs_uv"], g["obs_inv_sigma2"])
    for v in ("id", "rcm"):
        d = res[v]
        assert d["erase"] == oer.tolist(), v
        for s, o in ((d["s1"], o1), (d["s2"], o2)):
            assert s["iterations"] == o["iterations"] and s["successful_steps"] == o["successful_steps"] and s["termination"] == o["termination"], (v, s, o)
            assert abs(s["final_cost"] - o["final_cost"]) <= 1e-9 * o["final_cost"], (v, s, o)
        np.testing.assert_allclose(np.array(d["poses"]), oposes, rtol=0, atol=1e-7)
        np.testing.assert_allclose(np.array(d["pts"]), opts, rtol=1e-7, atol=1e-5)        # (landmarks 30 - 60 m away: 1e-7 of their depth)
