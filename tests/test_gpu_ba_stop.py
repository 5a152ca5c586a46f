"""B6: StopFlagCallback (reference include/CeresOptimizer.h:332-349, src/CeresOptimizer.cc:509-514) raised MID-SOLVE.

The reference polls the flag after every LM iteration; when it is set Ceres terminates with USER_SUCCESS and the last accepted
iterate is what gets written back.  The HIP path enqueues all iterations ahead of time, so the flag is forwarded to a pinned
byte that k_ba_iter_begin reads before every iteration.  Checks: termination == 4 (user stop), fewer iterations than the cap,
and the returned state is EXACTLY the state an uninterrupted solve capped at that iteration count returns (bit-identical:
same kernels, same order) -- i.e. the accepted iterate was written back, not a half-applied candidate."""
import threading
import time

import numpy as np
import pytest

from ceres_mono_orb_slam2_amd import synth

pytestmark = pytest.mark.gpu


def _raise_after(flag, seconds):
    """A thread that raises the flag `seconds` after the caller calls go.set() (right before the solve).  It is started and
    parked on the event beforehand and then spins on the clock: time.sleep() plus thread start-up jitter (a millisecond or
    more inside a long pytest session) is as long as a whole LocalBA (7 ms) - the first version of this helper made the
    test miss the solve altogether once in the full suite."""
    go = threading.Event()

    def run():
        go.wait()
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end:
            pass
        flag[0] = 1
    t = threading.Thread(target=run)
    t.start()
    return t, go


def test_globalba_flag_raised_mid_solve_stops_at_an_iteration_boundary():
    from ceres_mono_orb_slam2_amd import optimizer
    g = synth.make_ba_graph(21, ncam=500, npts=50000, nobs=250000, n_fixed=2)
    a = (g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
    optimizer.global_bundle_adjustment(*a, n_iterations=5)                       # warm-up: allocations, the iteration graph (captured from 3 iterations on)
    t_full = 1e9
    for _ in range(2):                                                            # (the shorter of two: a slow first call would push the raise past the solve)
        t0 = time.perf_counter()
        _, _, full = optimizer.global_bundle_adjustment(*a, n_iterations=50)
        t_full = min(t_full, time.perf_counter() - t0)
    assert full["iterations"] >= 20, full
    t_one = 1e9                                                                   # what a call spends before its second iteration: validation, structure pass, uploads
    for _ in range(2):
        t0 = time.perf_counter()
        optimizer.global_bundle_adjustment(*a, n_iterations=1)
        t_one = min(t_one, time.perf_counter() - t0)
    t_one = min(t_one, 0.8 * t_full)
    for frac in (0.3, 0.5, 0.2, 0.7, 0.4, 0.6):                                   # the raise is timed INSIDE the iterations' share of the call: further tries at other points of it
        flag = np.zeros(1, np.uint8)
        th, go = _raise_after(flag, t_one + frac * (t_full - t_one))
        go.set()
        poses, pts, s = optimizer.global_bundle_adjustment(*a, n_iterations=50, stop_flag=flag)
        th.join()
        if s["termination"] == 4 and 1 <= s["iterations"] < full["iterations"]:
            break
    assert s["termination"] == 4, s                                               # SOLVER_TERMINATE_SUCCESSFULLY
    assert 1 <= s["iterations"] < full["iterations"], (s, full)
    # the state is the accepted iterate after exactly s["iterations"] iterations
    rposes, rpts, r = optimizer.global_bundle_adjustment(*a, n_iterations=s["iterations"])
    assert r["iterations"] == s["iterations"] and r["successful_steps"] == s["successful_steps"]
    assert r["final_cost"] == s["final_cost"]
    assert np.array_equal(rposes, poses) and np.array_equal(rpts, pts)


def test_localba_flag_raised_mid_solve():
    """Raised during pass 1 or pass 2 of LocalBundleAdjustment: the running solve ends at its next iteration boundary; if
    that was pass 1 the reference returns before pass 2 without writing anything back (src/CeresOptimizer.cc:509-512 after
    `goto reoptimize`), if it was pass 2 the accepted iterate is written back."""
    from ceres_mono_orb_slam2_amd import optimizer
    g = synth.make_ba_graph(22, ncam=100, npts=10000, nobs=50000, n_fixed=1)
    a = (g["K4"], g["poses0"], g["cam_fixed"], np.ones(100, np.uint8), g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
    optimizer.local_bundle_adjustment(*a)
    t_full = 1e9
    for _ in range(3):                                                            # (the shortest of three: a slow first call would push every raise past the solve)
        t0 = time.perf_counter()
        ab0, poses0, pts0, er0, f1, f2 = optimizer.local_bundle_adjustment(*a)
        t_full = min(t_full, time.perf_counter() - t0)
    assert ab0 == 0 and f1["iterations"] == 5 and f2["iterations"] >= 5
    seen = set()
    for rnd in range(4):                                                          # the raise is timed, the solve takes ~7 ms: several tries
        for frac in (0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8):
            flag = np.zeros(1, np.uint8)
            th, go = _raise_after(flag, frac * t_full)
            go.set()
            ab, poses, pts, er, s1, s2 = optimizer.local_bundle_adjustment(*a, stop_flag=flag)
            th.join()
            if ab:                                                                # stopped in pass 1 (or between the passes)
                assert np.array_equal(poses, g["poses0"]) and np.array_equal(pts, g["pts0"])    # nothing written back
                seen.add("pass1")
            elif s2["termination"] == 4:
                assert s2["iterations"] < f2["iterations"] and s1["iterations"] == 5
                assert np.isfinite(poses).all() and s2["final_cost"] <= s2["initial_cost"]
                seen.add("pass2")
            else:
                seen.add("late")                                                  # the flag came after the last iteration
        if seen & {"pass1", "pass2"}:
            break
    assert seen & {"pass1", "pass2"}, seen


def test_flag_set_before_the_call():
    from ceres_mono_orb_slam2_amd import optimizer
    g = synth.make_ba_graph(23, ncam=20, npts=800, nobs=4000, n_fixed=1)
    flag = np.ones(1, np.uint8)
    poses, pts, s = optimizer.global_bundle_adjustment(g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"],
                                                       g["obs_inv_sigma2"], n_iterations=20, stop_flag=flag)
    assert s["termination"] == 4 and s["iterations"] == 0                        # callback after iteration 0
    assert np.array_equal(pts, g["pts0"])


def test_wait_timeout_is_its_own_outcome_not_a_rejected_step():
    """VERDICT r3 weak #6 / ADVICE r3: a workgroup of the persistent Cholesky that waits longer than its time limit for another
    one used to raise chol_fail - the LM step was rejected, the radius shrank and the caller got a different, valid-looking
    solution.  Now it is its own outcome: ba_summary.termination 7, the entry point returns ORBHIP_ETIMEOUT (-7), the LM radius is
    untouched.  ba_set_wait_limit_ms(1e-5) (one 10-ns tick) makes every wait that is not satisfied at once run out (the rows near the
    bottom of a 600-unknown system wait ~100 us for their first panel), then restores the limit: the next solve must be
    bit-identical to the one before the forced timeout."""
    import ctypes as C
    from ceres_mono_orb_slam2_amd import optimizer, _lib
    L = _lib.load()
    g = synth.make_ba_graph(77, ncam=101, npts=3000, nobs=15000, n_fixed=1)           # 600 unknowns: 19 block rows, k_chol_persist
    a = (g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"].astype(np.float64),
         np.ones(len(g["obs_cam"]), np.uint8), 6)
    poses0, pts0, s0 = optimizer.bundle_adjustment(*a)
    assert s0["termination"] != 7 and s0["iterations"] >= 2
    _lib.check(L.ba_set_wait_limit_ms(1e-5), "ba_set_wait_limit_ms")
    try:
        K4 = np.ascontiguousarray(g["K4"], np.float64).reshape(-1, 4); poses = np.ascontiguousarray(g["poses0"], np.float64).copy()
        pts = np.ascontiguousarray(g["pts0"], np.float64).reshape(-1, 3).copy(); cf = np.ascontiguousarray(g["cam_fixed"], np.uint8)
        oc = np.ascontiguousarray(g["obs_cam"], np.int32); op = np.ascontiguousarray(g["obs_pt"], np.int32)
        uv = np.ascontiguousarray(g["obs_uv"], np.float64).reshape(-1, 2); w = a[7]; rb = a[8]
        o = _lib.BaOptions(6, float(optimizer.HUBER_DELTA), 0, None)
        s = _lib.BaSummary()
        rc = L.ba_solve(_lib.ptr(K4), _lib.ptr(poses), _lib.ptr(cf), len(cf), _lib.ptr(pts), len(pts), _lib.ptr(oc), _lib.ptr(op), _lib.ptr(uv),
                        _lib.ptr(w), _lib.ptr(rb), len(oc), C.byref(o), C.byref(s))
        d = s.as_dict()
        assert rc == -7, (rc, d)
        assert d["termination"] == 7 and d["successful_steps"] == 0 and d["iterations"] == 1
        assert d["final_radius"] == 1e4                                              # initial_trust_region_radius: not shrunk
        assert d["final_cost"] == d["initial_cost"] == s0["initial_cost"]
        assert np.array_equal(poses, np.ascontiguousarray(g["poses0"], np.float64))  # the last accepted iterate = the start
        assert b"time limit" in L.orbhip_last_error()
        with pytest.raises(_lib.OrbHipError):
            optimizer.bundle_adjustment(*a)
    finally:
        _lib.check(L.ba_set_wait_limit_ms(0.0), "ba_set_wait_limit_ms")
    poses1, pts1, s1 = optimizer.bundle_adjustment(*a)
    assert s1 == s0 and np.array_equal(poses1, poses0) and np.array_equal(pts1, pts0)
