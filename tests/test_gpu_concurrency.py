"""The reference's threading on ONE GPU (VERDICT r3 next #3a): Tracking runs per frame on its thread while LocalMapping runs
CeresOptimizer::LocalBundleAdjustment on another (src/LocalMapping.cc:89) and the loop closer's GlobalBundleAdjustemnt on a detached
third (src/LoopClosing.cc:590,656).  Here: three host threads share the device - the device-resident Tracking step (motion model +
local map) frame after frame on a high-priority stream, single LocalBA solves (C4 size: the persistent, flag-linked Cholesky) in
a loop, one GlobalBA at C5 size (persistent block launches).  Every result must be BIT-IDENTICAL to the same call made alone, no
call may time out, and the Tracking latency under that load is reported (p50 / p99) next to the solo latency."""
import os
import threading
import time

import numpy as np
import pytest

from ceres_mono_orb_slam2_amd import synth
from tests.test_gpu_track import _scenario, K4, BOUNDS, F32
from tests.test_gpu_track_local_map import _local_map

pytestmark = pytest.mark.gpu


def _digest(*arrs):
    import hashlib
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def test_tracking_localba_globalba_share_one_gpu(oracle):
    from ceres_mono_orb_slam2_amd import ORBextractor, tracking, optimizer, _lib
    L = _lib.load()
    S = _scenario(oracle, 21)
    ex = ORBextractor(2000, 1.2, 8, 20, 7)
    a1 = (ex, S["img"], K4, BOUNDS, S["T"], S["X"], S["desc"], S["octave"], S["angle"], S["valid"], 15.0, True)
    g1 = tracking.track_with_motion_model(*a1)
    T1 = oracle.pose7_to_matrix4d(g1["pose7"])
    M = _local_map(oracle, S, g1, 21)
    a2 = (ex, K4, BOUNDS, T1, F32(np.log(F32(1.2))), M["X"], M["Pn"], M["mind"], M["maxd"], M["D"], M["state"], M["slot_X"], M["slot_state"], 1.0, 0.8)

    lib_ms = [0.0]

    def track_once():
        r1 = tracking.track_with_motion_model(*a1); m1 = tracking.last_call_ms()
        r2 = tracking.track_local_map(*a2); lib_ms[0] = m1 + tracking.last_call_ms()       # (time spent inside the two C calls)
        return _digest(r1["kps"], r1["desc"], r1["match"], r1["owner"], r1["outlier"], r1["pose7"], r2["in_view"], r2["match"], r2["owner"], r2["outlier"], r2["pose7"])

    gl = synth.make_ba_graph(3, ncam=100, npts=10000, nobs=50000, n_fixed=1)
    la = (gl["K4"], gl["poses0"], gl["cam_fixed"], np.ones(100, np.uint8), gl["pts0"], gl["obs_cam"], gl["obs_pt"], gl["obs_uv"], gl["obs_inv_sigma2"])

    def lba_once():
        ab, poses, pts, er, s1, s2 = optimizer.local_bundle_adjustment(*la)
        return _digest(poses, pts, er) + str((s1["iterations"], s1["termination"], s2["iterations"], s2["termination"], s2["final_cost"]))

    gg = synth.make_ba_graph(21, ncam=500, npts=50000, nobs=250000, n_fixed=2)
    ga = (gg["K4"], gg["poses0"], gg["cam_fixed"], gg["pts0"], gg["obs_cam"], gg["obs_pt"], gg["obs_uv"], gg["obs_inv_sigma2"])

    def gba_once():
        poses, pts, s = optimizer.global_bundle_adjustment(*ga, n_iterations=6)
        return _digest(poses, pts) + str((s["iterations"], s["termination"], s["final_cost"]))

    # solo references (and solo Tracking latency)
    ref_t, ref_l, ref_g = track_once(), lba_once(), gba_once()
    solo, solo_lib = [], []
    for _ in range(100):
        t0 = time.perf_counter(); d = track_once(); solo.append(time.perf_counter() - t0); solo_lib.append(lib_ms[0] * 1e-3)
        assert d == ref_t, "solo Tracking step %d differs from the first one" % len(solo)
    # the three threads
    out = {"t": [], "l": [], "g": [], "lat": [], "lib": [], "llat": [], "err": []}
    stop = threading.Event()
    ready = threading.Barrier(3)          # the latency figures are steady-state: every thread has made its first call (allocations) before the clock runs

    def tracker():
        try:
            _lib.check(L.orbhip_set_thread_priority(1), "orbhip_set_thread_priority")
            track_once()                                     # (this thread's own workspace and resident frame)
            ready.wait(timeout=120)
            while not stop.is_set():
                t0 = time.perf_counter(); d = track_once(); out["lat"].append(time.perf_counter() - t0); out["lib"].append(lib_ms[0] * 1e-3); out["t"].append(d)
        except Exception as e:
            out["err"].append(repr(e))

    def local_mapper():
        try:
            out["l"].append(lba_once()); ready.wait(timeout=120)  # (first call of this thread: workspace allocation - hipMalloc / hipHostMalloc hold runtime locks for milliseconds)
            while not stop.is_set():
                t0 = time.perf_counter(); out["l"].append(lba_once()); out["llat"].append(time.perf_counter() - t0)
        except Exception as e:
            out["err"].append(repr(e))

    def loop_closer():
        try:
            out["g"].append(gba_once()); ready.wait(timeout=120)
            for _ in range(3):
                out["g"].append(gba_once())
        except Exception as e:
            out["err"].append(repr(e))
        finally:
            stop.set()

    # Latency is taken INSIDE the library (orbt_last_call_ms): these are Python threads, and what the interpreter lock adds around a
    # call is not the library's.  Round 4 found the 7-10 ms p99 of this test inside hipMemcpyAsync (rocprofv3 --hip-trace: two threads'
    # asynchronous copies of a few hundred kilobytes each blocked on the host for 8 ms, the GPU idle); the per-frame calls now move their
    # staging blocks with a copy kernel (csrc/common.h) and the p99 is that of the kernels.
    th = [threading.Thread(target=f) for f in (tracker, local_mapper, loop_closer)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not out["err"], out["err"]
    assert len(out["t"]) >= 5 and len(out["l"]) >= 1 and len(out["g"]) == 4, (len(out["t"]), len(out["l"]), len(out["g"]))
    assert all(d == ref_t for d in out["t"]), "%d of %d Tracking steps differ from the solo run" % (sum(d != ref_t for d in out["t"]), len(out["t"]))
    assert all(d == ref_l for d in out["l"]), "%d of %d LocalBA solves differ from the solo run" % (sum(d != ref_l for d in out["l"]), len(out["l"]))
    assert all(d == ref_g for d in out["g"]), "%d of %d GlobalBA solves differ from the solo run" % (sum(d != ref_g for d in out["g"]), len(out["g"]))
    q = lambda v, p: float(np.percentile(np.array(v) * 1e3, p))
    if out["llat"]:
        print("LocalBA beside Tracking + GlobalBA (wall, Python): median %.2f ms, max %.2f ms over %d solves" % (q(out["llat"], 50), q(out["llat"], 100), len(out["llat"])))
    # The latency figure is REPORTED, not asserted, by default (a wall-clock bound on shared or leased hardware is flaky
    # whatever its margin - it depends on the runtime's queue mapping and on what else the box runs); the bit-identity assertions above
    # are the hard ones.  ORBHIP_TEST_LATENCY_BOUND_MS=4 turns the bound into an assertion (0.6 - 1.3 ms measured; the copy-engine path
    # it guards against showed 7 - 10 ms).
    bound = float(os.environ.get("ORBHIP_TEST_LATENCY_BOUND_MS", "0") or 0)
    p99 = q(out["lib"], 99)
    if bound > 0:
        assert p99 < bound, "Tracking p99 inside the library beside LocalBA + GlobalBA: %.2f ms" % p99
    elif p99 >= 4.0:
        import warnings
        warnings.warn("Tracking p99 inside the library beside LocalBA + GlobalBA: %.2f ms (>= 4 ms; reported, not asserted)" % p99)
    print("Tracking (motion model + local map) alone: p50 %.3f ms, p99 %.3f ms (inside the library: %.3f / %.3f); beside LocalBA + GlobalBA on the same GPU: "
          "p50 %.3f ms, p99 %.3f ms seen from Python, p50 %.3f ms, p99 %.3f ms, max %.3f ms inside the library (orbt_last_call_ms), over %d frames "
          "(%d LocalBA, %d GlobalBA solves meanwhile, all bit-identical to their solo runs)" % (q(solo, 50), q(solo, 99), q(solo_lib, 50), q(solo_lib, 99),
                                                                                             q(out["lat"], 50), q(out["lat"], 99), q(out["lib"], 50), q(out["lib"], 99), q(out["lib"], 100),
                                                                                             len(out["lat"]), len(out["l"]), len(out["g"])))
