"""Random sweep of the LocalMapping steps against the oracle's composition: python tools/fuzz_localmap.py [first_seed=100] [count=60].
Scenes of tests/test_gpu_localmapping.py with random sizes (few / many points, few nodes -> long candidate lists, many nodes -> empty
intersections, mostly-mapped keyframes), CreateNewMapPoints for 1 .. 20 neighbours and Fuse selection for 1 .. 8 keyframes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
po.lib()
from tests.test_gpu_localmapping import make_scene, oracle_create_new_map_points, SF, LS, BOUNDS
from ceres_mono_orb_slam2_amd import localmapping
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0; t0 = time.time(); tot = 0
for s in range(first, first + count):
    rng = np.random.default_rng(s)
    n_nb = int(rng.integers(1, 21)); npts = int(rng.choice([60, 400, 2600, 3800])); clutter = int(rng.choice([0, 30, 500])); nnodes = int(rng.choice([2, 15, 60, 400]))
    pu = float(rng.choice([0.1, 0.6, 1.0]))
    cur, nbs = make_scene(s, n_nb=n_nb, npts=npts, clutter=clutter, nnodes=nnodes, p_unmapped=pu)
    ratio = np.float32(rng.choice([1.5, 1.8, 2.25]))
    try:
        m, ok, X, npr = localmapping.create_new_map_points(cur, nbs, SF, LS, ratio)
        om, ook, oX = oracle_create_new_map_points(po, cur, nbs, ratio)
        assert npr == n_nb and np.array_equal(m, om) and np.array_equal(ok, ook), "create_new_map_points"
        assert np.abs(X - oX).max() <= 1e-9 * max(1.0, np.abs(oX).max()), "points"
        tot += int(ok.sum())
        T = int(rng.integers(1, min(n_nb, 8) + 1)); Mq = int(rng.choice([1, 70, 900]))
        kfs = [dict(kps=q["kps"], desc=q["desc"], bounds=BOUNDS if t % 3 else np.array([-9.5, 1250.0, -3.0, 380.5], np.float32)) for t, q in enumerate(nbs[:T])]
        uv = np.array([q["kps"][rng.integers(0, len(q["kps"]), Mq), :2] + rng.normal(0, 2.0, (Mq, 2)).astype(np.float32) for q in kfs], np.float32)
        lvl = rng.integers(-1, 8, (T, Mq)).astype(np.int32); rad = (float(rng.choice([3.0, 10.0])) * SF[np.maximum(lvl, 0)]).astype(np.float32)
        mpd = cur["desc"][rng.integers(0, len(cur["desc"]), Mq)]
        ils = (1.0 / LS).astype(np.float32)
        bi, bd = localmapping.fuse_batch(kfs, uv, rad, lvl, mpd, ils)
        for t, q in enumerate(kfs):
            _, om2, obd, _ = po.search_by_projection(q["kps"], q["desc"], q["bounds"], uv[t], rad[t], mpd, q_pred_level=lvl[t], q_valid=(lvl[t] >= 0).astype(np.uint8),
                                                     inv_level_sigma2=ils, chi2_gate=5.99, th=256)
            assert np.array_equal(bi[t], om2) and np.array_equal(bd[t], obd), "fuse_batch keyframe %d" % t
    except AssertionError as e:
        bad += 1; print("seed", s, (n_nb, npts, clutter, nnodes, pu), "FAILED:", str(e)[:200], flush=True)
print("fuzz_localmap: %d seeds from %d, %d failures, %d new points in all, %.0f s" % (count, first, bad, tot, time.time() - t0))
sys.exit(1 if bad else 0)
