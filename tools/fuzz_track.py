"""Fuzz of the device-resident Tracking step against the oracle's composition (tests/test_gpu_track.py's scenario generator over
random seeds, thresholds, fractions of points without observations / dropped points, pose errors).  usage: fuzz_track.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
from tests import test_gpu_track as T
from ceres_mono_orb_slam2_amd import ORBextractor, tracking
po.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ex = ORBextractor(2000, 1.2, 8, 20, 7)
bad = 0
for seed in range(100, 100 + N):
    rng = np.random.default_rng(seed)
    kw = dict(no_obs_frac=float(rng.choice([0.0, 0.1, 0.5, 0.9])), drop_frac=float(rng.choice([0.0, 0.1, 0.6])), depth=float(rng.uniform(6, 40)),
              pose_noise=float(rng.choice([1e-3, 5e-3, 2e-2])))
    th = float(rng.choice([5.0, 15.0, 30.0, 60.0]))
    S = T._scenario(po, seed, **kw)
    got = tracking.track_with_motion_model(ex, S["img"], T.K4, T.BOUNDS, S["T"], S["X"], S["desc"], S["octave"], S["angle"], S["valid"], th, True)
    exp = T._expected(po, S, th)
    ok = (np.array_equal(got["kps"], exp["kps"]) and np.array_equal(got["desc"], exp["desc"]) and got["nmatches"] == exp["nmatches"] and
          np.array_equal(got["match"], exp["match"]) and np.array_equal(got["owner"], exp["owner"]) and got["n_inliers"] == exp["n_inliers"] and
          np.array_equal(got["outlier"], exp["outlier"]) and np.abs(got["pose7"] - exp["pose7"]).max() < 1e-7)
    if not ok:
        bad += 1
        print("DIFF seed", seed, kw, th, got["nmatches"], exp["nmatches"], got["n_inliers"], exp["n_inliers"], np.abs(got["pose7"] - exp["pose7"]).max())
print("fuzz_track: %d scenarios, %d differences" % (N, bad))
