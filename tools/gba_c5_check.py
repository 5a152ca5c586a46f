"""GlobalBA at C5 size (500 KF x 50000 pts x 250000 obs): GPU (scheme per environment) vs the CPU oracle after N iterations."""
import numpy as np, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceres_mono_orb_slam2_amd import synth, optimizer
from oracle import pyoracle
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
g = synth.make_ba_graph(1, ncam=500, npts=50000, nobs=250000, n_fixed=1)
a = (g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
poses, pts, s = optimizer.global_bundle_adjustment(*a, n_iterations=iters)
print("gpu", s)
if len(sys.argv) > 2:
    t0 = time.perf_counter()
    n = len(g["obs_cam"])
    oposes, opts, os_ = pyoracle.ba_solve(g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"],
                                          np.asarray(g["obs_inv_sigma2"], np.float32).astype(np.float64), np.ones(n, np.uint8), iters)
    oposes = oposes.copy(); oposes[:, 3:] /= np.linalg.norm(oposes[:, 3:], axis=1, keepdims=True)
    print("oracle %.1f s" % (time.perf_counter() - t0), os_)
    print("rel cost diff %.3e  pose diff %.3e" % (abs(s["final_cost"] - os_["final_cost"]) / os_["final_cost"], np.abs(poses - oposes).max()))
