import numpy as np, sys
sys.path.insert(0, "/root/repo")
from ceres_mono_orb_slam2_amd import synth, optimizer
from oracle import pyoracle as oracle
g = synth.make_ba_graph(11, ncam=121, npts=3000, nobs=15000, n_fixed=2)
n = len(g["obs_cam"]); w = g["obs_inv_sigma2"].astype(np.float64); rb = np.ones(n, np.uint8)
for it in (2, 4, 8, 12, 20):
    poses, pts, s = optimizer.bundle_adjustment(g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], w, rb, it)
    oposes, opts, os_ = oracle.ba_solve(g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], w, rb, it)
    print(it, s["iterations"], os_["iterations"], "rel cost diff %.3e" % (abs(s["final_cost"] - os_["final_cost"]) / os_["final_cost"]), "pose diff %.3e" % np.abs(poses - oposes).max(), "cost", s["final_cost"])
