import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceres_mono_orb_slam2_amd import optimizer, synth
g = synth.make_ba_graph(0, ncam=100, npts=10000, nobs=50000, n_fixed=1)
args = (g["K4"], g["poses0"], g["cam_fixed"], np.ones(100, np.uint8), g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
for _ in range(6): optimizer.local_bundle_adjustment(*args)
