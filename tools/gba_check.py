import numpy as np, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceres_mono_orb_slam2_amd import synth, optimizer
from oracle import pyoracle as po
ncam, npts, nobs, iters, nfix = 252, 12000, 60000, 6, int(sys.argv[1]) if len(sys.argv) > 1 else 2
g = synth.make_ba_graph(3, ncam=ncam, npts=npts, nobs=nobs, n_fixed=nfix)
n = len(g["obs_cam"]); w = g["obs_inv_sigma2"].astype(np.float64); rb = np.ones(n, np.uint8)
poses, pts, s = optimizer.bundle_adjustment(g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], w, rb, iters)
t0 = time.time()
oposes, opts, os_ = po.ba_solve(g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], w, rb, iters)
print('oracle s', time.time() - t0)
print(s); print(os_)
print('pose diff', np.abs(poses - oposes).max(), 'cost rel', abs(s['final_cost'] - os_['final_cost']) / os_['final_cost'])
d = np.linalg.norm(pts - opts, axis=1) / np.maximum(1, np.linalg.norm(opts, axis=1)); print('pt rel max', d.max(), 'median', np.median(d))
