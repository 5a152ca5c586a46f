import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceres_mono_orb_slam2_amd import synth, optimizer
from oracle import pyoracle as po
g = synth.make_ba_graph(3, ncam=25, npts=1500, nobs=7000, n_fixed=2)
n=len(g['obs_cam']); w=g['obs_inv_sigma2'].astype(np.float64); rb=np.ones(n,np.uint8)
poses, pts, s = optimizer.bundle_adjustment(g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], w, rb, 15)
oposes, opts, os_ = po.ba_solve(g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], w, rb, 15)
print(s); print(os_)
d = np.abs(pts-opts); rel = d/np.maximum(1,np.abs(opts))
i = np.unravel_index(rel.argmax(), rel.shape); print('max abs',d.max(),'max rel',rel.max(), pts[i[0]], opts[i[0]], np.abs(opts).max())
print('poses', np.abs(poses-oposes).max())
print(np.sort(rel.max(1))[-10:])
