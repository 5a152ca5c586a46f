import numpy as np, sys, time, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceres_mono_orb_slam2_amd import synth, optimizer
g = synth.make_ba_graph(0, ncam=100, npts=10000, nobs=50000, n_fixed=2)
local = np.ones(100, np.uint8)
args = (g["K4"], g["poses0"], g["cam_fixed"], local, g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
def work(tag):
    for i in range(4):
        t0=time.perf_counter(); optimizer.local_bundle_adjustment(*args); print(tag, 'ms', (time.perf_counter()-t0)*1e3, flush=True)
work('main')
t = threading.Thread(target=work, args=('worker',)); t.start(); t.join()
