"""Single LocalBA (C4 size) latency: wall per solve and device time per solve (ba_set_profiling), 12 solves after warm-up."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceres_mono_orb_slam2_amd import optimizer, synth
g = synth.make_ba_graph(0, ncam=100, npts=10000, nobs=50000, n_fixed=1)
args = (g["K4"], g["poses0"], g["cam_fixed"], np.ones(100, np.uint8), g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
for _ in range(3): optimizer.local_bundle_adjustment(*args)
optimizer.set_profiling(True); optimizer.get_profile()
n = 12
t0 = time.perf_counter()
for _ in range(n): r = optimizer.local_bundle_adjustment(*args)
dt = time.perf_counter() - t0
dev_ms, ns, nit = optimizer.get_profile()
print("LocalBA single: wall %.3f ms per solve, device %.3f ms per solve, %.1f us per LM iteration (%d iterations per solve), final cost %.6f" %
      (dt / n * 1e3, dev_ms / n, dev_ms / max(nit, 1) * 1e3, nit // n, r[5]["final_cost"]))
