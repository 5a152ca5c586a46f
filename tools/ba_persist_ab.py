"""Device time of a single C4-size LocalBA and a C4 GlobalBA with the persistent Cholesky (default) or the per-step launches
(ORBHIP_BA_PERSIST=0): run once per setting (the switch is read once per process)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceres_mono_orb_slam2_amd import synth, optimizer
g = synth.make_ba_graph(0, ncam=100, npts=10000, nobs=50000, n_fixed=1)
local = np.ones(100, np.uint8)
args = (g["K4"], g["poses0"], g["cam_fixed"], local, g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
for _ in range(3): r = optimizer.local_bundle_adjustment(*args)
optimizer.set_profiling(True); optimizer.get_profile()
N = 20
t0 = time.perf_counter()
for _ in range(N): r = optimizer.local_bundle_adjustment(*args)
wall = (time.perf_counter() - t0) / N * 1e3
ms, ns, it = optimizer.get_profile()
print(json.dumps({"persist": os.environ.get("ORBHIP_BA_PERSIST", "1"), "localba_wall_ms": round(wall, 3), "localba_device_ms": round(ms / N, 3), "lm_iterations": it // N,
                  "device_us_per_lm_iteration": round(ms * 1e3 / it, 1), "final_cost": r[5]["final_cost"]}))
