"""Phase timing of k_ba_schur inside a 64-problem batched LocalBA (tools/build_experiments.sh builds the product sources with
-DORBHIP_SCHUR_PROF into tools/exp_lib/liborbslam_hip_sprof.so): lane 0 of the waves of problem 0 stamps s_memrealtime (100 MHz) at the phase boundaries, per block row, relative to the
workgroup's start.  Prints the mean over rows and launches, and the rows with the longest workgroups."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ceres_mono_orb_slam2_amd import _lib, optimizer, synth
_lib.LIB_PATH = os.path.join(ROOT, "tools", "exp_lib", "liborbslam_hip_sprof.so")
L = _lib.load()
L.ba_debug_chol_prof.argtypes = [C.c_void_p, C.c_int]
gs = [synth.make_ba_graph(s, ncam=100, npts=10000, nobs=50000, n_fixed=1) for s in range(16)]
local = np.ones(100, np.uint8)
probs = [(g["K4"], g["poses0"], g["cam_fixed"], local, g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"]) for g in gs] * 4
optimizer.local_bundle_adjustment_batch(probs)
L.ba_debug_chol_prof(None, 1)
optimizer.local_bundle_adjustment_batch(probs)
buf = (C.c_ulonglong * 1280)()
L.ba_debug_chol_prof(buf, 0)
a = np.array(buf, dtype=np.float64).reshape(128, 10)
rows = [k for k in range(128) if a[k, 9] > 0]
us = a[rows, :9] / a[rows, 9:10] * 0.01          # ticks of 10 ns -> us, per launch
names = ["pass over camera a's list done (wave 0)", "reduction of the 27 sums done", "diagonal block + rhs stored", "-", "wave 0 end", "wave 1 end", "wave 2 end", "wave 3 end"]
for i, n in enumerate(names): print("%-30s mean %6.1f us   max %6.1f" % (n, us[:, i].mean(), us[:, i].max()))
end = us[:, 4:8].max(1)
print("workgroup end: mean %.1f us, max %.1f; launches %d, rows %d" % (end.mean(), end.max(), int(a[rows[0], 9]), len(rows)))
order = np.argsort(-end)[:5]
for k in order: print("row", rows[k], ["%.1f" % v for v in us[k, :8]])
