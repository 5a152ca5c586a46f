"""Phase timing of the problem-parallel batched Cholesky k_chol_wg inside a batched LocalBA: builds ba_solver.hip with
-DORBHIP_CHOL_PROF into a scratch library (tools/scratch/exp_build.sh prof -DORBHIP_CHOL_PROF must have run), runs 32-problem
lockstep batches and prints the time wave 0 of workgroup 0 spends per phase (s_memrealtime, 100 MHz), per block column."""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
so = os.path.join(ROOT, "tools", "scratch", "lib_prof", "liborbslam_hip.so")
from ceres_mono_orb_slam2_amd import _lib, optimizer, synth
_lib.LIB_PATH = so
L = _lib.load()
L.ba_debug_chol_prof.argtypes = [C.c_void_p, C.c_int]
ST = os.environ.get("ORBHIP_BENCH_STRUCTURE", "band")      # band | covis | dense (synth.make_ba_graph_covis)
gs = [synth.make_ba_graph(s, ncam=100, npts=10000, nobs=50000, n_fixed=1) if ST == "band" else synth.make_ba_graph_covis(3000 + s, ncam=100, npts=10000, nobs=50000, structure=ST) for s in range(2)]
local = np.ones(100, np.uint8)
probs = [(g["K4"], g["poses0"], g["cam_fixed"], local, g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"]) for g in gs] * 16
optimizer.local_bundle_adjustment_batch(probs)
L.ba_debug_chol_prof(None, 1)
N = 2
for _ in range(N): optimizer.local_bundle_adjustment_batch(probs)
buf = (C.c_ulonglong * 1280)()
L.ba_debug_chol_prof(buf, 0)
a = np.array(buf, dtype=np.float64).reshape(128, 10)
names = ["prologue (T, first loads)", "update steps", "layout change + last update", "factor + inverse", "L = T' X^T + stores + barrier"]
cols = [k for k in range(128) if a[k, 9] > 0]
nf = a[0, 9] / float(os.environ.get("CW_GROUPS0", "1"))          # groups of column 0 per factorisation (skyline walk of the bench graph: 1; the dense walk of a 19-block-row system had 3)
tot = a[cols, :5].sum(0) * 10.0 / nf
print("factorisations profiled: %.0f" % nf)
for i, n in enumerate(names): print("%-34s %8.1f us per factorisation" % (n, tot[i] / 1e3))
print("%-34s %8.1f us" % ("sum", tot.sum() / 1e3))
print("per column (us):", [round(float(a[k, :5].sum() * 10.0 / nf / 1e3), 1) for k in cols])
