"""phase stamps of k_chol_bsolve_sky4 (scratch build: bash tools/scratch/exp_build.sh bw_prof -DBW_PROF -DORBHIP_CHOL_PROF -Iinclude, then
ORBHIP_LIB=tools/scratch/lib_bw_prof/liborbslam_hip.so python tools/bw_prof.py <ncam> <npts> <nobs> <iterations>); 10-ns ticks
accumulated by lane 0 of wave 0 (first row) and wave 1 (second row) of problem 0 over every launch of the run"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceres_mono_orb_slam2_amd import optimizer, synth, _lib
L = _lib.load(); L.ba_debug_df_stamps.argtypes = [C.c_void_p, C.c_int]
ncam, npts, nobs, iters = (int(a) for a in sys.argv[1:5])
g = synth.make_ba_graph(1, ncam=ncam, npts=npts, nobs=nobs, n_fixed=1)
a = (g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
optimizer.global_bundle_adjustment(*a, n_iterations=iters)
L.ba_debug_df_stamps(None, 1)
_, _, s = optimizer.global_bundle_adjustment(*a, n_iterations=iters)
st = np.zeros(16, np.uint64); L.ba_debug_df_stamps(st.ctypes.data, 0)
n = s["iterations"]; steps = (6 * (ncam - 1) + 31) // 32
names = ["prologue", "top of step (copies requested / landed)", "x = Linv^T y (wave 0)", "first barrier", "updates (waves 1, 2) / wave 0's copies", "second barrier", "super-block ends", "tail"]
for wv in range(2):
    print("wave", wv, "launches", n, "steps", steps)
    for k in range(8):
        per = st[8 * wv + k] * 10.0 / n
        print("  %-46s %8.2f us per launch   %6.1f ns per step" % (names[k], per / 1e3, per / steps))
print("sum wave 0: %.1f us per launch" % (st[:8].sum() * 10.0 / n / 1e3))
