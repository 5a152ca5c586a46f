"""Timeline of the one-launch two-level Cholesky (k_chol_persist_2l) inside a C5-size GlobalBA iteration: builds ba_solver.hip with
-DORBHIP_CHOL_PROF into a scratch library; the kernel's roles stamp s_memrealtime (100 MHz) at fixed points of the LAST
factorisation run; prints per step: chain step length, factor, how late the staging waves' waits were satisfied, when the latest row
published L(.,j) / finished step j, and per outer block when the workers finished its far tiles."""
import os as _os
_os.environ.setdefault("ORBHIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tools", "exp_lib", "liborbslam_hip.so"))   # ORBHIP_BA_PERSIST=2 exists in experiments builds only (bash tools/build_experiments.sh)
import ctypes as C, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
out = os.path.join(ROOT, "gpurun_out", "cholprof"); os.makedirs(out, exist_ok=True)
so = os.path.join(out, "liborbslam_hip_p2.so")
csrc = os.path.join(ROOT, "ceres_mono_orb_slam2_amd", "csrc")
srcs = [os.path.join(csrc, f) for f in ("ba_solver.hip", "capi_common.hip", "orb_extractor.hip", "orb_matcher.hip", "orb_frame.hip", "orb_vocab.hip", "orb_track.hip")]
if not os.path.exists(so) or os.environ.get("REBUILD"):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                           "-DORBHIP_CHOL_PROF", "-I", os.path.join(ROOT, "include"), "-shared", "-o", so] + srcs)
os.environ["ORBHIP_BA_PERSIST"] = "2"                     # the one-launch kernel is opt-in
from ceres_mono_orb_slam2_amd import _lib, optimizer, synth
_lib.LIB_PATH = so
L = _lib.load()
L.ba_debug_p2_prof.argtypes = [C.c_void_p, C.c_int]
ncam = int(sys.argv[1]) if len(sys.argv) > 1 else 500
g = synth.make_ba_graph(1, ncam=ncam, npts=100 * ncam, nobs=500 * ncam, n_fixed=1)
a = (g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
optimizer.global_bundle_adjustment(*a, n_iterations=3)
L.ba_debug_p2_prof(None, 1)
optimizer.global_bundle_adjustment(*a, n_iterations=1)
buf = (C.c_ulonglong * 1024)()
L.ba_debug_p2_prof(buf, 0)
t = np.array(buf, dtype=np.float64).reshape(8, 128)
nb = int(np.count_nonzero(t[0]))
t0 = t[0, 0]
us = lambda x: (x - t0) / 100.0
print("steps", nb, "chain total %.1f us" % us(t[0, nb - 1]))
print(" k  start   step  factor  wait_ok(after start)  L_latest(after start of k)  rows_done  next_row_done")
for k in range(nb):
    step = (t[0, k + 1] - t[0, k]) / 100.0 if k + 1 < nb else 0.0
    f = lambda r: ((t[r, k] - t[0, k]) / 100.0) if t[r, k] > 0 else float("nan")
    print("%3d %7.1f %6.2f %6.2f %8.2f %12.2f %12.2f %12.2f" % (k, us(t[0, k]), step, f(1), f(2), f(5), f(6), f(7)))
print("outer block: far tiles done at (us) / first two tile columns done / chain reached the block's last step at")
ns = 4
for b in range((nb + ns - 1) // ns):
    if t[3, b] > 0: print("%3d %9.1f %9.1f %9.1f" % (b, us(t[3, b]), us(t[4, b]) if t[4, b] > 0 else float("nan"), us(t[0, min((b + 1) * ns - 1, nb - 1)])))
