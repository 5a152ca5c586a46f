"""Phase timing of the chain workgroup of the persistent Cholesky inside real solves - C4-size LocalBA (k_chol_persist, default) or,
with the argument `c5`, a 500-keyframe GlobalBA (since round 5 also k_chol_persist, walking the skyline; before: k_chol_persist_blk, one
launch per 128-column outer block): loads the scratch library built with -DORBHIP_CHOL_PROF and prints the mean time wave 0 of the chain spends in each phase of a step
(100 MHz s_memrealtime ticks -> ns)."""
import ctypes as C, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
so = os.path.join(ROOT, "tools", "scratch", "lib_prof", "liborbslam_hip.so")      # bash tools/scratch/exp_build.sh prof -DORBHIP_CHOL_PROF (as tools/chol_wg_prof.py)
from ceres_mono_orb_slam2_amd import _lib, optimizer, synth
_lib.LIB_PATH = so
L = _lib.load()
L.ba_debug_chol_prof.argtypes = [C.c_void_p, C.c_int]
c5 = "c5" in sys.argv
if c5:
    g = synth.make_ba_graph(1, ncam=500, npts=50000, nobs=250000, n_fixed=1)
    args = (g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
    solve = lambda: optimizer.global_bundle_adjustment(*args, n_iterations=10)
else:
    g = synth.make_ba_graph(0, ncam=100, npts=10000, nobs=50000, n_fixed=1)
    args = (g["K4"], g["poses0"], g["cam_fixed"], np.ones(100, np.uint8), g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
    solve = lambda: optimizer.local_bundle_adjustment(*args)
solve()
L.ba_debug_chol_prof(None, 1)
for _ in range(4): solve()
buf = (C.c_ulonglong * 1280)()
L.ba_debug_chol_prof(buf, 0)
a = np.array(buf, dtype=np.float64).reshape(128, 10)
names = ["factor + inverse (wave 0)", "barrier (staging waves late)", "X stores issued", "L(k+1,k) + publication of X", "D update + publication of L(k+1,k)", "barrier"]
steps = [k for k in range(128) if a[k, 9] > 0]
res = {"steps": len(steps), "visits_per_step": a[steps[0], 9], "ns_per_phase_mean": {}, "per_step_ns": {}}
for i, n in enumerate(names):
    res["ns_per_phase_mean"][n] = float(np.mean([a[k, i] / a[k, 9] for k in steps]) * 10.0)
    res["per_step_ns"][n] = [round(float(a[k, i] / a[k, 9] * 10.0)) for k in steps]
res["mean_step_ns"] = float(sum(res["ns_per_phase_mean"].values()))
print(json.dumps({k: v for k, v in res.items() if k != "per_step_ns"}, indent=1))
out = os.path.join(ROOT, "gpurun_out", "cholprof"); os.makedirs(out, exist_ok=True)
json.dump(res, open(os.path.join(out, "chol_persist_prof_c5.json" if c5 else "chol_persist_prof.json"), "w"), indent=1)
