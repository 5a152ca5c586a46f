"""GlobalBA at C5 size on a LOOP-CLOSED map (500-keyframe chain, 300 far landmarks seen by the first six free and the last six keyframes:
94 block rows, a band of 3 and dense last rows): ms per LM iteration (device, ba_set_profiling).  With ORBHIP_LIB = the experiments
build and ORBHIP_BA_LA_NARROW=0 the same system takes the two-level scheme.  usage: python tools/loop_closed_c5.py [iterations=20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceres_mono_orb_slam2_amd import optimizer, synth
it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ncam, npts = 500, 50000
g = synth.make_ba_graph(1000, ncam=ncam, npts=npts, nobs=250000, n_fixed=1)
rng = np.random.default_rng(5); nf = 300
X = np.stack([rng.uniform(-20, 20, nf), rng.uniform(-4, 4, nf), 0.8 * ncam + rng.uniform(30, 80, nf)], 1)
oc, op, uv = list(g["obs_cam"]), list(g["obs_pt"]), list(g["obs_uv"])
for k in range(nf):
    for c in list(range(1, 7)) + list(range(ncam - 6, ncam)):
        x, z = synth.project(g["K4"][0], g["poses_gt"][c], X[k][None])
        oc.append(c); op.append(npts + k); uv.append(x[0] + rng.normal(0, 1.0, 2))
a = (g["K4"], g["poses0"], g["cam_fixed"], np.vstack([g["pts0"], X * 1.01]), np.array(oc, np.int32), np.array(op, np.int32), np.array(uv),
     np.concatenate([g["obs_inv_sigma2"], np.ones(12 * nf, np.float32)]))
optimizer.global_bundle_adjustment(*a, n_iterations=2)
optimizer.set_profiling(True); optimizer.get_profile()
t0 = time.perf_counter(); poses, pts, s = optimizer.global_bundle_adjustment(*a, n_iterations=it); dt = time.perf_counter() - t0
dev_ms, _, nit = optimizer.get_profile()
print("loop-closed C5: %d LM iterations, %.3f ms per iteration on the device (%.1f ms wall for the solve), final cost %.6f" % (nit, dev_ms / max(nit, 1), dt * 1e3, s["final_cost"]))
