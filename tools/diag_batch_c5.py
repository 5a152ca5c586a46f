"""Diagnostic: ba_solve_batch vs single ba_solve at C5-like sizes (which batch positions differ)."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceres_mono_orb_slam2_amd import synth, optimizer
def mk(seed, ncam, npts, nobs):
    g = synth.make_ba_graph(seed, ncam=ncam, npts=npts, nobs=nobs, n_fixed=1)
    n = len(g["obs_cam"]); w = np.asarray(g["obs_inv_sigma2"], np.float32).astype(np.float64)
    return (g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], w, np.ones(n, np.uint8))
cfgs = [(500, 5000, 25000, 8), (500, 5000, 25000, 8)]
ref = {}
for ncam, npts, nobs, nb in cfgs:
    probs = [mk(3000 + q, ncam, npts, nobs) for q in range(nb)]
    if os.environ.get("SINGLE_FIRST"):
        for q in range(nb):
            ref[q] = optimizer.bundle_adjustment(*probs[q], 1)
    b = optimizer.bundle_adjustment_batch(probs, n_iterations=1)
    bad = []
    for q in range(nb):
        p, x, s = optimizer.bundle_adjustment(*probs[q], 1)
        if not np.array_equal(p, b[q][0]):
            bad.append((q, s["final_cost"], b[q][2]["final_cost"], s["successful_steps"], b[q][2]["successful_steps"]))
    print(ncam, npts, nobs, "batch", nb, "bad positions", bad, flush=True)
