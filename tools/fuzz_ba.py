"""Fuzz of ba_solve against the oracle over degenerate graph structures (synth.make_degenerate_ba).  usage: fuzz_ba.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceres_mono_orb_slam2_amd import synth, optimizer
from oracle import pyoracle as po
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for seed in range(N):
    d = synth.make_degenerate_ba(seed)
    try:
        poses, pts, s = optimizer.bundle_adjustment(d["K4"], d["poses0"], d["cam_fixed"], d["pts0"], d["obs_cam"], d["obs_pt"], d["obs_uv"], d["obs_w"], d["obs_robust"], d["iterations"])
    except Exception as e:
        print("GPU raised", seed, d["kind"], e); bad += 1; continue
    ooc, oop, ouv, ow, orb = d["oracle_obs"]
    oposes, opts, os_ = po.ba_solve(d["K4"], d["poses0"], d["cam_fixed"], d["pts0"], ooc, oop, ouv, ow, orb, d["iterations"])
    same = (s["iterations"], s["successful_steps"], s["termination"]) == (os_["iterations"], os_["successful_steps"], os_["termination"])
    dc = abs(s["final_cost"] - os_["final_cost"]) / max(os_["final_cost"], 1e-9 * max(os_["initial_cost"], 1e-30), 1e-300)
    dx = np.abs(poses - oposes).max()
    # rank-deficient graphs amplify the summation order of the reduced system: 1e-8 .. 3e-8 on the cost has been seen
    # (two iterations, final cost 1e-9 of the initial one); discrete outputs must always be identical
    if not same or dc > 5e-8 or dx > 1e-6:
        bad += 1; print("DIFF seed", seed, "kind", d["kind"], s, os_, "dcost", dc, "dx", dx)
print("fuzz_ba: %d problems, %d differences" % (N, bad))
