"""GlobalBA on the two-level Cholesky: the one-launch persistent kernel (default, ORBHIP_BA_PERSIST=2), persistent block launches
(=1) and one launch per 32-column step (=0): results must be bit-identical; prints the time per call of each.  usage: gba_persist_ab.py [ncam npts nobs iters]"""
import os as _os
_os.environ.setdefault("ORBHIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tools", "exp_lib", "liborbslam_hip.so"))   # ORBHIP_BA_PERSIST=2 exists in experiments builds only (bash tools/build_experiments.sh)
import hashlib, os, subprocess, sys

_CHILD = r'''
import sys, time, hashlib, numpy as np
sys.path.insert(0, %r)
from ceres_mono_orb_slam2_amd import synth, optimizer
ncam, npts, nobs, iters = %d, %d, %d, %d
g = synth.make_ba_graph(1, ncam=ncam, npts=npts, nobs=nobs, n_fixed=1)
a = (g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    poses, pts, s = optimizer.global_bundle_adjustment(*a, n_iterations=iters)
    best = min(best, time.perf_counter() - t0)
h = hashlib.sha256(np.ascontiguousarray(poses).tobytes() + np.ascontiguousarray(pts).tobytes()).hexdigest()
print("RESULT", h, "%%.3f" %% (best * 1e3), s["iterations"], "%%.9e" %% s["final_cost"])
'''

def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfgs = [tuple(int(x) for x in sys.argv[1:5])] if len(sys.argv) >= 5 else [(500, 50000, 250000, 10), (190, 12000, 60000, 6), (173, 9000, 50000, 5)]
    bad = 0
    for c in cfgs:
        out = {}
        for mode in ("2", "1", "0"):
            env = dict(os.environ, ORBHIP_BA_PERSIST=mode)
            r = subprocess.run([sys.executable, "-c", _CHILD % ((root,) + c)], env=env, capture_output=True, text=True, timeout=900)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
            if not line:
                print(c, "mode", mode, "FAILED", r.stdout[-500:], r.stderr[-1500:]); bad += 1; continue
            out[mode] = line[0].split()
        if len(out) == 3:
            same = out["1"][1] == out["0"][1] == out["2"][1]
            bad += not same
            print(c, "one launch %s ms, block launches %s ms, steps %s ms, iterations %s cost %s / %s / %s, identical: %s" %
                  (out["2"][2], out["1"][2], out["0"][2], out["1"][3], out["2"][4], out["1"][4], out["0"][4], same))
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
