"""Bit-identity of the landmark blocks summed inside k_ba_eval<0> against the separate pass (ORBHIP_BA_PT_FUSE=0, experiments build):
the same solves in two subprocesses, SHA-256 of every output.  usage: python tools/pt_fuse_ab.py   (needs tools/exp_lib/liborbslam_hip.so)"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, hashlib, numpy as np
sys.path.insert(0, sys.argv[1])
from ceres_mono_orb_slam2_amd import optimizer, synth
h = hashlib.sha256()
shapes = [(10, 300, 1400, 2), (23, 500, 2600, 1), (6, 120, 500, 2), (48, 900, 4500, 1), (100, 10000, 50000, 1), (3, 60, 200, 1), (40, 64, 2200, 1)]
for k, (c, p, o, f) in enumerate(shapes):
    g = synth.make_ba_graph(70 + k, ncam=c, npts=p, nobs=o, n_fixed=f)
    poses, pts, s = optimizer.bundle_adjustment(g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"].astype(np.float64),
                                                np.ones(len(g["obs_cam"]), np.uint8), 8)
    h.update(poses.tobytes()); h.update(pts.tobytes()); h.update(repr(sorted(s.items())).encode())
    ab, poses, pts, er, s1, s2 = optimizer.local_bundle_adjustment(g["K4"], g["poses0"], g["cam_fixed"], np.ones(c, np.uint8), g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
    h.update(poses.tobytes()); h.update(pts.tobytes()); h.update(er.tobytes()); h.update(repr(sorted(s2.items())).encode())
print("SHA", h.hexdigest())
'''
out = []
for flag in ("1", "0"):
    env = dict(os.environ, ORBHIP_LIB=os.path.join(ROOT, "tools", "exp_lib", "liborbslam_hip.so"), ORBHIP_BA_PT_FUSE=flag)
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True, timeout=900)
    if r.returncode: sys.exit(r.stderr[-2000:])
    out.append([l for l in r.stdout.splitlines() if l.startswith("SHA")][-1])
    print("ORBHIP_BA_PT_FUSE=%s %s" % (flag, out[-1]))
print("bit-identical" if out[0] == out[1] else "DIFFERENT")
sys.exit(0 if out[0] == out[1] else 1)
