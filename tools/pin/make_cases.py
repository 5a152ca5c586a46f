"""Writes the inputs of the pinning harness (tools/pin/CMakeLists.txt): synthetic frames as binary PGM and pose / bundle-adjustment
problems as flat little-endian files (layouts: pin_solver.cpp).  usage: python tools/pin/make_cases.py <out dir>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ceres_mono_orb_slam2_amd import synth

out = sys.argv[1] if len(sys.argv) > 1 else "pin_cases"
os.makedirs(out, exist_ok=True)
k = 0
for (w, h, fam) in [(1241, 376, "blocks"), (1241, 376, "checker"), (1241, 376, "flat"), (640, 480, "blocks"), (640, 480, "checker"), (752, 480, "blocks")]:
    for seed in (0, 1):
        img = synth.make_frame(seed, w, h, fam)
        with open(os.path.join(out, "frame_%03d.pgm" % k), "wb") as f:
            f.write(b"P5\n%d %d\n255\n" % (w, h)); f.write(np.ascontiguousarray(img, np.uint8).tobytes())
        k += 1
for i, n in enumerate((2000, 500, 50, 7)):
    p = synth.make_pose_problem(i, n=n)
    with open(os.path.join(out, "pose_%03d.bin" % i), "wb") as f:
        f.write(np.int32(n).tobytes())
        for a, dt in ((p["K4"], np.float64), (p["pose0"], np.float64), (p["Xw"], np.float64), (p["uv"], np.float64), (p["inv_sigma2"], np.float32)):
            f.write(np.ascontiguousarray(a, dt).tobytes())
for i, (ncam, npts, nobs, it) in enumerate(((6, 120, 500, 20), (12, 400, 2000, 30), (30, 2000, 9000, 10), (100, 10000, 50000, 5))):
    g = synth.make_ba_graph(i, ncam=ncam, npts=npts, nobs=nobs, n_fixed=1)
    with open(os.path.join(out, "ba_%03d.bin" % i), "wb") as f:
        f.write(np.array([ncam, npts, len(g["obs_cam"]), it], np.int32).tobytes())
        for a, dt in ((g["K4"], np.float64), (g["poses0"], np.float64), (g["cam_fixed"], np.uint8), (g["pts0"], np.float64), (g["obs_cam"], np.int32), (g["obs_pt"], np.int32),
                      (g["obs_uv"], np.float64), (g["obs_inv_sigma2"], np.float32)):
            f.write(np.ascontiguousarray(a, dt).tobytes())
print("wrote %d frames, 4 pose problems, 4 bundle-adjustment problems to %s" % (k, out))
