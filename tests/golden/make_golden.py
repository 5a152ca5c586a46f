#!/usr/bin/env python3
"""Generate the committed golden fixtures from the CPU oracle (run from the repo root).

The reference ships no golden vectors and cannot be run (OpenCV/Ceres are
absent), so these are ORACLE outputs on seeded synthetic inputs: they pin the oracle against
accidental change and give the GPU tests data-only fixtures that travel to the GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po            # noqa: E402
from ceres_mono_orb_slam2_amd import synth   # noqa: E402

out = os.path.dirname(os.path.abspath(__file__))
nf = 300
E = po.OracleExtractor(nf)
frames = np.stack([synth.make_frame(100 + i, 160, 120, fam) for i, fam in enumerate(["blocks", "checker", "flat"])])
cap = nf + 40
kps = np.zeros((len(frames), cap, 28), np.uint8); desc = np.zeros((len(frames), cap, 32), np.uint8)
counts = np.zeros(len(frames), np.int32)
for i, f in enumerate(frames):
    k, d = E.extract(f)
    counts[i] = len(k); kps[i, :len(k)] = k.view(np.uint8).reshape(-1, 28); desc[i, :len(k)] = d
np.savez_compressed(os.path.join(out, "orb_160x120.npz"), frames=frames, kps=kps, desc=desc, counts=counts,
                    nfeatures=np.int32(nf))
print("orb_160x120:", counts)

# matcher fixture: descriptors of two shifted frames + oracle matches
seq, offs = synth.make_sequence(7, 320, 240, 2, "blocks", max_shift=5)
E2 = po.OracleExtractor(500)
k1, d1 = E2.extract(seq[0]); k2, d2 = E2.extract(seq[1])
bi, bd, sd = po.hamming_best2(d1, d2)
m, n = po.match_frames(d1, k1["angle"], d2, k2["angle"], 0.9, 50, True)
np.savez_compressed(os.path.join(out, "match_320x240.npz"), d1=d1, d2=d2, a1=k1["angle"], a2=k2["angle"],
                    best_idx=bi, best_d=bd, second_d=sd, match12=m, nmatch=np.int32(n))
print("match_320x240:", len(d1), len(d2), n)

# BA fixtures: a small pose problem and a small graph with the oracle's solutions
p = synth.make_pose_problem(5, n=200)
ninl, pose, outl, s = po.pose_optimization(p["K4"], p["pose0"], p["Xw"], p["uv"], p["inv_sigma2"])
g = synth.make_ba_graph(6, ncam=6, npts=120, nobs=500, n_fixed=2)
w = g["obs_inv_sigma2"].astype(np.float64)
poses, pts, s2 = po.ba_solve(g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], w,
                             np.ones(len(w), np.uint8), 20)
np.savez_compressed(os.path.join(out, "ba_small.npz"), K4=p["K4"], pose0=p["pose0"], Xw=p["Xw"], uv=p["uv"],
                    inv_sigma2=p["inv_sigma2"], pose_opt=pose, outlier=outl, n_inliers=np.int32(ninl),
                    pose_final_cost=s["final_cost"],
                    gK4=g["K4"], gposes0=g["poses0"], gfixed=g["cam_fixed"], gpts0=g["pts0"], gobs_cam=g["obs_cam"],
                    gobs_pt=g["obs_pt"], gobs_uv=g["obs_uv"], gobs_w=w, gposes=poses, gpts=pts,
                    gfinal_cost=s2["final_cost"], giters=np.int32(s2["iterations"]))
print("ba_small:", ninl, s["final_cost"], s2)
