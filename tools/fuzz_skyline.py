"""Random reduced-system STRUCTURES through the three forms of the factorisation: every problem alone (k_chol_persist, walking its skyline),
inside a lockstep batch of >= 32 (k_chol_wg, walking its skyline) and - a second process with ORBHIP_BA_PERSIST=0 ORBHIP_BA_WG=0 - through the
dense step kernels: poses, points and summaries must be BIT-IDENTICAL.  Graphs: 6 .. 210 keyframes (2 .. 39 block rows: both sides of
the 32-row limit), consecutive-view tracks plus random far links (a dense last row, a dense first column, ragged links, several at once),
one or two fixed keyframes, keyframes without observations, loop closures (far landmarks seen by the first and the last keyframes: a narrow
prefix of block rows under dense last ones - the ring + workgroup-set form of k_chol_persist), up to 300 keyframes (56 block rows).  usage: python tools/fuzz_skyline.py [first_seed=0] [batches=4]"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, hashlib, numpy as np
sys.path.insert(0, sys.argv[1])
from ceres_mono_orb_slam2_amd import optimizer, synth
first, nbatch = int(sys.argv[2]), int(sys.argv[3])
def graph(seed):
    rng = np.random.default_rng(seed)
    ncam = int(rng.choice([6, 11, 17, 24, 40, 64, 90, 120, 160, 175, 210, 260, 300]))
    npts = int(ncam * rng.integers(8, 20)); nobs = int(npts * rng.uniform(3.0, 6.0))
    g = synth.make_ba_graph(seed, ncam=ncam, npts=npts, nobs=nobs, n_fixed=int(rng.integers(1, 3)))
    oc, op, uv, w = list(g["obs_cam"]), list(g["obs_pt"]), list(g["obs_uv"]), list(g["obs_inv_sigma2"])
    have = set(zip(oc, op))
    free = np.flatnonzero(g["cam_fixed"] == 0)
    for kind in rng.permutation(4)[: int(rng.integers(0, 3))]:
        n = int(rng.integers(4, max(npts // 5, 5)))
        if kind == 0: cams = np.full(n, ncam - 1)
        elif kind == 1: cams = np.full(n, free[0])
        elif kind == 2: cams = rng.integers(0, ncam, n)
        else: cams = np.full(n, free[len(free) // 2])
        for c, p in zip(cams, rng.integers(0, npts, n)):
            c = int(c); p = int(p)
            if (c, p) in have: continue
            x, z = synth.project(g["K4"][0], g["poses_gt"][c], g["pts_gt"][p][None])
            if z[0] < 1.0: continue
            have.add((c, p)); oc.append(c); op.append(p); uv.append(x[0] + rng.normal(0, 1.0, 2)); w.append(1.0)
    pts0 = g["pts0"]
    if rng.random() < 0.35:                                     # a loop closure: far landmarks seen by the first free and the last keyframes (dense LAST block rows)
        nf, ne = int(rng.integers(20, 120)), int(rng.integers(1, 7))
        X = np.stack([rng.uniform(-20, 20, nf), rng.uniform(-4, 4, nf), 0.8 * ncam + rng.uniform(30, 80, nf)], 1)
        for k in range(nf):
            for c in list(free[:ne]) + list(range(ncam - ne, ncam)):
                x, z = synth.project(g["K4"][0], g["poses_gt"][int(c)], X[k][None])
                oc.append(int(c)); op.append(npts + k); uv.append(x[0] + rng.normal(0, 1.0, 2)); w.append(1.0)
        pts0 = np.vstack([pts0, X * 1.01])
    oc = np.array(oc, np.int32); op = np.array(op, np.int32); uv = np.array(uv); w = np.array(w, np.float64)
    if rng.random() < 0.3:                                      # a keyframe that loses all its observations (not in the reduced system)
        drop = int(rng.integers(1, ncam)); m = oc != drop
        oc, op, uv, w = oc[m], op[m], uv[m], w[m]
    return (g["K4"], g["poses0"], g["cam_fixed"], pts0, oc, op, uv, w, np.ones(len(oc), np.uint8))
out = {}
for b in range(nbatch):
    probs = [graph(first + 36 * b + k) for k in range(36)]
    res = optimizer.bundle_adjustment_batch(probs, n_iterations=5)
    for k, (pr, (poses, pts, s)) in enumerate(zip(probs, res)):
        p1, x1, s1 = optimizer.bundle_adjustment(*pr, n_iterations=5)
        same = s == s1 and np.array_equal(poses, p1) and np.array_equal(pts, x1)
        h = hashlib.sha256(p1.tobytes() + x1.tobytes() + repr(sorted(s1.items())).encode()).hexdigest()
        out[str(first + 36 * b + k)] = [h, bool(same), int(s1["iterations"])]
print("RESULT " + json.dumps(out))
'''
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
res = []
for env in ({}, {"ORBHIP_BA_PERSIST": "0", "ORBHIP_BA_WG": "0"}):
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(first), str(nb)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=3000)
    if r.returncode: sys.exit(r.stderr[-3000:])
    res.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:]))
bad = 0
for k in res[0]:
    a, b = res[0][k], res[1][k]
    if not a[1] or not b[1] or a[0] != b[0]:
        bad += 1; print("DIFF seed", k, "batch == single (skyline forms):", a[1], " (step kernels):", b[1], " skyline == steps:", a[0] == b[0])
its = [v[2] for v in res[0].values()]
print("fuzz_skyline: %d problems from seed %d, %d differences; LM iterations %d .. %d" % (len(res[0]), first, bad, min(its), max(its)))
sys.exit(1 if bad else 0)
