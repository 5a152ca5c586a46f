"""k_chol_bsolve_sky4 (four waves, blocks requested three steps ahead) against k_chol_bsolve_sky (1024 threads): the same bits?
Runs itself twice (ORBHIP_BA_BSOLVE_WAVES=1 / 0, read once per process) over single solves of several sizes, a lockstep batch and a
C5-like GlobalBA, and compares the SHA-256 of every output.  usage: python tools/bsolve_ab.py"""
import sys, os, json, hashlib, subprocess, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import numpy as np
    from ceres_mono_orb_slam2_amd import optimizer, synth
    out = []
    def h(*arrs):
        m = hashlib.sha256()
        for a in arrs: m.update(np.ascontiguousarray(a).tobytes())
        return m.hexdigest()[:16]
    # single LocalBA solves: 1 .. 41 block rows (a partial top super-block, exactly one, several)
    for seed, ncam, npts, nobs in ((1, 8, 300, 1500), (2, 44, 2000, 9000), (3, 100, 10000, 50000), (4, 130, 6000, 30000), (5, 220, 9000, 45000)):
        g = synth.make_ba_graph(seed, ncam=ncam, npts=npts, nobs=nobs, n_fixed=1)
        t0 = time.perf_counter()
        r = optimizer.local_bundle_adjustment(g["K4"], g["poses0"], g["cam_fixed"], np.ones(ncam, np.uint8), g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
        out.append(["lba", ncam, h(r[1], r[2], r[3]), r[4]["iterations"] + r[5]["iterations"], optimizer.get_last_plan()["backward_substitution"]])
    # GlobalBA, 300 keyframes, band 3
    g = synth.make_ba_graph(7, ncam=300, npts=20000, nobs=100000, n_fixed=1)
    poses, pts, s = optimizer.global_bundle_adjustment(g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"], n_iterations=6)
    out.append(["gba", 300, h(poses, pts), s["iterations"], optimizer.get_last_plan()["backward_substitution"]])
    # a lockstep batch of 8
    gs = [synth.make_ba_graph(20 + k, ncam=60 + 7 * k, npts=3000, nobs=15000, n_fixed=1) for k in range(8)]
    ab, res = optimizer.local_bundle_adjustment_batch([(g["K4"], g["poses0"], g["cam_fixed"], np.ones(len(g["cam_fixed"]), np.uint8), g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"]) for g in gs])
    out.append(["batch", 8, h(*[a for r in res for a in r[:3]]), 0, optimizer.get_last_plan()["backward_substitution"]])
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(); sys.exit(0)
    res = {}
    for v in ("1", "0"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, ORBHIP_BA_BSOLVE_WAVES=v), capture_output=True, text=True, timeout=900)
        if r.returncode != 0: print(r.stderr[-3000:]); sys.exit(1)
        res[v] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    bad = 0
    for a, b in zip(res["1"], res["0"]):
        same = a == b
        bad += 0 if same else 1
        print(("same bits   " if same else "DIFFERENT   "), a, "" if same else b)
    print("differences:", bad)
    sys.exit(1 if bad else 0)
