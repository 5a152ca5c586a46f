"""ORBHIP_BA_ORDER=rcm against the caller's keyframe order (`id`) on maps whose keyframe ids do not follow the covisibility chain:
SURVEY 8(d)'s C4 graph with its keyframes shuffled (synth.shuffle_keyframes), one subprocess per order (the variable is read once per
process).  Prints the skyline the solver saw (band in tiles, factorisation form), the time of single solves and of 64-problem lockstep
batches, and how far the two orders' results are apart.  usage: python tools/order_ab.py"""
import sys, os, json, subprocess, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import numpy as np
    from ceres_mono_orb_slam2_amd import optimizer, synth
    gs = [synth.shuffle_keyframes(synth.make_ba_graph(s, ncam=100, npts=10000, nobs=50000, n_fixed=1), 100 + s) for s in range(16)]
    prob = lambda g: (g["K4"], g["poses0"], g["cam_fixed"], np.ones(100, np.uint8), g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
    optimizer.local_bundle_adjustment(*prob(gs[0]))
    t0 = time.perf_counter()
    for _ in range(5): r = optimizer.local_bundle_adjustment(*prob(gs[0]))
    t_single = (time.perf_counter() - t0) / 5
    plan1 = optimizer.get_last_plan()
    probs = [prob(g) for g in gs] * 4
    optimizer.local_bundle_adjustment_batch(probs)
    t0 = time.perf_counter()
    for _ in range(3): ab, res = optimizer.local_bundle_adjustment_batch(probs)
    t_batch = (time.perf_counter() - t0) / 3
    plan64 = optimizer.get_last_plan()
    print("RESULT " + json.dumps({"single_ms": t_single * 1e3, "batch64_ms": t_batch * 1e3, "plan_single": plan1, "plan_batch": plan64,
                                  "poses": r[1].tolist(), "final_cost": r[5]["final_cost"], "iterations": [r[4]["iterations"], r[5]["iterations"]], "erased": int(r[3].sum())}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(); sys.exit(0)
    import numpy as np
    res = {}
    for v in ("id", "rcm"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, ORBHIP_BA_ORDER=v), capture_output=True, text=True, timeout=900)
        if r.returncode != 0: print(r.stderr[-3000:]); sys.exit(1)
        res[v] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
        d = res[v]
        print("%-3s  band %2d tiles  single: %s, %.2f ms   64-problem batch: %s / %s, %.1f ms (%.0f solves/s)" % (
            v, d["plan_single"]["band_tiles"], d["plan_single"]["lookahead_form"], d["single_ms"], d["plan_batch"]["lookahead_form"],
            d["plan_batch"]["backward_substitution"], d["batch64_ms"], 64e3 / d["batch64_ms"]))
    a, b = res["id"], res["rcm"]
    dp = float(np.max(np.abs(np.array(a["poses"]) - np.array(b["poses"]))))
    print("id vs rcm: final cost %.12g / %.12g (rel. %.1e), iterations %s / %s, erased %d / %d, max |pose difference| %.1e" % (
        a["final_cost"], b["final_cost"], abs(a["final_cost"] - b["final_cost"]) / a["final_cost"], a["iterations"], b["iterations"], a["erased"], b["erased"], dp))
