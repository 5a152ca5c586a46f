"""LocalBA / PoseOptimization throughput legs of bench.py (BASELINE.json configs[2], configs[3])."""
import time

import numpy as np

from . import optimizer, synth


def run(dev, cpu=True, n_localba=6, n_pose_batch=256, rank=0):
    import torch
    out = {}
    # ---- C4: LocalBundleAdjustment, 100 KF x 10k pts x 50k obs, reference two-pass schedule (5 + 10 iterations)
    g = synth.make_ba_graph(0, ncam=100, npts=10000, nobs=50000, n_fixed=2)
    local = np.ones(100, np.uint8)
    args = (g["K4"], g["poses0"], g["cam_fixed"], local, g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
    import threading
    optimizer.local_bundle_adjustment(*args)                       # warm-up (module load, allocator)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_localba):
        ab, poses, pts, er, s1, s2 = optimizer.local_bundle_adjustment(*args)
    dt = time.perf_counter() - t0
    out["localba_single_stream_solves_per_s"] = n_localba / dt
    out["localba_ms_per_solve_latency"] = dt / n_localba * 1e3
    # throughput: independent LocalBA problems (the sub-map sharding of SURVEY 8(e) inside one GPU): `nbatch` problems
    # per call solved in lockstep (ba_local_bundle_adjustment_batch: one grid row per problem), `nthreads` such calls
    # in flight from host threads (one HIP stream + device workspace per thread)
    nbatch, nthreads, n_each = 16, 8, 3
    gs = [g] + [synth.make_ba_graph(s, ncam=100, npts=10000, nobs=50000, n_fixed=2) for s in (1, 2, 3)]
    probs = [(q["K4"], q["poses0"], q["cam_fixed"], local, q["pts0"], q["obs_cam"], q["obs_pt"], q["obs_uv"], q["obs_inv_sigma2"])
             for q in (gs[i % 4] for i in range(nbatch))]
    bar = threading.Barrier(nthreads + 1)

    def work():
        for _ in range(2):
            optimizer.local_bundle_adjustment_batch(probs)
        bar.wait()
        for _ in range(n_each):
            optimizer.local_bundle_adjustment_batch(probs)

    ths = [threading.Thread(target=work) for _ in range(nthreads)]
    for t in ths:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    out["localba_solves_per_s"] = nbatch * nthreads * n_each / dt
    out["localba_concurrency"] = nbatch * nthreads
    out["localba_note"] = ("100 KF x 10000 pts x 50000 obs, reference two-pass schedule (5 Huber + 10 iterations with the "
                           "re-added blocks), host-pointer C ABI end to end (H2D/D2H copies and host structure setup "
                           "included); %d independent problems per lockstep batch x %d host threads" % (nbatch, nthreads))
    out["localba_lm_iterations"] = int(s1["iterations"] + s2["iterations"])
    out["localba_final_cost"] = float(s2["final_cost"])
    # ---- C3: PoseOptimization, 1 camera x 2000 observations, batched device-resident
    probs = [synth.make_pose_problem(100 + i, n=2000) for i in range(8)]
    reps = n_pose_batch // len(probs)
    offs = np.arange(0, 2000 * n_pose_batch + 1, 2000, dtype=np.int32)
    K4 = torch.from_numpy(np.stack([p["K4"] for p in probs] * reps)).to(dev)
    pose0 = torch.from_numpy(np.stack([p["pose0"] for p in probs] * reps)).to(dev)
    Xw = torch.from_numpy(np.concatenate([p["Xw"] for p in probs] * reps)).to(dev)
    uv = torch.from_numpy(np.concatenate([p["uv"] for p in probs] * reps)).to(dev)
    isg = torch.from_numpy(np.concatenate([p["inv_sigma2"] for p in probs] * reps)).to(dev)
    d_off = torch.from_numpy(offs).to(dev)
    poses = pose0.clone()
    optimizer.pose_optimization_batch(K4, poses, Xw, uv, isg, d_off)
    torch.cuda.synchronize()
    nrep = 10
    t0 = time.perf_counter()
    for _ in range(nrep):
        poses.copy_(pose0)
        optimizer.pose_optimization_batch(K4, poses, Xw, uv, isg, d_off)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["poseopt_solves_per_s"] = nrep * n_pose_batch / dt
    out["poseopt_note"] = "1 camera x 2000 observations per problem, %d problems per launch, device-resident" % n_pose_batch
    # ---- C5: GlobalBundleAdjustemnt of one 500-KF sub-map per GPU (~50 k pts, 250 k obs, 2994 x 2994 reduced system),
    #      loop-closure setting: 50 iterations, Huber (src/LoopClosing.cc:656); each rank solves ITS OWN sub-map
    gg = synth.make_ba_graph(1000 + rank, ncam=500, npts=50000, nobs=250000, n_fixed=1)
    gargs = (gg["K4"], gg["poses0"], gg["cam_fixed"], gg["pts0"], gg["obs_cam"], gg["obs_pt"], gg["obs_uv"], gg["obs_inv_sigma2"])
    optimizer.global_bundle_adjustment(*gargs, n_iterations=2)     # warm-up
    t0 = time.perf_counter()
    gposes, gpts, gs = optimizer.global_bundle_adjustment(*gargs, n_iterations=50)
    dt = time.perf_counter() - t0
    out["globalba_500kf_ms"] = dt * 1e3
    out["globalba_500kf_iterations"] = int(gs["iterations"])
    out["globalba_500kf_ms_per_iteration"] = dt * 1e3 / max(int(gs["iterations"]), 1)
    out["globalba_note"] = "500 KF x 50000 pts x 250000 obs, <= 50 LM iterations, host-pointer C ABI end to end, one sub-map per GPU"
    out["_final_points"] = np.ascontiguousarray(gpts)            # merged across ranks with ONE all-gather (bench.py, N > 1)
    if cpu:
        from oracle import pyoracle as po
        t0 = time.perf_counter()
        po.local_ba(*args)
        dt = time.perf_counter() - t0
        out["cpu_localba_solves_per_s"] = 1.0 / dt
        t0 = time.perf_counter()
        for p in probs:
            po.pose_optimization(p["K4"], p["pose0"], p["Xw"], p["uv"], p["inv_sigma2"])
        out["cpu_poseopt_solves_per_s"] = len(probs) / (time.perf_counter() - t0)
        out["cpu_note"] = "oracle (CPU port), 1 thread: 1 LocalBA solve, %d PoseOptimization solves" % len(probs)
    return out
