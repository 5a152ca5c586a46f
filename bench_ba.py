"""LocalBA / PoseOptimization / GlobalBA legs of bench.py (BASELINE.json configs[2], configs[3], configs[4]).

bench.py runs this module in a PROCESS OF ITS OWN (`python bench_ba.py --out DIR ...`, round 5): the twelve
host threads of the batched LocalBA leg want GPU_MAX_HW_QUEUES=12 (one hardware queue per stream, +7 % solves/s), which the HIP
runtime reads once per process and which costs the front-end's host-fed pipeline its copy / compute overlap (110 k -> 50 k frames/s,
VERDICT r4 weak #7).  The child writes `result.json` and `final_points.npy` into DIR."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ceres_mono_orb_slam2_amd import optimizer, synth  # noqa: E402

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X FP64 matrix peak (AMD datasheet; the in-container guide does not state it)


def reduced_solve_flops(g):
    """flops of the Schur-complement GEMMs + the dense Cholesky of one LM iteration.
    Schur: per point seen by k free cameras k^2 * 216 + k * 108; Cholesky: n^3 / 3 with n = 6 * free cameras."""
    free = g["cam_fixed"] == 0
    used = np.zeros(len(free), bool); used[np.unique(g["obs_cam"])] = True
    k = np.bincount(g["obs_pt"][free[g["obs_cam"]]], minlength=len(g["pts0"])).astype(np.float64)
    n = 6.0 * int((free & used).sum())
    return float((k * k * 216 + k * 108).sum()), n ** 3 / 3.0


def skyline_tiles(g, nb_tile=32):
    """The envelope k_chol_wg walks (csrc/ba_host.inc tile_first): per 32-row tile row the first tile column that holds a structural
    non-zero of the reduced system.  Returns (tiles inside the envelope, lower-triangle tiles, tile updates inside, dense tile updates)."""
    free = np.flatnonzero(g["cam_fixed"] == 0)
    used = np.zeros(len(g["cam_fixed"]), bool); used[np.unique(g["obs_cam"])] = True
    free = free[used[free]]
    col = -np.ones(len(g["cam_fixed"]), np.int64); col[free] = np.arange(len(free))
    c = col[g["obs_cam"]]; m = c >= 0
    first_of_pt = np.full(len(g["pts0"]), 1 << 30); np.minimum.at(first_of_pt, g["obs_pt"][m], c[m])
    min_partner = np.arange(len(free)); np.minimum.at(min_partner, c[m], first_of_pt[g["obs_pt"][m]])
    nb = (6 * len(free) + nb_tile - 1) // nb_tile
    f = []
    for i in range(nb):
        cams = range((nb_tile * i) // 6, min(len(free) - 1, (nb_tile * i + nb_tile - 1) // 6) + 1)
        f.append(min([i] + [6 * int(min_partner[b]) // nb_tile for b in cams]))
    inside = sum(i - f[i] + 1 for i in range(nb))
    upd = sum(max(0, cc - max(f[i], f[cc])) for cc in range(nb) for i in range(cc, nb) if f[i] <= cc)
    dense_upd = sum(cc * (nb - cc) for cc in range(nb))
    return inside, nb * (nb + 1) // 2, upd, dense_upd


TILE_FLOPS = 2.0 * 32 ** 3        # one 32 x 32 x 32 tile product on the matrix cores: a tile update, or a tile times the inverse of a diagonal block


def executed_flops(g, skyline=True):
    """What ONE LM iteration EXECUTES in the reduced solve: (matrix-core flops of the factorisation, Schur-complement flops).
    Factorisation: TILE_FLOPS per tile update and per tile solve that the kernels run - inside the skyline when the look-ahead family
    factors (k_chol_wg / k_chol_persist), every tile of the lower triangle otherwise.  The Schur GEMMs (k_ba_schur) run on the VALU."""
    f_schur, _ = reduced_solve_flops(g)
    t_in, t_all, u_in, u_all = skyline_tiles(g)
    return (TILE_FLOPS * ((t_in + u_in) if skyline else (t_all + u_all)), f_schur, (t_in, t_all, u_in, u_all))


def _roof(g, lm_iterations, seconds, what, skyline=True, **extra):
    """`frac` = flops the MATRIX CORES execute (tile updates + tile solves of the factorisation) / time / FP64 matrix peak.
    `executed_incl_schur` adds the Schur GEMMs (VALU) - round 5's `frac_executed`; `dense_equiv_rate` is SURVEY 8(d)'s algorithmic
    figure (Schur + n^3/3 whatever the structure) - round 5's `frac`, kept under a name that says what it is."""
    f_mfma, f_schur, (t_in, t_all, u_in, u_all) = executed_flops(g, skyline)
    _, f_chol_dense = reduced_solve_flops(g)
    ach = f_mfma * lm_iterations / seconds / 1e12
    d = {"achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS, "ms": seconds * 1e3, "what": what,
         "executed_incl_schur": (f_mfma + f_schur) * lm_iterations / seconds / 1e12 / FP64_MFMA_PEAK_TFLOPS,
         "dense_equiv_rate": (f_schur + f_chol_dense) * lm_iterations / seconds / 1e12 / FP64_MFMA_PEAK_TFLOPS,
         "mfma_flops_per_iteration": f_mfma, "schur_flops_per_iteration": f_schur, "lm_iterations": lm_iterations,
         "skyline": {"tiles_inside_envelope": t_in, "lower_triangle_tiles": t_all, "tile_updates_inside": u_in, "dense_tile_updates": u_all,
                     "walked": "envelope" if skyline else "dense"}}
    d.update(extra)
    return d


def _ncores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def run(dev, cpu=True, n_localba=12, n_pose_batch=256, rank=0, quick=False):
    """quick: a shortened leg for tests of the N > 1 plumbing (one single solve, a small lockstep batch, ten GlobalBA iterations, no
    eight-sub-map batch) - its figures are not benchmark numbers and the record says so."""
    import torch
    out = {}
    if quick:
        n_localba = 1; out["quick"] = True
    roof = {"bound": "mfma", "kernel": "FP64-MFMA block Cholesky of the reduced camera system (k_chol_wg / k_chol_persist / k_chol_persist_blk), which walks the system's skyline",
            "definition": "`frac` = flops the matrix cores EXECUTE (2 x 32^3 per tile update and per tile solve that runs: inside the skyline for the look-ahead "
                          "family, the whole lower triangle for the two-level scheme) x LM iterations / time / FP64 matrix peak.  `executed_incl_schur` adds the "
                          "Schur-complement GEMMs, which run on the VALU (round 5's `frac_executed`); `dense_equiv_rate` is SURVEY 8(d)'s algorithmic figure "
                          "(Schur + n^3/3 whatever the structure: round 5's `frac`).  Cases: c4_* = LocalBA 100 KF x 10 k pts x 50 k obs, c5* = GlobalBA 500 KF x 50 k x "
                          "250 k; plain = SURVEY 8(d)'s odometry band (consecutive-view tracks), covis = windowed tracks with gaps + a current keyframe that shares "
                          ">= 15 landmarks with every local keyframe (src/CeresOptimizer.cc:353-363), dense = every keyframe pair shares landmarks, loop = a chain whose "
                          "ends are tied by a loop closure (synth.make_ba_graph_covis)",
            "peak_source": "AMD MI355X datasheet, FP64 matrix 78.6 TFLOP/s", "cases": {}}
    # ---- C4: LocalBundleAdjustment, 100 KF x 10k pts x 50k obs (all free except the gauge keyframe 0), reference two-pass
    #      schedule (5 + 10 iterations)
    g = synth.make_ba_graph(0, ncam=100, npts=10000, nobs=50000, n_fixed=1)
    local = np.ones(100, np.uint8)
    args = (g["K4"], g["poses0"], g["cam_fixed"], local, g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
    f_schur, f_chol = reduced_solve_flops(g)
    for _ in range(1 if quick else 3):
        optimizer.local_bundle_adjustment(*args)                   # warm-up (module load, allocator, pinned staging: the first calls on a fresh box run 1 - 2 ms long)
    torch.cuda.synchronize()
    optimizer.set_profiling(True); optimizer.get_profile()
    t0 = time.perf_counter()
    for _ in range(n_localba):
        ab, poses, pts, er, s1, s2 = optimizer.local_bundle_adjustment(*args)
    dt = time.perf_counter() - t0
    dev_ms, nsolv, nit = optimizer.get_profile()                   # HIP events on the solver's stream, both passes of every solve
    out["localba_single_stream_solves_per_s"] = n_localba / dt
    out["localba_ms_per_solve_latency"] = dt / n_localba * 1e3
    out["localba_ms_per_solve_device"] = dev_ms / n_localba
    roof["cases"]["c4_single"] = _roof(g, nit, dev_ms * 1e-3,
                                       "one LocalBA at a time; device time of the %d LM iterations of %d solves (HIP events on the solve stream)" % (nit, n_localba),
                                       cholesky_n=600 - 6, plan=optimizer.get_last_plan())
    # throughput: independent LocalBA problems (the sub-map sharding of SURVEY 8(e) inside one GPU): `nbatch` problems
    # per call solved in lockstep (ba_local_bundle_adjustment_batch: one grid row per problem), `nthreads` such calls
    # in flight from host threads (one HIP stream + device workspace per thread)
    # (64 per call since round 2: the lockstep launches are bound by the sum of their kernels' exclusive times - ~0.85 ms of GPU
    # per solve - and larger launches fill the chip better: 16 x 8 -> 1110, 32 x 8 -> 1170, 64 x 8 -> 1270 solves/s)
    # (12 threads x 64 problems: 1737 - 1791 solves/s against 1552 - 1698 with 8 in round 3, tools/ba_batch_thr.py.  Round 5: EIGHT timed
    # batches per thread instead of two - the threads leave the barrier together, so the window opens with every one of them in its host
    # structure pass and the GPU idle, and closes with the stragglers alone: with two batches per thread that ramp was 16 % of the window
    # (tools/ba_batch_thr.py 64:12:N, N = 2 / 4 / 8 / 16: 2848 / 3151 / 3312 / 3294 solves/s; a kernel trace of the steady part shows one
    # batch iteration per 1.2 ms = 3555 solves/s))
    nbatch, nthreads, n_each = (8, 2, 1) if quick else (64, 12, 8)

    def batched_leg(gs, what):
        """`nthreads` host threads, each solving the lockstep batch `gs` n_each times; returns (solves/s, roofline case)."""
        probs = [(q["K4"], q["poses0"], q["cam_fixed"], local, q["pts0"], q["obs_cam"], q["obs_pt"], q["obs_uv"], q["obs_inv_sigma2"]) for q in gs]
        bar = threading.Barrier(nthreads + 1)
        iters = [0] * nthreads
        plans = [None] * nthreads

        def work(k):
            for _ in range(2):
                optimizer.local_bundle_adjustment_batch(probs)
            optimizer.get_profile()
            bar.wait()
            for _ in range(n_each):
                optimizer.local_bundle_adjustment_batch(probs)
            iters[k] = optimizer.get_profile()[2]
            plans[k] = optimizer.get_last_plan()

        ths = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
        for t in ths:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        nsolves = len(gs) * nthreads * n_each
        mean_it = sum(iters) / float(nsolves)
        # (the problems of a batch differ in their skylines by a tile or two: the executed flops of problem 0 stand for all)
        case = _roof(gs[0], mean_it * nsolves, dt,
                     "%d distinct local maps (%s) per lockstep batch x %d host threads; wall time of %d solves (copies and host structure setup included)"
                     % (len(gs), what, nthreads, nsolves), lm_iterations_per_solve=mean_it, solves_per_s=nsolves / dt, plan=plans[0])
        return nsolves / dt, case

    gs = [g] + [synth.make_ba_graph(s, ncam=100, npts=10000, nobs=50000, n_fixed=1) for s in range(1, nbatch)]   # 64 distinct local maps
    out["localba_solves_per_s"], roof["cases"]["c4_batched"] = batched_leg(gs, "odometry band")
    out["localba_concurrency"] = nbatch * nthreads
    # the same size with the structures a map of the reference has: windowed tracks + the current keyframe's shared
    # landmarks (an arrowhead over a band of 30 keyframes), and a full reduced system
    for st in ("covis", "dense"):
        gq = [synth.make_ba_graph_covis(100 + s, ncam=100, npts=10000, nobs=50000, structure=st) for s in range(nbatch)]
        out["localba_%s_solves_per_s" % st], roof["cases"]["c4_%s" % st] = batched_leg(gq, st)
        if not quick:                                              # and one solve at a time
            qa = (gq[0]["K4"], gq[0]["poses0"], gq[0]["cam_fixed"], local, gq[0]["pts0"], gq[0]["obs_cam"], gq[0]["obs_pt"], gq[0]["obs_uv"], gq[0]["obs_inv_sigma2"])
            optimizer.local_bundle_adjustment(*qa); optimizer.get_profile()
            t0 = time.perf_counter()
            for _ in range(6):
                optimizer.local_bundle_adjustment(*qa)
            dts = time.perf_counter() - t0
            dms, _, nits = optimizer.get_profile()
            roof["cases"]["c4_%s_single" % st] = _roof(gq[0], nits, dms * 1e-3, "one LocalBA (%s) at a time; device time of the %d LM iterations of 6 solves" % (st, nits),
                                                       ms_per_solve_latency=dts / 6 * 1e3, plan=optimizer.get_last_plan())
    out["localba_note"] = ("100 KF x 10000 pts x 50000 obs, reference two-pass schedule (5 Huber + 10 iterations with the "
                           "re-added blocks), host-pointer C ABI end to end (H2D/D2H copies and host structure setup "
                           "included); %d distinct problems per lockstep batch x %d host threads; all keyframes free except "
                           "the gauge keyframe 0" % (nbatch, nthreads))
    out["localba_lm_iterations"] = int(s1["iterations"] + s2["iterations"])
    out["localba_final_cost"] = float(s2["final_cost"])
    # ---- C3: PoseOptimization, 1 camera x 2000 observations, batched device-resident
    pprobs = [synth.make_pose_problem(100 + i, n=2000) for i in range(8)]
    probs = pprobs
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
    optimizer.get_profile()
    t0 = time.perf_counter()
    gposes, gpts, gsum = optimizer.global_bundle_adjustment(*gargs, n_iterations=10 if quick else 50)
    dt = time.perf_counter() - t0
    dev_ms, _, nit = optimizer.get_profile()
    plan5 = optimizer.get_last_plan()
    roof["cases"]["c5"] = _roof(gg, nit, dev_ms * 1e-3, "one 500-KF GlobalBA (odometry band); device time of its %d LM iterations (HIP events on the solve stream)" % nit,
                                skyline=plan5["lookahead_form"] != "none", cholesky_n=6 * 499, plan=plan5)
    # a LOOP-CLOSED 500-keyframe map: what GlobalBundleAdjustemnt runs on (src/LoopClosing.cc:656) - a chain whose last block rows reach back to column 0
    gl = synth.make_ba_graph_covis(3000 + rank, ncam=500, npts=50000, nobs=250000, structure="loop")
    largs = (gl["K4"], gl["poses0"], gl["cam_fixed"], gl["pts0"], gl["obs_cam"], gl["obs_pt"], gl["obs_uv"], gl["obs_inv_sigma2"])
    optimizer.set_profiling(True)
    optimizer.global_bundle_adjustment(*largs, n_iterations=2); optimizer.get_profile()
    t0 = time.perf_counter()
    _, _, lsum = optimizer.global_bundle_adjustment(*largs, n_iterations=5 if quick else 20)
    dtl = time.perf_counter() - t0
    lms, _, lnit = optimizer.get_profile()
    optimizer.set_profiling(False)
    planl = optimizer.get_last_plan()
    roof["cases"]["c5_loop"] = _roof(gl, lnit, lms * 1e-3, "one loop-closed 500-KF GlobalBA; device time of its %d LM iterations" % lnit,
                                     skyline=planl["lookahead_form"] != "none", ms_per_iteration=lms / max(lnit, 1), wall_ms=dtl * 1e3, plan=planl)
    out["globalba_loop_500kf_ms_per_iteration"] = lms / max(lnit, 1)
    if not quick:
        # a DENSE 2994 x 2994 reduced system (every keyframe pair of 500 shares landmarks): the two-level family (k_chol_persist_blk), every tile runs
        gd = synth.make_ba_graph_covis(3100 + rank, ncam=500, npts=50000, nobs=250000, structure="dense")
        dargs = (gd["K4"], gd["poses0"], gd["cam_fixed"], gd["pts0"], gd["obs_cam"], gd["obs_pt"], gd["obs_uv"], gd["obs_inv_sigma2"])
        optimizer.set_profiling(True)
        optimizer.global_bundle_adjustment(*dargs, n_iterations=2); optimizer.get_profile()
        optimizer.global_bundle_adjustment(*dargs, n_iterations=8)
        dms, _, dnit = optimizer.get_profile()
        optimizer.set_profiling(False)
        pland = optimizer.get_last_plan()
        roof["cases"]["c5_dense"] = _roof(gd, dnit, dms * 1e-3, "one 500-KF GlobalBA whose every keyframe pair shares landmarks (a dense 2994 x 2994 reduced system); device time of "
                                          "its %d LM iterations - the Schur complement's 124 750 blocks of ~4 pairs each included, which is most of it" % dnit,
                                          skyline=pland["lookahead_form"] != "none", ms_per_iteration=dms / max(dnit, 1), plan=pland)
        out["globalba_dense_500kf_ms_per_iteration"] = dms / max(dnit, 1)
    # ---- C5 as BASELINE config 5 words it, on ONE GPU: the eight 500-KF sub-maps as one lockstep batch (ba_solve_batch: what a node
    #      with fewer GPUs than sub-maps does).  A single GlobalBA is bound by the latency chain of its factorisation (c5 above);
    #      eight in lockstep share every launch of the chain.  (Eight host threads with one solve each: 464 ms against 379; 323 with the
    #      structure passes of the eight problems on four host threads.)
    def _sub(g):
        w = np.asarray(g["obs_inv_sigma2"], np.float32).astype(np.float64)          # F7: weight = invSigma2 (a float in the reference)
        return (g["K4"], g["poses0"], g["cam_fixed"], g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], w, np.ones(len(w), np.uint8))
    if not quick:
        subs = [_sub(gg)] + [_sub(synth.make_ba_graph(2000 + 8 * rank + k, ncam=500, npts=50000, nobs=250000, n_fixed=1)) for k in range(1, 8)]
        optimizer.bundle_adjustment_batch(subs, n_iterations=2)                           # warm-up (workspace, graph capture)
        t0 = time.perf_counter()
        res8 = optimizer.bundle_adjustment_batch(subs, n_iterations=50)
        dt8 = time.perf_counter() - t0
        it8 = sum(int(r8[2]["iterations"]) for r8 in res8)
        roof["cases"]["c5_batched8"] = _roof(gg, it8, dt8,
                                             "eight distinct 500-KF GlobalBA sub-maps as one lockstep batch on this GPU; wall time of their %d LM "
                                             "iterations, copies and host structure setup (the calling thread and its helpers, ORBHIP_BA_PREP_THREADS) included" % it8,
                                             plan=optimizer.get_last_plan())
        out["globalba_8_submaps_ms"] = dt8 * 1e3
    out["roofline"] = roof
    out["globalba_500kf_ms"] = dt * 1e3
    out["globalba_500kf_iterations"] = int(gsum["iterations"])
    out["globalba_500kf_ms_per_iteration"] = dt * 1e3 / max(int(gsum["iterations"]), 1)
    out["globalba_note"] = "500 KF x 50000 pts x 250000 obs, <= 50 LM iterations, host-pointer C ABI end to end, one sub-map per GPU"
    out["_final_points"] = np.ascontiguousarray(gpts)            # merged across ranks with ONE all-gather (bench.py, N > 1)
    if cpu:
        from oracle import pyoracle as po
        cores = _ncores()
        t0 = time.perf_counter()
        po.local_ba(*args)
        dt1 = time.perf_counter() - t0
        po.set_ba_threads(4)                                       # options.num_threads = 4 (src/CeresOptimizer.cc:516)
        t0 = time.perf_counter()
        po.local_ba(*args)
        dt4 = time.perf_counter() - t0
        po.set_ba_threads(1)
        lprobs = [(q["K4"], q["poses0"], q["cam_fixed"], local, q["pts0"], q["obs_cam"], q["obs_pt"], q["obs_uv"], q["obs_inv_sigma2"]) for q in gs]
        nthr = min(cores, 32)
        ths = [threading.Thread(target=po.local_ba, args=lprobs[c % len(lprobs)]) for c in range(nthr)]
        c0, t0 = time.process_time(), time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dta = time.perf_counter() - t0
        used = max(1, int(round((time.process_time() - c0) / dta)))
        t0 = time.perf_counter()
        for p in pprobs:
            po.pose_optimization(p["K4"], p["pose0"], p["Xw"], p["uv"], p["inv_sigma2"])
        dtp = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "kind": "port", "host_cores": cores,
            "localba_1_thread": {"value": 1.0 / dt1, "unit": "solves/s", "cores": 1, "sample": "1 LocalBA solve at C4 size, %.2f s" % dt1},
            "localba_4_threads": {"value": 1.0 / dt4, "unit": "solves/s", "cores": 4,
                                  "sample": "the same solve with 4 worker threads in the evaluation and the Schur elimination (the reference's "
                                            "num_threads = 4), %.2f s" % dt4},
            "localba_all_cores": {"value": nthr / dta, "unit": "solves/s", "cores": used, "threads": nthr,
                                  "sample": "%d independent LocalBA solves, one single-thread oracle each, %.2f s; cores = CPU seconds / wall seconds" % (nthr, dta)},
            "poseopt_1_thread": {"value": len(pprobs) / dtp, "unit": "solves/s", "cores": 1, "sample": "%d PoseOptimization solves" % len(pprobs)},
        }
        out["cpu_localba_solves_per_s"] = 1.0 / dt1
        out["cpu_poseopt_solves_per_s"] = len(pprobs) / dtp
    return out


def main(argv=None):
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--cpu", type=int, default=0)
    ap.add_argument("--oracle-lib", default="", help="the -march=native oracle build bench.py verified (empty: the canonical build)")
    ap.add_argument("--quick", type=int, default=0)
    ap.add_argument("--out", required=True)
    a = ap.parse_args(argv)
    import torch
    from ceres_mono_orb_slam2_amd import _lib
    torch.cuda.set_device(a.device)
    _lib.check(_lib.load().orbhip_set_default_device(a.device), "orbhip_set_default_device")
    if a.cpu and a.oracle_lib:
        from oracle import pyoracle as po
        po.use_library(a.oracle_lib)
    res = run(torch.device("cuda", a.device), cpu=bool(a.cpu), rank=a.rank, quick=bool(a.quick))
    pts = res.pop("_final_points")
    res["process"] = {"own_process": True, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
    os.makedirs(a.out, exist_ok=True)
    np.save(os.path.join(a.out, "final_points.npy"), pts)
    with open(os.path.join(a.out, "result.json"), "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
