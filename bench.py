#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native ORB front-end + BA back-end.

Metric (BASELINE.json): "frames/sec ORB extract+match @1241x376 + LocalBA solves/sec; 1/2/4/8 GPU".
`value` = whole-job frames/s of ORB extract (2000 features, 8 levels) + brute-force Hamming match
of every frame against its predecessor, on synthetic 1241x376 frames already resident in HBM
(BASELINE.json configs[1]).  LocalBA solves/s (configs[3]) is reported in `localba`.

A "step" = one pass of the hot path over one batch of `--batch` frames per GPU.  Frames shard
across ranks with no data-path collective (weak scaling): value = frames all ranks processed / max
rank time.  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W_IMG, H_IMG, NFEAT = 1241, 376, 2000
# algorithmic bytes per 1241x376 frame (DESIGN.md "Kernels")
PX_TOTAL = 1444097
BYTES = {"pyramid": 1407767 + 977481, "fast_cells": 1444097, "blur": 2 * 1444097, "describe": 2000 * (32 + 28)}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_TOPS = 39.3          # 256 CU x 4 SIMD x 16 lanes x 2.4 GHz: a wave64 integer VALU instruction issues over 4 cycles;
                               # measured 35-39 T lane-instr/s for xor/bcnt/pk_*16/dot4/dot2/sad/alignbyte (tools/ubench/valu_rate.hip)
MATCH_LANE_OPS_PER_PAIR = 19.5   # VALU instructions per Hamming distance in k_match_pairs (ISA-checked: 8 xor + 8 v_bcnt + v_lshl_or + v_max + v_min + half a v_min3)


def make_frames(batch, seed):
    """`batch` frames cut from a few synthetic canvases as chains of <=8 px translations."""
    from ceres_mono_orb_slam2_amd import synth
    fams = ["blocks", "checker", "blocks", "checker", "blocks", "flat", "blocks", "checker"]
    ncanvas = min(len(fams), max(1, batch // 8))
    per = (batch + ncanvas - 1) // ncanvas
    out = []
    for c in range(ncanvas):
        fr, _ = synth.make_sequence(seed * 100 + c, W_IMG, H_IMG, per, fams[c], max_shift=8)
        out.append(fr)
    return np.concatenate(out)[:batch]


def cpu_baseline(frames, n_sample):
    """Oracle (CPU port of the reference algorithm), single thread, on a bounded sample of the same workload."""
    from oracle import pyoracle as po
    E = po.OracleExtractor(NFEAT)
    n = min(n_sample, len(frames))
    t0 = time.perf_counter()
    prev = None
    for i in range(n):
        k, d = E.extract(frames[i])
        if prev is not None:
            po.match_frames(d, k["angle"], prev[1], prev[0]["angle"], 0.9, 50, True)
        else:
            po.match_frames(d, k["angle"], d, k["angle"], 0.9, 50, True)
        prev = (k, d)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d of the bench's 1241x376 frames: oracle extract (2000 features) + 1 brute-force match each, "
                      "single thread, %.1f s" % (n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=160)   # ~13 s of single-thread oracle work
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-ba", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    # ORBHIP_BENCH_SHARED_GPU=1: dry run of the N > 1 code path on a box with ONE GPU (every rank uses device 0, gloo carries
    # the collectives on host tensors); the real multi-GPU run is one rank per GPU over RCCL
    shared = os.environ.get("ORBHIP_BENCH_SHARED_GPU") == "1"
    if shared:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if shared else dev          # where collective payloads live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from ceres_mono_orb_slam2_amd import ORBextractor, ORBmatcher, _lib
    _lib.check(_lib.load().orbhip_set_default_device(local_rank), "orbhip_set_default_device")
    B = args.batch
    frames = make_frames(B, seed=rank)                 # each rank: its own frames (frame sharding)
    d_frames = torch.from_numpy(frames).to(dev)
    ex = ORBextractor(NFEAT, 1.2, 8, 20, 7, device=local_rank)
    mt = ORBmatcher(0.9, True)
    cap = ex.max_keypoints
    kps = torch.empty((B, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    match12 = torch.empty((B, cap), dtype=torch.int32, device=dev)
    nmatch = torch.empty((B,), dtype=torch.int32, device=dev)
    pair_a = torch.arange(B, dtype=torch.int32, device=dev)
    pair_b = (pair_a + B - 1) % B                      # frame i against its predecessor

    def step():
        ex.extract_batch(d_frames, out=(kps, desc, counts))
        mt.match_frames_batch(kps, desc, counts, pair_a, pair_b, out=(match12, nmatch))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ex.set_profiling(True)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ex.extract_batch(d_frames, out=(kps, desc, counts))
        ev[k][0].record()
        mt.match_frames_batch(kps, desc, counts, pair_a, pair_b, out=(match12, nmatch))
        ev[k][1].record()
    barrier()
    dt = time.perf_counter() - t0
    stage_ms, ncalls = ex.stage_ms()
    ex.set_profiling(False)
    match_ms = sum(a.elapsed_time(b) for a, b in ev)
    from ceres_mono_orb_slam2_amd import sharding
    dt = sharding.max_over_ranks(dt, device=cdev)

    # sanity of the timed work (not timed): every frame produced keypoints and matches
    c = counts.cpu().numpy(); nm = nmatch.cpu().numpy()
    assert (c > 0).all(), "extractor produced an empty/overflowed frame"
    mean_kp, mean_match = float(c.mean()), float(nm.mean())

    # ---- LocalBA / PoseOptimization legs: every rank solves its own independent problems (sub-map sharding, no
    #      collective in the solve); for N > 1 the landmark updates are merged with ONE all-gather.
    localba = None
    if not args.no_ba:
        ok = 1
        try:
            from ceres_mono_orb_slam2_amd import ba_bench
            localba = ba_bench.run(dev, cpu=(not args.no_cpu) and world == 1, rank=rank)
        except Exception as e:                       # never lose the headline line to the secondary leg
            localba, ok = {"error": repr(e)}, 0
        if world > 1:
            flag = torch.tensor([ok], dtype=torch.int32, device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # collectives below only if EVERY rank's leg succeeded
            if int(flag.item()) == 1:
                t = torch.tensor([localba["localba_solves_per_s"], localba["poseopt_solves_per_s"]], dtype=torch.float64, device=cdev)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                localba["localba_solves_per_s"], localba["poseopt_solves_per_s"] = float(t[0]), float(t[1])
                pts = torch.from_numpy(localba["_final_points"]).to(cdev)
                torch.cuda.synchronize(); tg = time.perf_counter()
                allp, _, counts = sharding.allgather_landmarks(pts)
                torch.cuda.synchronize()
                localba["landmark_allgather_ms"] = (time.perf_counter() - tg) * 1e3
                localba["landmark_allgather_points"] = int(allp.shape[0])
        if isinstance(localba, dict):
            localba.pop("_final_points", None)
    if rank == 0:
        K = args.steps
        fps = B * world * K / dt
        per_call = {k: v / max(ncalls, 1) for k, v in stage_ms.items()}
        per_call["match"] = match_ms / K
        # HBM roofline of the dominant streaming kernel of the extract path
        hbm_stages = {k: per_call[k] for k in ("pyramid", "fast_cells", "blur", "describe")}
        dom = max(hbm_stages, key=hbm_stages.get)
        ach = BYTES[dom] * B / (per_call[dom] * 1e-3) / 1e9
        # HBM traffic per launch from the committed rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, separate runs;
        # profiles/r01_pmc_traffic.json) scaled to this batch; None if the summary is absent
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            kname = {"pyramid": "k_resize", "fast_cells": "k_fast_cells", "blur": "k_blur7", "describe": "k_describe"}[dom]
            traffic = pmc["kernels"][kname]["hbm_bytes_per_frame"] * B
        except Exception:
            pass
        roof = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": BYTES[dom] * B}
        # the kernel is an HBM kernel by bytes but binds on the integer VALU issue rate: report that fraction too, from the
        # committed PMC pass (profiles/r01_pmc_valu.json: SQ_INSTS_VALU x 4 cycles / SIMD-cycles of the launch)
        try:
            vp = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_valu.json")))["kernels"][kname]
            roof["valu_issue_frac"] = vp["valu_busy_frac"]
            roof["valu_insts_per_wave"] = vp["valu_insts_per_wave"]
        except Exception:
            pass
        pairs_per_s = B / (per_call["match"] * 1e-3)
        lane_ops = mean_kp * mean_kp * MATCH_LANE_OPS_PER_PAIR
        kernels = {k: {"ms_per_launch_batch": v} for k, v in per_call.items()}
        for k in BYTES:
            kernels[k]["algorithmic_GBps"] = BYTES[k] * B / (per_call[k] * 1e-3) / 1e9
        kernels["match"]["valu_Tops"] = pairs_per_s * lane_ops / 1e12
        kernels["match"]["valu_frac_of_peak"] = pairs_per_s * lane_ops / 1e12 / VALU_PEAK_TOPS
        out = {
            "metric": "frames/sec ORB extract+match @1241x376 + LocalBA solves/sec; 1/2/4/8 GPU",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "KITTI 1241x376, 2000 features/frame, 8 levels, ORB extract + brute-force Hamming "
                                   "match vs previous frame (ratio 0.9, TH_LOW 50, rotation histogram)",
                       "frames_per_gpu_per_step": B, "sharding": "frames across ranks, no collective",
                       "mean_keypoints": mean_kp, "mean_matches": mean_match},
            "roofline": roof,
            "kernels": kernels,
        }
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(frames, args.cpu_sample)
        if localba is not None:
            out["localba"] = localba
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
