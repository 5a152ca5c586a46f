#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native ORB front-end + BA back-end.

Metric (BASELINE.json): "frames/sec ORB extract+match @1241x376 + LocalBA solves/sec; 1/2/4/8 GPU".
`value` = whole-job frames/s of ORB extract (2000 features, 8 levels) + brute-force Hamming match
of every frame against its predecessor, on synthetic 1241x376 frames already resident in HBM
(BASELINE.json configs[1]).  LocalBA solves/s (configs[3]), PoseOptimization (configs[2]) and the
500-KF GlobalBA sub-map per GPU + landmark all-gather (configs[4]) are reported in `localba`.

A "step" = one pass of the hot path over `--batches-per-step` batches of `--batch` frames per GPU
(32 x 256 = 8192 distinct resident frames by default, so that 20 steps are > 1 s of GPU time).
Frames shard across ranks with no data-path collective (weak scaling): value = frames all ranks
processed / max rank time.  One JSON line is printed by rank 0.

Launching: `python bench.py --gpus N` spawns its own N ranks (one process per GPU, RCCL) when it
is not already running under torchrun (WORLD_SIZE unset); under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one of the ranks.
Either way the rank count must equal --gpus.  ORBHIP_BENCH_SHARED_GPU=1 = dry run of the N > 1
path on a box with fewer GPUs (every rank on device 0, gloo on host tensors).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

# The LocalBA leg keeps 12 host threads' HIP streams busy; the runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4: streams that share a queue serialise).  8 queues: +5 % solves/s, one per thread another ~2 % (same-box sweeps of
# tools/ba_batch_thr.py 64:12: 8 -> 2573, 12 -> 2631, 16 -> 2612, 24 -> 2520 solves/s).  The runtime
# reads it once per process, and more than 4 queues cost the front-end's host-fed pipeline its copy / compute overlap (round 4:
# 110 k -> 50 k frames/s) - so the BA legs run in a process of their own with this value (run_ba_leg below; a value the caller
# exported wins) and the front-end process keeps the runtime's default.
BA_LEG_HW_QUEUES = "12"
# The front-end process: two pipelined extract + match streams, each with a side stream for its blur, must not share a hardware
# queue (with the default 4 they alias: 156 k frames/s against 170 k with 6, 8 or 12; round-5 sweep).  The
# host-fed pipeline no longer cares (pcie_pipeline below: its stage hand-offs are made on the host).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W_IMG, H_IMG, NFEAT = 1241, 376, 2000
# algorithmic bytes per 1241x376 frame (DESIGN.md "Kernels")
PX_TOTAL = 1444097
BYTES = {"pyramid": 1407767 + 977481, "fast_cells": 1444097, "blur": 2 * 1444097, "describe": 2000 * (32 + 28)}
# k_octree: the candidate keys it reads (4 B each, ~33 k per frame on the bench's frames), the per-cell counts, the keys' node ids
# written and read once, the selected keys written: 149 KB per frame (profiles/r04_pmc_traffic.json agrees) - a latency-bound
# kernel (one workgroup per (frame, level), a serial split loop): its HBM fraction is tiny by construction, the line reports its
# VALU-busy and waiting fractions beside it
OCTREE_BYTES = 149000
KERNEL_OF = {"pyramid": "k_resize_mfma", "fast_cells": "k_fast_cells", "blur": "k_blur7_mfma", "describe": "k_describe", "octree": "k_octree"}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_TOPS = 39.3          # 256 CU x 4 SIMD x 16 lanes x 2.4 GHz: a wave64 integer VALU instruction issues over 4 cycles;
                               # measured 35-39 T lane-instr/s for xor/bcnt/pk_*16/dot4/dot2/sad/alignbyte (tools/ubench/valu_rate.hip)
MATCH_LANE_OPS_PER_PAIR = 19.5   # VALU instructions per Hamming distance in k_match_pairs (ISA-checked: 8 xor + 8 v_bcnt + v_lshl_or + v_max + v_min + half a v_min3)
PMC_TRAFFIC = ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")     # committed rocprofv3 PMC summaries, newest first
PMC_VALU = ("r06_pmc_valu.json", "r05_pmc_valu.json", "r04_pmc_valu.json", "r03_pmc_valu.json", "r02_pmc_valu.json", "r01_pmc_valu.json")
MFMA_I8_PEAK_TOPS = 5000.0     # MI355X_MICROARCH.md: I8 at twice the bf16 rate (bf16 dense ~2.5 PF); v_mfma_i32_16x16x64_i8 measured 4.7 POPS
                               # (tools/ubench/mfma_i8.hip)
MATCH_OPS_PER_DISTANCE = 512   # 256 bit positions x (multiply + add): the matcher's distances as an int8 matrix product
MATCH_BYTES_PER_KP = 2 * (32 + 28) + 4      # k_match_pairs per keypoint of a pair: both frames' descriptor + keypoint record read once, match12 written


def make_frames(batch, seed, pad_w=0, pad_h=0):
    """`batch` frames cut from a few synthetic canvases as chains of <=8 px translations; pad_w / pad_h make
    them larger so that several distinct 1241x376 crops can be taken from every frame."""
    from ceres_mono_orb_slam2_amd import synth
    fams = ["blocks", "checker", "blocks", "checker", "blocks", "flat", "blocks", "checker"]
    ncanvas = min(len(fams), max(1, batch // 8))
    per = (batch + ncanvas - 1) // ncanvas
    out = []
    for c in range(ncanvas):
        fr, _ = synth.make_sequence(seed * 100 + c, W_IMG + pad_w, H_IMG + pad_h, per, fams[c], max_shift=8)
        out.append(fr)
    return np.concatenate(out)[:batch]


# ------------------------------------------------------------------------------------------ CPU baseline
def _ncores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def _cpu_frames_worker(frames, lo, n, out, k, deadline=None):
    from oracle import pyoracle as po
    E = po.OracleExtractor(NFEAT)
    prev = None
    done = 0
    for i in range(lo, lo + n):
        if deadline is not None and time.perf_counter() >= deadline:
            break
        kp, d = E.extract(frames[i % len(frames)])
        ref = prev if prev is not None else (kp, d)
        po.match_frames(d, kp["angle"], ref[1], ref[0]["angle"], 0.9, 50, True)
        prev = (kp, d)
        done += 1
    out[k] = done


def prepare_cpu_oracle(frames):
    """the CPU legs are timed on a `-O3 -march=native -ffp-contract=off` build of the oracle, compiled on THIS
    host (oracle/pyoracle.build_native) and used only after it reproduced the canonical (-march=x86-64-v2) build bit for bit
    on two of the bench's frames (keypoints, descriptors, matches) and on a small LocalBA (poses, points, flags)."""
    from oracle import pyoracle as po
    from ceres_mono_orb_slam2_amd import synth
    info = {"flags": po.CANONICAL_FLAGS, "native": False}
    po.use_library(None)
    so = po.build_native()
    if so is None:
        info["note"] = "native build failed on this host: canonical build timed"
        return info
    g = synth.make_ba_graph(2, ncam=8, npts=200, nobs=900, n_fixed=1)
    a = (g["K4"], g["poses0"], g["cam_fixed"], np.ones(8, np.uint8), g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])

    def probe():
        E = po.OracleExtractor(NFEAT)
        k0, d0 = E.extract(frames[0]); k1, d1 = E.extract(frames[1])
        m, nm = po.match_frames(d1, k1["angle"], d0, k0["angle"], 0.9, 50, True)
        rc, poses, pts, erase, _, s2 = po.local_ba(*a)
        return [k0.tobytes(), d0.tobytes(), k1.tobytes(), d1.tobytes(), m.tobytes(), int(nm),
                np.asarray(poses).tobytes(), np.asarray(pts).tobytes(), np.asarray(erase).tobytes(), int(s2["iterations"])]
    ref = probe()
    po.use_library(so)
    same = probe() == ref
    if not same:
        po.use_library(None)
        info["note"] = "native build is NOT bit-identical to the canonical one on this host: canonical build timed"
        return info
    info.update({"flags": po.NATIVE_FLAGS, "native": True, "library": so,
                 "bit_identical_to_canonical": "2 frames (keypoints, descriptors, matches) + 1 small LocalBA (poses, points, erase flags, iterations)"})
    return info


def cpu_baseline(frames, n_sample, all_seconds):
    """Oracle (CPU port of the reference algorithm) on a bounded sample of the same workload: one thread (the reference's
    own threading for extract / match) and all host cores running independent frame sub-sequences for `all_seconds`
    (throughput-fair).  `cores` of the all-core leg = CPU seconds consumed / wall seconds (a container may be allowed
    fewer cores than its affinity mask shows)."""
    from oracle import pyoracle as po
    po.lib()
    n = min(n_sample, len(frames))
    res = [0]
    t0 = time.perf_counter()
    _cpu_frames_worker(frames, 0, n, res, 0)
    dt1 = time.perf_counter() - t0
    nthr = min(_ncores(), 64)
    res = [0] * nthr
    deadline = time.perf_counter() + all_seconds
    ths = [threading.Thread(target=_cpu_frames_worker, args=(frames, c * 16, 1 << 30, res, c, deadline)) for c in range(nthr)]
    c0, t0 = time.process_time(), time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dta = time.perf_counter() - t0
    used = (time.process_time() - c0) / dta
    return {"value": n / dt1, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d of the bench's 1241x376 frames: oracle extract (2000 features) + 1 brute-force match each, "
                      "single thread (the reference's threading for this path), %.1f s" % (n, dt1),
            "all_cores": {"value": sum(res) / dta, "unit": "frames/s", "cores": max(1, int(round(used))), "threads": nthr,
                          "sample": "%d frames as %d independent sub-sequences (one oracle instance per thread) in %.1f s; "
                                    "cores = CPU seconds / wall seconds" % (sum(res), nthr, dta)},
            "host_cores": _ncores()}


# ------------------------------------------------------------------------------------------ launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_devices():
    """HIP devices visible to this process, counted by the library (no torch import in the launcher)."""
    try:
        sys.path.insert(0, ROOT)
        from ceres_mono_orb_slam2_amd import _lib
        return int(_lib.load().orbhip_device_count())
    except Exception:
        return 0


def numa_cpus_of_rank(local_rank, n_local):
    """CPUs a rank's host threads (the 8 LocalBA threads, the structure passes) are pinned to: the rank's share of the CPUs this
    process may run on, contiguous - on a two-socket node with GPUs 0-3 on socket 0 and 4-7 on socket 1 that keeps a rank's threads
    on its GPU's NUMA node (contiguous CPU numbering per socket)."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    if n_local <= 1 or len(cpus) < n_local:
        return None
    per = len(cpus) // n_local
    return cpus[local_rank * per:(local_rank + 1) * per]


def worker_envs(n, port, base=None):
    """Environment of each of the N ranks the launcher starts (one process per GPU; LOCAL_RANK selects the device)."""
    envs = []
    for r in range(n):
        e = dict(os.environ if base is None else base)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                  "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "ORBHIP_BENCH_WORKER": "1"})
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        envs.append(e)
    return envs


def launch(argv, n):
    """`python bench.py --gpus N` outside torchrun: start N ranks of this script, pass rank 0's stdout through.  Fails fast - before
    any rank is started - when fewer than N devices are visible (ORBHIP_BENCH_SHARED_GPU=1: the dry run on one GPU; --launch-check:
    no GPU needed)."""
    if os.environ.get("ORBHIP_BENCH_SHARED_GPU") != "1" and "--launch-check" not in argv:
        nvis = visible_devices()
        if nvis < n:
            sys.stderr.write("bench.py --gpus %d: only %d HIP device(s) visible (ORBHIP_BENCH_SHARED_GPU=1 runs the N > 1 code path on one GPU as a dry run)\n" % (n, nvis))
            return 2
    envs = worker_envs(n, _free_port())
    procs = []
    for r, e in enumerate(envs):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    return rc


def launch_check(rank, world):
    """--launch-check: rendezvous only (gloo, CPU): proves the launcher built `world` ranks.  No GPU needed."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([rank], dtype=torch.int64)
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        ranks = [int(g.item()) for g in got]
        dist.barrier()
        dist.destroy_process_group()
    else:
        ranks = [0]
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks": ranks,
                          "local_rank_env": os.environ.get("LOCAL_RANK", "0")}))


def pcie_pipeline(torch, dev, ex, mt, frames_host, B, resident_fps, n_batches=24):
    """Host-fed figure (never `value`): frames start in PINNED HOST memory and keypoints + descriptors + matches end there
    (what a drop-in operator() fed by src/Frame.cc:116 sees).  Upload of batch m + 1, extract + match of batch m and download of
    batch m - 1 overlap on three HIP streams (triple-buffered device inputs / outputs).  Round 5: the stage hand-offs are made on
    the HOST - an upload, a run and a download thread wait for each other's events with hipEventSynchronize; no stream ever holds a
    cross-stream dependency.  The earlier form (one host thread, hipStreamWaitEvent between the streams) ran at 0.93 of its bound
    for exactly one stream-to-hardware-queue mapping (GPU_MAX_HW_QUEUES = number of streams, created in one order) and at 0.42 - 0.55
    for every other one (tools/hostfed_sweep.py, 30 settings: queue counts 2 ... 8, creation order, idle extra streams) - upload,
    kernels and download simply ran one after the other; S independent upload -> run -> download chains gave 0.65 - 0.79, copies by
    the library's copy kernel 0.65 - 0.77.  This form: 0.87 - 0.91 for 4 and 8 queues, with or without idle extra streams.
    Also measures the bare H2D / D2H rates of the same buffers, which bound the pipeline."""
    import queue as _q
    NBUF = 3
    cap = ex.max_keypoints
    H, W = frames_host.shape[1:]
    pin_in = [torch.from_numpy(np.roll(frames_host, k, axis=0).copy()).pin_memory() for k in range(2)]
    din = [torch.empty((B, H, W), dtype=torch.uint8, device=dev) for _ in range(NBUF)]
    mk = lambda: (torch.empty((B, cap, 7), dtype=torch.float32, device=dev), torch.empty((B, cap, 32), dtype=torch.uint8, device=dev),
                  torch.empty((B,), dtype=torch.int32, device=dev), torch.empty((B, cap), dtype=torch.int32, device=dev),
                  torch.empty((B,), dtype=torch.int32, device=dev))
    dout = [mk() for _ in range(NBUF)]
    hout = [tuple(torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in dout[0]) for _ in range(2)]
    pa = torch.arange(B, dtype=torch.int32, device=dev); pb = (pa + B - 1) % B
    s_up, s_run, s_dn = (torch.cuda.Stream(device=dev) for _ in range(3))
    in_bytes = B * H * W
    out_bytes = sum(t.numel() * t.element_size() for t in dout[0])

    def rate(fn, nbytes, reps=4, trials=4):
        best = 0.0
        for _ in range(trials):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            best = max(best, nbytes * reps / (time.perf_counter() - t0) / 1e9)
        return best
    h2d = rate(lambda: din[0].copy_(pin_in[0], non_blocking=True), in_bytes)
    d2h = rate(lambda: [h.copy_(d, non_blocking=True) for h, d in zip(hout[0], dout[0])], out_bytes)

    def run(n):
        # hand-offs: queues of (batch index, event); free-lists bound the buffers in flight
        q_up, q_run = _q.Queue(), _q.Queue()
        free_in, free_out = _q.Queue(), _q.Queue()
        for b in range(NBUF):
            free_in.put((b, None)); free_out.put((b, None))
        err = []

        def uploader():
            try:
                torch.cuda.set_device(dev)
                for m in range(n):
                    b, ev = free_in.get()
                    if ev is not None: ev.synchronize()                  # the extractor has finished reading this input buffer
                    with torch.cuda.stream(s_up):
                        din[b].copy_(pin_in[m % 2], non_blocking=True)
                        e = torch.cuda.Event(); e.record(s_up)
                    q_up.put((m, b, e))
            except Exception as ex_:
                err.append(repr(ex_)); q_up.put(None)

        def runner():
            try:
                torch.cuda.set_device(dev)
                for m in range(n):
                    it = q_up.get()
                    if it is None: break
                    _, b, e = it
                    ob, oev = free_out.get()
                    e.synchronize()
                    if oev is not None: oev.synchronize()                # output buffers downloaded
                    with torch.cuda.stream(s_run):
                        ex.extract_batch(din[b], out=dout[ob][:3])
                        e_in = torch.cuda.Event(); e_in.record(s_run)
                        mt.match_frames_batch(dout[ob][0], dout[ob][1], dout[ob][2], pa, pb, out=dout[ob][3:])
                        e_out = torch.cuda.Event(); e_out.record(s_run)
                    free_in.put((b, e_in))
                    q_run.put((m, ob, e_out))
            except Exception as ex_:
                err.append(repr(ex_))
            q_run.put(None)

        def downloader():
            try:
                torch.cuda.set_device(dev)
                while True:
                    it = q_run.get()
                    if it is None: break
                    m, ob, e = it
                    e.synchronize()
                    with torch.cuda.stream(s_dn):
                        for h, d in zip(hout[m % 2], dout[ob]):
                            h.copy_(d, non_blocking=True)
                        e2 = torch.cuda.Event(); e2.record(s_dn)
                    free_out.put((ob, e2))
            except Exception as ex_:
                err.append(repr(ex_))
        ths = [threading.Thread(target=f) for f in (uploader, runner, downloader)]
        for t in ths: t.start()
        for t in ths: t.join()
        torch.cuda.synchronize()
        if err: raise RuntimeError("; ".join(err))
    run(NBUF)
    t0 = time.perf_counter()
    run(n_batches)
    dt = time.perf_counter() - t0
    fps = B * n_batches / dt
    k, d, c = ex.extract_batch(din[(n_batches - 1) % NBUF]); torch.cuda.synchronize()
    same = bool(torch.equal(hout[(n_batches - 1) % 2][2], c.cpu()))
    bound = min(h2d * 1e9 / (in_bytes / B), d2h * 1e9 / (out_bytes / B), resident_fps)
    return {"value": fps, "unit": "frames/s", "batches": n_batches, "ms_per_batch": dt / n_batches * 1e3, "form": "three host threads, host-side hand-offs",
            "h2d_GBps": h2d, "d2h_GBps": d2h, "h2d_bytes_per_frame": in_bytes // B, "d2h_bytes_per_frame": out_bytes // B,
            "pipeline_h2d_GBps": fps * in_bytes / B / 1e9, "pipeline_d2h_GBps": fps * out_bytes / B / 1e9,
            "bound_frames_per_s": bound, "frac_of_bound": fps / bound, "downloaded_counts_equal_resident_run": same,
            "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)")}


def rccl_summary(path):
    """A few facts out of RCCL's NCCL_DEBUG=INFO log of this rank: the communicator line (rank / nranks / device), how many channels
    it built and over which transports the peers are reached."""
    import re
    out = {"log": path}
    try:
        txt = open(path, errors="replace").read()
    except OSError as e:
        out["error"] = repr(e); return out
    m = re.findall(r"comm \S+ rank (\d+) nranks (\d+) cudaDev (\d+)[^\n]*", txt)
    if m: out["rank"], out["nranks"], out["device"] = (int(v) for v in m[-1])
    init = [l.strip() for l in txt.splitlines() if "Init COMPLETE" in l]
    if init: out["init_complete"] = init[-1][-200:]
    out["channel_lines"] = sum(1 for l in txt.splitlines() if re.search(r"Channel \d+", l))
    via = re.findall(r"via (\S+)", txt)
    out["transports"] = {k: via.count(k) for k in sorted(set(via))}
    ver = re.search(r"(RCCL version [^\n]+|NCCL version [^\n]+)", txt)
    if ver: out["version"] = ver.group(1).strip()
    return out


def run_ba_leg(local_rank, rank, cpu, oracle_lib, quick=False):
    """The LocalBA / PoseOptimization / GlobalBA legs (bench_ba.py) in a process of their own, on this
    rank's device; returns bench_ba.run's dictionary (with `_final_points`)."""
    import tempfile
    env = dict(os.environ)
    env.setdefault("GPU_MAX_HW_QUEUES", BA_LEG_HW_QUEUES)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):     # (not a rank of the job: no rendezvous)
        env.pop(k, None)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    with tempfile.TemporaryDirectory(prefix="orbhip_ba_") as d:
        cmd = [sys.executable, os.path.join(ROOT, "bench_ba.py"), "--device", str(local_rank), "--rank", str(rank),
               "--cpu", "1" if cpu else "0", "--oracle-lib", oracle_lib or "", "--quick", "1" if quick else "0", "--out", d]
        p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if p.returncode != 0:
            raise RuntimeError("BA leg failed (rc %d): %s" % (p.returncode, (p.stderr or p.stdout)[-1500:]))
        res = json.load(open(os.path.join(d, "result.json")))
        res["_final_points"] = np.load(os.path.join(d, "final_points.npy"))
    return res


# ------------------------------------------------------------------------------------------ the benchmark
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per extract/match launch sequence")
    ap.add_argument("--batches-per-step", type=int, default=32, help="batches per step (distinct resident frames)")
    ap.add_argument("--cpu-sample", type=int, default=96)    # ~8 s of single-thread oracle work
    ap.add_argument("--cpu-all-seconds", type=float, default=8.0)
    ap.add_argument("--streams", type=int, default=2, help="pipeline of the timed pass: batch m runs on HIP stream m %% S with its own extractor context, so the "
                    "latency-bound octree of one batch overlaps the issue-bound kernels of another.  Same-box sweeps at the end of round 4 (one stream "
                    "130.3 k frames/s): S = 2 -> 148.1 k, 3 -> 141.1 k, 4 -> 131.1 k, 5 -> 135.5 k, 7 -> 126.7 k, the same for GPU_MAX_HW_QUEUES 4 ... 24.  "
                    "Streams that merely EXIST count too (four contexts alive, two in use: 115 k; contexts re-created after a trial run: 113 k) - the "
                    "hardware queues of a process share a few dispatch pipes -, which is why the count is a fixed default and not tuned at run time.  "
                    "1 = the timed pass on one stream")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the secondary pipelined figure (only with --streams 1)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-fed (PCIe-inclusive) figure")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--ba-quick", action="store_true", help="shortened BA legs (tests of the N > 1 plumbing; not benchmark figures)")
    ap.add_argument("--launch-check", action="store_true", help="rendezvous of the N ranks only (CPU, gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch(sys.argv[1:], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but %d ranks were launched (WORLD_SIZE)" % (args.gpus, world)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.launch_check:
        return launch_check(rank, world)
    nccl_log_dir = None
    pinned_cpus = None
    if world > 1:
        pinned_cpus = numa_cpus_of_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
        if pinned_cpus:
            try:
                os.sched_setaffinity(0, pinned_cpus)
            except OSError:
                pinned_cpus = None

    # ---- the BA legs run in a process of their own (12 hardware queues) and go FIRST, before this process creates its HIP context
    shared = os.environ.get("ORBHIP_BENCH_SHARED_GPU") == "1"
    B, M = args.batch, args.batches_per_step
    nbase = min(2, M)
    nvar = (M + nbase - 1) // nbase
    base_host = [make_frames(B, seed=rank * 16 + b, pad_w=3 * (nvar - 1), pad_h=2 * (nvar - 1)) for b in range(nbase)]
    frames_host = np.ascontiguousarray(base_host[0][:, :H_IMG, :W_IMG])          # = batches[0] below
    dev_ord = 0 if shared else local_rank
    oracle_build = None
    if not args.no_cpu and world == 1 and rank == 0:
        try:
            oracle_build = prepare_cpu_oracle(frames_host)
        except Exception as e:
            oracle_build = {"error": repr(e)}
    localba, ba_ok = None, 1
    if not args.no_ba:
        try:
            nat = oracle_build.get("library") if isinstance(oracle_build, dict) and oracle_build.get("native") else None
            localba = run_ba_leg(dev_ord, rank, (not args.no_cpu) and world == 1, nat, quick=args.ba_quick)
        except Exception as e:                       # never lose the headline line to the secondary leg
            localba, ba_ok = {"error": repr(e)}, 0

    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    # ORBHIP_BENCH_SHARED_GPU=1: dry run of the N > 1 code path on a box with fewer GPUs than ranks (every rank uses
    # device 0, gloo carries the collectives on host tensors); the real multi-GPU run is one rank per GPU over RCCL
    ndev = torch.cuda.device_count()
    if shared:
        local_rank = 0
    assert local_rank < ndev, ("rank %d needs GPU %d but only %d visible (ORBHIP_BENCH_SHARED_GPU=1 runs the N > 1 path "
                               "on one GPU as a dry run)" % (rank, local_rank, ndev))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if shared else dev          # where collective payloads live
    backend = None
    # ORBHIP_BENCH_FORCE_DIST=1: build the process group (RCCL) and run the landmark all-gather even at world size 1 - the
    # one-GPU box's only way to execute the N > 1 code path on a real RCCL communicator (tests/test_gpu_rccl.py)
    use_dist = world > 1 or os.environ.get("ORBHIP_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = "gloo" if shared else "nccl"
        if not shared and "NCCL_DEBUG_FILE" not in os.environ and os.environ.get("NCCL_DEBUG", "").upper() not in ("TRACE",):
            # RCCL's own account of the communicator (ranks, channels, transports) goes to a file per rank; rank 0 summarises it into
            # `collective.rccl` so that the first real N-GPU run shows its topology at a glance
            import tempfile
            nccl_log_dir = tempfile.mkdtemp(prefix="orbhip_rccl_")
            os.environ["NCCL_DEBUG"] = "INFO"; os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,GRAPH"
            os.environ["NCCL_DEBUG_FILE"] = os.path.join(nccl_log_dir, "rccl_rank%d.log" % rank)
        if shared:
            # (gloo's C++ side prints "[Gloo] Rank r is connected to ..." on STDOUT: rank 0's stdout must carry the JSON line only)
            sys.stdout.flush(); saved = os.dup(1); os.dup2(2, 1)
            try:
                dist.init_process_group("gloo", rank=rank, world_size=world)
                dist.barrier()
            finally:
                sys.stdout.flush(); os.dup2(saved, 1); os.close(saved)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        assert dist.get_world_size() == args.gpus

    from ceres_mono_orb_slam2_amd import ORBextractor, ORBmatcher, _lib, sharding
    _lib.check(_lib.load().orbhip_set_default_device(local_rank), "orbhip_set_default_device")
    # M distinct batches resident in HBM: two synthetic base batches per rank (its own frames: frame sharding), generated
    # a little larger than 1241x376 (base_host above), and crops of them at offsets (3v, 2v) px: a crop keeps the <=8 px frame-to-frame
    # chains and moves every corner relative to the cell grid, so no two batches repeat work
    base = [torch.from_numpy(b).to(dev) for b in base_host]
    batches = []
    for m in range(M):
        v = m // nbase
        batches.append(base[m % nbase][:, 2 * v:2 * v + H_IMG, 3 * v:3 * v + W_IMG].contiguous())
    del base, base_host
    S = max(1, args.streams)
    exs = [ORBextractor(NFEAT, 1.2, 8, 20, 7, device=local_rank) for _ in range(S)]
    ex = exs[0]
    mt = ORBmatcher(0.9, True)
    cap = ex.max_keypoints
    kps = torch.empty((B, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    match12 = torch.empty((B, cap), dtype=torch.int32, device=dev)
    nmatch = torch.empty((B,), dtype=torch.int32, device=dev)
    pair_a = torch.arange(B, dtype=torch.int32, device=dev)
    pair_b = (pair_a + B - 1) % B                      # frame i against its predecessor
    kp_sum = torch.zeros((), dtype=torch.int64, device=dev)
    nm_sum = torch.zeros((), dtype=torch.int64, device=dev)
    bad = torch.zeros((), dtype=torch.int64, device=dev)

    outs = [(kps, desc, counts, match12, nmatch)] + [tuple(torch.empty_like(t) for t in (kps, desc, counts, match12, nmatch)) for _ in range(S - 1)]
    side = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else None

    def one(m, k, events):
        o = outs[k]
        exs[k].extract_batch(batches[m], out=o[:3])
        if events is not None:
            e0 = torch.cuda.Event(enable_timing=True); e0.record()
        mt.match_frames_batch(o[0], o[1], o[2], pair_a, pair_b, out=o[3:])
        if events is not None:
            e1 = torch.cuda.Event(enable_timing=True); e1.record()
            events.append((e0, e1))

    def step(events=None):
        for m in range(M):
            if S == 1:
                one(m, 0, events)
            else:
                with torch.cuda.stream(side[m % S]):
                    one(m, m % S, events)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- pass 1 (not `value`): ONE stream, stage events on - the kernels of a batch run alone (the blur beside FAST + octree on the
    #      extractor's side stream, as always), so their durations are exclusive and add up along the critical path; this pass
    #      yields the per-kernel times of `roofline` / `kernels` and the one-stream rate
    def step_one(events=None):
        for m in range(M):
            one(m, 0, events)
    for _ in range(max(1, min(args.warmup, 2))):
        step_one()
    barrier()
    ex.set_profiling(True)
    ev = []
    k1 = max(2, min(args.steps, 5))
    t1 = time.perf_counter()
    for k in range(k1):
        step_one(ev)
    barrier()
    dt1 = time.perf_counter() - t1
    stage_ms, ncalls = ex.stage_ms()
    ex.set_profiling(False)
    match_ms = sum(a.elapsed_time(b) for a, b in ev)
    dt1 = sharding.max_over_ranks(dt1, device=cdev)
    one_stream = {"value": B * M * world * k1 / dt1, "unit": "frames/s", "steps": k1, "ms_per_step": dt1 / k1 * 1e3,
                  "note": "the same frames and kernels on ONE stream (pass 1): the per-kernel times of `roofline` and `kernels` come from here"}
    # ---- pass 2, the timed configuration: batch m on stream m % S with its own extractor context (default S = 2).  The kernels of
    #      consecutive batches overlap - the matcher's matrix-core products and the latency-bound octree of one batch run beside the
    #      VALU-issue-bound pyramid / FAST / blur kernels of the next - every batch is extracted AND matched
    #      inside the timed region, exactly K steps between barrier + synchronize
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    dt = sharding.max_over_ranks(dt, device=cdev)

    # sanity of the timed work (not timed): one more pass, every frame of every batch produced keypoints and matches
    for m in range(M):
        ex.extract_batch(batches[m], out=(kps, desc, counts))
        mt.match_frames_batch(kps, desc, counts, pair_a, pair_b, out=(match12, nmatch))
        kp_sum += counts.sum(); nm_sum += nmatch.sum(); bad += (counts <= 0).sum()
    torch.cuda.synchronize()
    assert int(bad.item()) == 0, "extractor produced an empty/overflowed frame"
    # ... and, when the CPU oracle is in play anyway (cpu_baseline), the LAST batch of that pass against it: two frames' keypoints
    # and descriptors and their match list, bit for bit (the checker, not the thing measured)
    oracle_check = None
    if not args.no_cpu and world == 1 and rank == 0:
        try:
            from oracle import pyoracle as po
            po.use_library(None)
            last = batches[M - 1][B - 2:B].cpu().numpy()
            E = po.OracleExtractor(NFEAT)
            ref = [E.extract(last[i]) for i in range(2)]
            cn = counts[B - 2:B].cpu().numpy()
            same = True
            for i in range(2):
                n = int(cn[i])
                same = same and n == len(ref[i][0]) and np.array_equal(kps[B - 2 + i, :n].cpu().numpy().view(np.uint8).reshape(n, 28), ref[i][0].view(np.uint8).reshape(n, 28)) \
                    and np.array_equal(desc[B - 2 + i, :n].cpu().numpy(), ref[i][1])
            om, on = po.match_frames(ref[1][1], ref[1][0]["angle"], ref[0][1], ref[0][0]["angle"], 0.9, 50, True)
            same = same and int(nmatch[B - 1].item()) == on and np.array_equal(match12[B - 1, :int(cn[1])].cpu().numpy(), om)
            oracle_check = {"frames": 2, "pairs": 1, "bit_identical_to_oracle": bool(same),
                            "what": "frames %d and %d of the last batch of the untimed sanity pass: keypoints, descriptors, match list" % (B - 2, B - 1)}
            assert same, "bench.py: the extractor / matcher output of the benchmarked configuration differs from the oracle"
        except ImportError as e:
            oracle_check = {"error": repr(e)}
    mean_kp, mean_match = float(kp_sum.item()) / (B * M), float(nm_sum.item()) / (B * M)

    # ---- secondary figure (never `value`): the same work pipelined over 2 HIP streams / extractor contexts, rank 0 only, a
    #      few steps.  The default run keeps ONE stream so that the per-kernel event times are exclusive and add up to the step.
    pipelined = None
    if S == 1 and world == 1 and not args.no_pipelined:
        try:
            S3 = 2
            ex3 = exs + [ORBextractor(NFEAT, 1.2, 8, 20, 7, device=local_rank) for _ in range(S3 - 1)]
            out3 = outs + [tuple(torch.empty_like(t) for t in (kps, desc, counts, match12, nmatch)) for _ in range(S3 - 1)]
            st3 = [torch.cuda.Stream(device=dev) for _ in range(S3)]

            def step3():
                for m in range(M):
                    k = m % S3
                    with torch.cuda.stream(st3[k]):
                        ex3[k].extract_batch(batches[m], out=out3[k][:3])
                        mt.match_frames_batch(out3[k][0], out3[k][1], out3[k][2], pair_a, pair_b, out=out3[k][3:])
            step3(); torch.cuda.synchronize()
            k3 = max(2, args.steps // 4)
            t3 = time.perf_counter()
            for _ in range(k3):
                step3()
            torch.cuda.synchronize()
            d3 = time.perf_counter() - t3
            same = all(bool(torch.equal(out3[k][4], out3[0][4])) for k in range(1, S3)) if M % S3 == 0 else None
            pipelined = {"streams": S3, "value": B * M * k3 / d3, "unit": "frames/s", "steps": k3, "ms_per_step": d3 / k3 * 1e3,
                         "note": "same frames and kernels, batch m on stream m % 2 with its own extractor context (bench.py --streams 2 "
                                 "makes it the timed configuration); kernel times then overlap, so the roofline above is taken from the "
                                 "one-stream run"}
            del ex3, out3
        except Exception as e:                       # never lose the headline line to a secondary figure
            pipelined = {"error": repr(e)}

    # ---- secondary figure (never `value`): the match leg at the size the metric names, 2000 x 2000 per pair (the timed frames keep
    #      1816 keypoints on average: one canvas in eight is SURVEY 8(d)'s low-texture family, which stays below its quota)
    match_full = None
    if world == 1 and cap >= 2000:
        try:
            g = torch.Generator(device=dev); g.manual_seed(5)
            d2k = torch.randint(0, 256, (B, cap, 32), dtype=torch.uint8, device=dev, generator=g)
            k2k = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
            k2k[:, :, 3] = torch.rand((B, cap), device=dev, generator=g) * 360.0
            c2k = torch.full((B,), 2000, dtype=torch.int32, device=dev)
            mo = (torch.empty_like(match12), torch.empty_like(nmatch))
            for _ in range(5): mt.match_frames_batch(k2k, d2k, c2k, pair_a, pair_b, out=mo)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): mt.match_frames_batch(k2k, d2k, c2k, pair_a, pair_b, out=mo)
            e1.record(); torch.cuda.synchronize()
            ms2k = e0.elapsed_time(e1) / 20
            match_full = {"keypoints_per_frame": 2000, "pairs_per_launch": B, "ms_per_launch": ms2k, "distances_per_launch": 4.0e6 * B,
                          "int8_Tops": 4.0e6 * B * MATCH_OPS_PER_DISTANCE / (ms2k * 1e-3) / 1e12,
                          "note": "random descriptors, every frame at the full 2000 keypoints; the brute-force pass does not depend on the data"}
            del d2k, k2k, c2k, mo
        except Exception as e:
            match_full = {"error": repr(e)}

    # ---- secondary figure (never `value`): the host-fed pipeline, rank 0 of a one-GPU run only
    pcie = None
    if world == 1 and not args.no_pcie:
        try:
            pcie = pcie_pipeline(torch, dev, ex, mt, frames_host, B, B * M * world * args.steps / dt)
        except Exception as e:
            pcie = {"error": repr(e)}

    # ---- LocalBA / PoseOptimization / GlobalBA legs: every rank solves its own independent problems (sub-map sharding,
    #      no collective in the solve); for N > 1 the landmark updates are merged with ONE all-gather.
    #      (the legs themselves ran first, above)
    collective = None
    if not args.no_ba:
        ok = ba_ok
        if use_dist:
            flag = torch.tensor([ok], dtype=torch.int32, device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # collectives below only if EVERY rank's leg succeeded
            if int(flag.item()) == 1:
                keys = ["localba_solves_per_s", "localba_single_stream_solves_per_s", "poseopt_solves_per_s"]
                t = torch.tensor([localba[k] for k in keys], dtype=torch.float64, device=cdev)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                for k, v in zip(keys, t.tolist()):
                    localba[k] = v
                pts = torch.from_numpy(localba["_final_points"]).to(cdev)
                allp, _, cnts = sharding.allgather_landmarks(pts)          # warm-up (communicator setup)
                barrier(); tg = time.perf_counter()
                allp, _, cnts = sharding.allgather_landmarks(pts)
                torch.cuda.synchronize()
                dump = os.environ.get("ORBHIP_BENCH_DUMP_DIR")          # (tests: every rank's own points and rank 0's merged array)
                if dump:
                    np.save(os.path.join(dump, "final_points_rank%d.npy" % rank), localba["_final_points"])
                    if rank == 0: np.save(os.path.join(dump, "merged_points.npy"), allp.cpu().numpy())
                collective = {"backend": backend + (" (RCCL)" if backend == "nccl" else " (shared-GPU dry run)"),
                              "world_size": dist.get_world_size(), "op": "all_gather of every rank's GlobalBA landmarks",
                              "landmark_allgather_ms": (time.perf_counter() - tg) * 1e3,
                              "points_per_rank": cnts, "merged_points": int(allp.shape[0]),
                              "payload_bytes": int(allp.shape[0]) * 24}
                if nccl_log_dir:
                    collective["rccl"] = rccl_summary(os.path.join(nccl_log_dir, "rccl_rank%d.log" % rank))
                # the same merge through the C ABI (orbhip_comm_create / orbhip_allgather_landmarks: RCCL directly, what a C++ embedding
                # of the reference calls) - on real RCCL communicators only (two ranks cannot share a GPU there); never fatal to the line
                if backend == "nccl" and os.environ.get("ORBHIP_BENCH_CABI_MERGE", "1") != "0":
                    try:
                        lc = sharding.LandmarkCommunicator(rank, dist.get_world_size(), dev_ord)
                        cap_t = torch.tensor([pts.shape[0]], dtype=torch.int64, device=cdev)
                        dist.all_reduce(cap_t, op=dist.ReduceOp.MAX)
                        cap_r = int(cap_t.item())
                        lc.allgather_landmarks(pts, cap_per_rank=cap_r)                  # warm-up
                        barrier(); tc = time.perf_counter()
                        cp, _, ccnt = lc.allgather_landmarks(pts, cap_per_rank=cap_r)
                        tcabi = (time.perf_counter() - tc) * 1e3
                        collective["c_abi"] = {"entry_point": "orbhip_allgather_landmarks (one ncclAllGather of fixed-size slots)", "ms": tcabi,
                                               "points_per_rank": [int(x) for x in ccnt], "equals_torch_distributed": bool(torch.equal(cp, allp))}
                        lc.close()
                    except Exception as e:
                        collective["c_abi"] = {"error": repr(e)}
        if isinstance(localba, dict):
            localba.pop("_final_points", None)
    if rank == 0:
        K = args.steps
        fps = B * M * world * K / dt
        per_call = {k: v / max(ncalls, 1) for k, v in stage_ms.items()}
        per_call["match"] = match_ms / max(len(ev), 1)
        # the dominant kernel = the longest launch of the step, the matcher included (its bytes: both frames' records once)
        bytes_of = dict(BYTES); bytes_of["match"] = int(round(mean_kp * MATCH_BYTES_PER_KP)); bytes_of["octree"] = OCTREE_BYTES
        match_mfma = os.environ.get("ORBHIP_MATCH_MFMA", "1") != "0" and cap <= 4080        # (the library's own switch and limit)
        kernel_of = dict(KERNEL_OF); kernel_of["match"] = "k_match_pairs_mfma" if match_mfma else "k_match_pairs"
        # batches of >= 8 frames run k_blur7 on the extractor's side stream beside FAST + octree (ORBHIP_OVERLAP_BLUR, default on):
        # its event duration is then NOT an exclusive time (it shares the chip), so it cannot be the "dominant kernel" of the
        # roofline; it is reported as what it is
        blur_concurrent = os.environ.get("ORBHIP_OVERLAP_BLUR", "1" if B >= 8 else "0") != "0"
        # The critical path of one batch on one stream: pyramid -> max(FAST + octree, blur on the side stream) -> describe -> match.
        # `roofline` prices the LONGEST KERNEL ON THAT PATH (k_blur7 when the blur leg is the longer one: its duration beside FAST +
        # octree is a concurrent one, and it is the time the path waits for), `critical_path` the whole path.
        main_leg = per_call["fast_cells"] + per_call["octree"]
        blur_leg = per_call["blur"] if blur_concurrent else 0.0
        on_path = {"pyramid": per_call["pyramid"], "describe": per_call["describe"], "match": per_call["match"]}
        if blur_concurrent and blur_leg >= main_leg:
            on_path["blur"] = blur_leg
        else:
            on_path["fast_cells"] = per_call["fast_cells"]
            on_path["octree"] = per_call["octree"]               # (round 5, VERDICT r4 weak #5: k_octree is on the path and may be its longest kernel)
            if not blur_concurrent: on_path["blur"] = per_call["blur"]
        crit_ms = per_call["pyramid"] + (max(main_leg, blur_leg) if blur_concurrent else main_leg + per_call["blur"]) + per_call["describe"] + per_call["match"]
        stages = dict(on_path)
        dom = max(stages, key=stages.get)
        hbm_stages = {k: v for k, v in on_path.items() if k not in ("match", "octree")}
        hdom = max(hbm_stages, key=hbm_stages.get)
        kname = kernel_of[dom]
        ach = bytes_of[dom] * B / (per_call[dom] * 1e-3) / 1e9
        # HBM traffic per launch from the COMMITTED rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, separate runs),
        # scaled to this batch -- not measured in this run (counters need rocprofv3); None if the summary is absent
        traffic, traffic_src = None, None
        for f in PMC_TRAFFIC:
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", f)))
                traffic, traffic_src = pmc["kernels"][kname]["hbm_bytes_per_frame"] * B, "profiles/" + f
                break
            except Exception:
                continue
        roof = {"kernel": kname, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": (traffic_src + " (committed PMC pass of this kernel, scaled to the batch; not collected in this run)") if traffic_src else None,
                "algorithmic_bytes_per_launch": bytes_of[dom] * B,
                "ms_per_launch": per_call[dom], "launches_timed": ncalls if dom != "match" else len(ev),
                "timing": "HIP events on the launch stream inside the timed region (orbx_set_profiling; the matcher: torch events on the same stream)"}
        # the kernel streams bytes but BINDS on the integer VALU issue rate: report that fraction too, from the
        # committed PMC pass (SQ_INSTS_VALU x 4 cycles / SIMD-cycles of the launch)
        for f in PMC_VALU:
            try:
                vp = json.load(open(os.path.join(ROOT, "profiles", f)))["kernels"][kname]
                roof["limiter"] = "valu_issue" if dom != "octree" else "latency (one workgroup per (frame, level), serial node splits)"
                roof["valu_issue_frac"] = vp["valu_busy_frac"]
                roof["valu_insts_per_wave"] = vp["valu_insts_per_wave"]
                for kk in ("wait_any_frac_of_wave_cycles", "avg_waves_per_simd"):
                    if kk in vp: roof[kk] = vp[kk]
                roof["valu_source"] = "profiles/" + f
                break
            except Exception:
                continue
        pairs_per_s = B / (per_call["match"] * 1e-3)
        lane_ops = mean_kp * mean_kp * MATCH_LANE_OPS_PER_PAIR
        match_valu = {"achieved": pairs_per_s * lane_ops / 1e12, "peak": VALU_PEAK_TOPS, "unit": "T lane-ops/s",
                      "frac": pairs_per_s * lane_ops / 1e12 / VALU_PEAK_TOPS,
                      "lane_ops_per_launch": lane_ops * B, "lane_ops_per_distance": MATCH_LANE_OPS_PER_PAIR,
                      "distances_per_pair": mean_kp * mean_kp}
        # on the matrix cores the matcher is an int8 product: 512 operations per distance against the dense I8 peak
        tops = pairs_per_s * mean_kp * mean_kp * MATCH_OPS_PER_DISTANCE / 1e12
        match_mfma_roof = {"kernel": "k_match_pairs_mfma", "bound": "mfma", "achieved": tops, "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s",
                           "frac": tops / MFMA_I8_PEAK_TOPS, "ops_per_launch": mean_kp * mean_kp * MATCH_OPS_PER_DISTANCE * B,
                           "ops_per_distance": MATCH_OPS_PER_DISTANCE, "distances_per_pair": mean_kp * mean_kp, "ms_per_launch": per_call["match"],
                           "note": "v_mfma_i32_16x16x64_i8, 4 instructions per 16 x 16 distances; the VALU kernel (ORBHIP_MATCH_MFMA=0) needs "
                                   "19.5 lane-instructions per distance"}
        if dom == "match":
            if match_mfma:
                roof.update({k: match_mfma_roof[k] for k in ("bound", "achieved", "peak", "unit", "frac")})
                roof["limiter"] = "mfma"; roof["mfma"] = match_mfma_roof
                roof.pop("valu_issue_frac", None); roof.pop("valu_insts_per_wave", None); roof.pop("valu_source", None)
            else:           # VALU-issue bound: its HBM fraction is tiny by construction
                roof["limiter"] = "valu_issue"
                roof["valu_issue"] = match_valu
                roof["valu_issue_frac"] = match_valu["frac"]
        # the longest HBM-streaming kernel keeps its own block (the contract's "hbm" roofline), with the PMC traffic
        hk = KERNEL_OF[hdom]
        hach = BYTES[hdom] * B / (per_call[hdom] * 1e-3) / 1e9
        hroof = {"kernel": hk, "bound": "hbm", "achieved": hach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hach / HBM_PEAK_GBS,
                 "algorithmic_bytes_per_launch": BYTES[hdom] * B, "ms_per_launch": per_call[hdom], "traffic": None}
        for f in PMC_TRAFFIC:
            try:
                hroof["traffic"] = json.load(open(os.path.join(ROOT, "profiles", f)))["kernels"][hk]["hbm_bytes_per_frame"] * B
                hroof["traffic_source"] = "profiles/" + f
                break
            except Exception:
                continue
        for f in PMC_VALU:
            try:
                vp = json.load(open(os.path.join(ROOT, "profiles", f)))["kernels"][hk]
                hroof.update({"limiter": "valu_issue", "valu_issue_frac": vp["valu_busy_frac"], "valu_insts_per_wave": vp["valu_insts_per_wave"],
                              "valu_source": "profiles/" + f})
                break
            except Exception:
                continue
        roof["hbm_stream_kernel"] = hroof
        fe_bytes0 = sum(BYTES.values()) + bytes_of["match"]
        roof["critical_path"] = {"legs_ms": {"pyramid": per_call["pyramid"], "fast_cells+octree": main_leg, "blur (side stream, concurrent)": blur_leg,
                                             "describe": per_call["describe"], "match": per_call["match"]},
                                 "longer_middle_leg": "k_blur7" if (blur_concurrent and blur_leg >= main_leg) else "k_fast_cells + k_octree",
                                 "ms_per_batch": crit_ms, "one_stream_ms_per_batch_measured": dt1 / k1 / M * 1e3,
                                 "algorithmic_GBps": fe_bytes0 * B / (crit_ms * 1e-3) / 1e9, "frac_of_hbm_peak": fe_bytes0 * B / (crit_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "note": "pyramid + max(FAST + octree, blur) + describe + match of one batch on one stream (pass 1); the timed configuration "
                                         "overlaps the paths of consecutive batches on %d streams" % S}
        if match_mfma: roof["match_kernel_mfma"] = match_mfma_roof
        else: roof["match_kernel_valu_issue"] = match_valu
        if blur_concurrent:
            roof["concurrent_kernels"] = {"k_blur7": {"ms_per_launch": per_call["blur"], "stream": "the extractor's side stream, beside k_fast_cells + k_octree",
                                                       "note": "a concurrent duration, not an exclusive one: alone the launch takes ~0.40 ms (ORBHIP_OVERLAP_BLUR=0, profiles/); when it is longer than FAST + octree it IS the middle leg of the critical path and `roofline` prices it"}}
        # whole front-end (extract + match of one frame) against both ceilings: algorithmic bytes / frame x frames/s / HBM peak,
        # and VALU lane-ops / frame (committed PMC pass: SQ_INSTS_VALU x 64 lanes of the five extract kernels, + the matcher's
        # counted instructions) x frames/s / issue peak
        fe_bytes = sum(BYTES.values()) + bytes_of["match"]
        fe = {"algorithmic_bytes_per_frame": fe_bytes, "hbm_GBps": fe_bytes * fps / world / 1e9, "hbm_frac": fe_bytes * fps / world / 1e9 / HBM_PEAK_GBS}
        for f in PMC_VALU:
            try:
                vj = json.load(open(os.path.join(ROOT, "profiles", f)))
                per_frame = sum(v["valu_insts"] for v in vj["kernels"].values() if "valu_insts" in v and not v.get("is_match")) * 64.0 / vj["frames_per_launch"]
                mj = vj["kernels"].get("k_match_pairs")
                mj = vj["kernels"].get("k_match_pairs_mfma") if match_mfma else mj
                m_ops = (mj["valu_insts"] * 64.0 / vj["frames_per_launch"]) if mj and mj.get("is_match") else (0.0 if match_mfma else lane_ops)
                fe.update({"valu_lane_ops_per_frame": per_frame + m_ops, "valu_Tops": (per_frame + m_ops) * fps / world / 1e12,
                           "valu_frac": (per_frame + m_ops) * fps / world / 1e12 / VALU_PEAK_TOPS, "valu_source": "profiles/" + f})
                break
            except Exception:
                continue
        roof["front_end"] = fe
        kernels = {k: {"ms_per_launch_batch": v} for k, v in per_call.items()}
        for k in list(BYTES) + ["octree"]:
            kernels[k]["algorithmic_GBps"] = bytes_of[k] * B / (per_call[k] * 1e-3) / 1e9
        if match_mfma:
            kernels["match"]["mfma_i8_Tops"] = tops
            kernels["match"]["mfma_frac_of_peak"] = tops / MFMA_I8_PEAK_TOPS
        else:
            kernels["match"]["valu_Tops"] = pairs_per_s * lane_ops / 1e12
            kernels["match"]["valu_frac_of_peak"] = pairs_per_s * lane_ops / 1e12 / VALU_PEAK_TOPS
        out = {
            "metric": "frames/sec ORB extract+match @1241x376 + LocalBA solves/sec; 1/2/4/8 GPU",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "KITTI 1241x376, 2000 features/frame requested (nfeatures = 2000; the octree keeps %.0f on average on these "
                                   "synthetic frames, so a match is %.0f x %.0f distances), 8 levels, ORB extract + brute-force Hamming "
                                   "match vs previous frame (ratio 0.9, TH_LOW 50, rotation histogram)" % (mean_kp, mean_kp, mean_kp),
                       "frames_per_gpu_per_step": B * M, "frames_per_launch": B, "batches_per_step": M,
                       "sharding": "frames across ranks, no collective in the data path", "streams": S,
                       "mean_keypoints": mean_kp, "mean_matches": mean_match},
            "roofline": roof,
            "kernels": kernels,
        }
        out["one_stream"] = one_stream
        if oracle_check is not None:
            out["oracle_check"] = oracle_check
        if pipelined is not None:
            out["pipelined"] = pipelined
        if pcie is not None:
            out["pcie_inclusive"] = pcie
        if match_full is not None:
            if "ms_per_launch" in match_full:     # what the step would take with every frame at its quota: only the match leg grows
                ms_step = dt1 / k1 / M * 1e3
                match_full["one_stream_frames_per_s_if_every_frame_had_2000"] = B / ((ms_step - per_call["match"] + match_full["ms_per_launch"]) * 1e-3)
            out["match_2000x2000"] = match_full
        if collective is not None:
            out["collective"] = collective
        if world > 1:
            out["host_threads_pinned_to_cpus"] = ("%d-%d (rank 0; every rank takes its contiguous share)" % (pinned_cpus[0], pinned_cpus[-1])) if pinned_cpus else None
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(frames_host, args.cpu_sample, args.cpu_all_seconds)
            out["cpu_baseline"]["build"] = oracle_build
        elif not args.no_cpu:
            # N > 1: the contract asks for the CPU baseline at N = 1 only; rank 0 still times a REDUCED sample (its share of the host
            # cores is a fraction of the box) so that the line carries the field - the figure to quote is the N = 1 run's
            from oracle import pyoracle as po
            po.use_library(None)
            out["cpu_baseline"] = cpu_baseline(frames_host, min(args.cpu_sample, 12), min(args.cpu_all_seconds, 2.0))
            out["cpu_baseline"]["note"] = "reduced sample on rank 0 of an N = %d run (canonical oracle build, this rank's share of the host cores): quote the N = 1 run" % world
        if localba is not None:
            if isinstance(localba.get("cpu_baseline"), dict):
                localba["cpu_baseline"]["build"] = oracle_build
            out["localba"] = localba
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
