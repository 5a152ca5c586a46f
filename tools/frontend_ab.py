"""Quick per-stage timing of the front-end on ONE 256-frame batch (no CPU legs, no BA): the A/B loop for kernel work.
usage: python tools/frontend_ab.py [reps=20] [B=256]      (prints one JSON line: ms per launch per stage + frames/s)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ceres_mono_orb_slam2_amd import ORBextractor, ORBmatcher
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
_fr = bench.make_frames(B, seed=0)
if os.environ.get("ORBHIP_AB_CONTRAST"):      # A/B aid: the same frames with their contrast scaled (0.25: most edges fall below iniTh = 20, as in low-texture real scenes) + fresh sigma-2 noise
    import numpy as np
    _c = float(os.environ["ORBHIP_AB_CONTRAST"]); _rng = np.random.default_rng(1)
    _fr = np.clip(np.rint(128.0 + (_fr.astype(np.float32) - 128.0) * _c + _rng.normal(0, 2.0, _fr.shape).astype(np.float32)), 0, 255).astype(np.uint8)
frames = torch.from_numpy(_fr).to(dev)
ex = ORBextractor(bench.NFEAT, 1.2, 8, 20, 7)
mt = ORBmatcher(0.9, True)
cap = ex.max_keypoints
kps = torch.empty((B, cap, 7), dtype=torch.float32, device=dev); desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=dev)
counts = torch.empty((B,), dtype=torch.int32, device=dev); match12 = torch.empty((B, cap), dtype=torch.int32, device=dev)
nmatch = torch.empty((B,), dtype=torch.int32, device=dev)
pa = torch.arange(B, dtype=torch.int32, device=dev); pb = (pa + B - 1) % B
for _ in range(3):
    ex.extract_batch(frames, out=(kps, desc, counts)); mt.match_frames_batch(kps, desc, counts, pa, pb, out=(match12, nmatch))
torch.cuda.synchronize()
ex.set_profiling(True)
ev = []
t0 = time.perf_counter()
for _ in range(reps):
    ex.extract_batch(frames, out=(kps, desc, counts))
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    mt.match_frames_batch(kps, desc, counts, pa, pb, out=(match12, nmatch))
    e1 = torch.cuda.Event(enable_timing=True); e1.record(); ev.append((e0, e1))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st, n = ex.stage_ms()
out = {k: round(v / max(n, 1), 4) for k, v in st.items()}
out["match"] = round(sum(a.elapsed_time(b) for a, b in ev) / reps, 4)
out["frames_per_s"] = round(B * reps / dt, 1)
out["checksum"] = [int(counts.sum().item()), int(nmatch.sum().item()), int(desc.to(torch.int64).sum().item()), int(match12.to(torch.int64).sum().item())]
print(json.dumps(out))
# ---- two-stream pipeline (opt-in in production): batch m on stream m % 2 with its own extractor context and outputs, so that
#      the latency-bound octree of one batch runs beside the VALU-bound kernels of the other
if len(sys.argv) > 3:
    S = int(sys.argv[3])
    exs = [ex] + [ORBextractor(bench.NFEAT, 1.2, 8, 20, 7) for _ in range(S - 1)]
    outs = [(kps, desc, counts, match12, nmatch)] + [tuple(torch.empty_like(t) for t in (kps, desc, counts, match12, nmatch)) for _ in range(S - 1)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    ex.set_profiling(False)
    def run(n):
        for m in range(n):
            k = m % S
            with torch.cuda.stream(streams[k]):
                exs[k].extract_batch(frames, out=outs[k][:3])
                mt.match_frames_batch(outs[k][0], outs[k][1], outs[k][2], pa, pb, out=outs[k][3:])
    run(2 * S); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(reps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"streams": S, "frames_per_s": round(B * reps / dt, 1), "nmatch": [int(o[4].sum().item()) for o in outs]}))
