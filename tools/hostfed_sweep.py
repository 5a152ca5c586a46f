"""Host-fed pipeline (bench.pcie_pipeline: three host threads, host-side stage hand-offs) under different process settings,
each in a fresh process: python tools/hostfed_sweep.py [B=256].  Round 5 used this harness on three earlier forms as well (one host
thread + hipStreamWaitEvent between an upload, a run and a download stream; S independent chains; copies by the library's copy
kernel): their results are in gpurun_out history / DESIGN.md section 6."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np, torch
import bench
from ceres_mono_orb_slam2_amd import ORBextractor, ORBmatcher
B = int(sys.argv[1]); extra = int(sys.argv[2])
fr = bench.make_frames(B, seed=0)
dev = torch.device("cuda", 0)
mt = ORBmatcher(0.9, True)
keep = [torch.cuda.Stream(device=dev) for _ in range(extra)]      # streams that merely exist, created before everything else
r = bench.pcie_pipeline(torch, dev, ORBextractor(bench.NFEAT, 1.2, 8, 20, 7), mt, fr, B, 1e12)
print(json.dumps({k: r[k] for k in ("value", "frac_of_bound", "h2d_GBps", "d2h_GBps")}))
''' % ROOT
B = sys.argv[1] if len(sys.argv) > 1 else "256"
for q in ("4", "8", "12"):
    for extra in (0, 3):
        env = dict(os.environ, GPU_MAX_HW_QUEUES=q)
        p = subprocess.run([sys.executable, "-c", CHILD, B, str(extra)], env=env, capture_output=True, text=True)
        last = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:]
        print("HWQ", q, "extra_streams", extra, last, flush=True)
