"""PCIe-inclusive front-end rates (never bench.py's `value`): (a) host-pointer orbx_extract, one frame per call;
(b) a 256-frame batch uploaded from pinned host memory, extracted + matched, counts downloaded, per step."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from ceres_mono_orb_slam2_amd import ORBextractor, ORBmatcher
B = 256
frames = bench.make_frames(B, seed=0)
ex = ORBextractor(2000, 1.2, 8, 20, 7)
for i in range(5): ex(frames[i])
t0 = time.perf_counter()
for i in range(64): ex(frames[i])
single = 64 / (time.perf_counter() - t0)
mt = ORBmatcher(0.9, True)
pin = torch.from_numpy(frames).pin_memory()
dev = torch.device("cuda:0")
d = torch.empty_like(pin, device=dev)
pa = torch.arange(B, dtype=torch.int32, device=dev); pb = (pa + B - 1) % B
def step():
    d.copy_(pin, non_blocking=True)
    k, de, c = ex.extract_batch(d)
    m, nm = mt.match_frames_batch(k, de, c, pa, pb)
    return c.cpu(), nm.cpu()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(json.dumps({"host_api_single_frame_fps": single, "pcie_inclusive_batch_fps": B / dt, "pcie_inclusive_batch_ms": dt * 1e3,
                  "upload_GBps_equiv": frames.nbytes / dt / 1e9}))
