import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from ceres_mono_orb_slam2_amd import ORBextractor
frames = bench.make_frames(8, seed=0)
ex = ORBextractor(2000, 1.2, 8, 20, 7)
for i in range(5): ex(frames[i % 8])
ex.set_profiling(True)
t0 = time.perf_counter()
for i in range(50): ex(frames[i % 8])
dt = (time.perf_counter() - t0) / 50
ms, n = ex.stage_ms()
print("wall ms/frame %.3f" % (dt * 1e3), {k: round(v / n, 4) for k, v in ms.items()}, n)
ex.set_profiling(False)
t0 = time.perf_counter()
for i in range(50): ex(frames[i % 8])
print("wall ms/frame (no profiling) %.3f" % ((time.perf_counter() - t0) / 50 * 1e3))
d = torch.from_numpy(frames[:1]).cuda()
for _ in range(3): ex.extract_batch(d)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): ex.extract_batch(d)
torch.cuda.synchronize(); print("device-resident batch of 1: ms %.3f" % ((time.perf_counter() - t0) / 50 * 1e3))
