"""When the queries of k_trk_greedy's dataflow pass settle (profiling build, -DORBHIP_TRK_PROF): settles per microsecond since the
pass started, polls of the slowest thread, queries still waiting after the first pass.  Tells a long thin dependency tail (latency
of a link) from a fat one (LDS throughput)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
out = os.path.join(ROOT, "gpurun_out", "trkprof"); os.makedirs(out, exist_ok=True)
so = os.path.join(out, "liborbslam_hip_trk.so")
csrc = os.path.join(ROOT, "ceres_mono_orb_slam2_amd", "csrc")
srcs = [os.path.join(csrc, f) for f in ("ba_solver.hip", "capi_common.hip", "orb_extractor.hip", "orb_matcher.hip", "orb_frame.hip", "orb_vocab.hip", "orb_track.hip")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-DORBHIP_TRK_PROF", "-I", os.path.join(ROOT, "include"), "-shared", "-o", so] + srcs)
from ceres_mono_orb_slam2_amd import _lib, ORBextractor, tracking, synth
_lib.LIB_PATH = so
L = _lib.load()
K4 = np.array([718.856, 718.856, 607.1928, 185.2157], np.float32)
B = np.array([0, 1241, 0, 376], np.float32)
seq, offs = synth.make_sequence(11, 1241, 376, 2, "blocks", max_shift=6)
ex = ORBextractor(2000, 1.2, 8, 20, 7)
k0, d0 = ex(seq[0])
n = len(k0); depth = 18.0
X = np.stack([(k0["x"] - K4[2]) / K4[0] * depth, (k0["y"] - K4[3]) / K4[1] * depth, np.full(n, depth)], 1).astype(np.float64)
sh = (offs[1] - offs[0]).astype(np.float64)
T = np.eye(4); T[0, 3] = -sh[0] * depth / K4[0] + 0.01; T[1, 3] = -sh[1] * depth / K4[1] - 0.01
a = (ex, seq[1], K4, B, T, X, d0, k0["octave"].astype(np.int32), k0["angle"].astype(np.float32), np.ones(n, np.uint8), 15.0, True)
for _ in range(10): r = tracking.track_with_motion_model(*a, copy=False)
L.orbt_debug_prof.argtypes = [C.c_void_p, C.c_int]
L.orbt_debug_prof(None, 1)
r = tracking.track_with_motion_model(*a, copy=False)
h = (C.c_int * 72)(); L.orbt_debug_prof(h, 0)
h = list(h)
print("matches", r["nmatches"], "left after pass one", h[65], "polls of the slowest thread", h[64])
print("settles per us:", [x for x in h[:64]][:max(i for i, x in enumerate(h[:64]) if x) + 1])
print("longest chain of waits (links):", h[66], " mean:", round(h[67] / max(1, sum(h[:64])), 2))
