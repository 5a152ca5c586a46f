"""Phase timing of k_pose_lm on one 2000-observation frame (profiling build, -DORBHIP_CHOL_PROF): where an LM iteration's time goes."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
out = os.path.join(ROOT, "gpurun_out", "cholprof"); os.makedirs(out, exist_ok=True)
so = os.path.join(out, "liborbslam_hip_pose.so")
csrc = os.path.join(ROOT, "ceres_mono_orb_slam2_amd", "csrc")
srcs = [os.path.join(csrc, f) for f in ("ba_solver.hip", "capi_common.hip", "orb_extractor.hip", "orb_matcher.hip", "orb_frame.hip", "orb_vocab.hip", "orb_track.hip")]
if not os.path.exists(so) or os.environ.get("REBUILD"):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                           "-DORBHIP_CHOL_PROF", "-I", os.path.join(ROOT, "include"), "-shared", "-o", so] + srcs)
from ceres_mono_orb_slam2_amd import _lib, optimizer, synth
_lib.LIB_PATH = so
L = _lib.load()
L.ba_debug_pose_ticks.argtypes = [C.c_void_p, C.c_int]
p = synth.make_pose_problem(5, n=int(sys.argv[1]) if len(sys.argv) > 1 else 1500)
a = (p["K4"], p["pose0"], p["Xw"], p["uv"], p["inv_sigma2"]) if isinstance(p, dict) else p[:5]
for _ in range(5): r = optimizer.pose_optimization(*a)
L.ba_debug_pose_ticks(None, 1)
N = 50
for _ in range(N): r = optimizer.pose_optimization(*a)
t = (C.c_ulonglong * 8)(); L.ba_debug_pose_ticks(t, 0)
names = ["loads + first evaluation", "thread 0: damped 6x6 solve, candidate", "evaluation at the candidate", "thread 0: decision, new state", "outlier check, write-back"]
it = r[3]["iterations"]
print("iterations per solve:", it)
names += ["  (wave 0: scaled system + factor)", "  (wave 0: substitutions, model cost change)", "  (wave 0: candidate pose)"]
for n_, v in zip(names, list(t)[:8]): print("%-42s %7.2f us per solve" % (n_, v / 100.0 / N))
print("total %.2f us (the wave-0 lines are the parts of the solve phase: they add to it)" % (sum(list(t)[:8]) / 100.0 / N))
