"""In-kernel timeline of the one-launch pyramid kernel (k_pyr_cone): builds the library with -DORBHIP_CONE_PROF into a scratch .so;
workgroup 77 stamps s_memrealtime at: start, level-0 box + tables landed, box stored, tables stored + barrier, end of every level."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
out = os.path.join(ROOT, "gpurun_out", "coneprof"); os.makedirs(out, exist_ok=True)
so = os.path.join(out, "liborbslam_hip_cone.so")
csrc = os.path.join(ROOT, "ceres_mono_orb_slam2_amd", "csrc")
srcs = [os.path.join(csrc, f) for f in ("ba_solver.hip", "capi_common.hip", "orb_extractor.hip", "orb_matcher.hip", "orb_frame.hip", "orb_vocab.hip", "orb_track.hip")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-DORBHIP_CONE_PROF", "-I", os.path.join(ROOT, "include"), "-shared", "-o", so] + srcs)
from ceres_mono_orb_slam2_amd import _lib, ORBextractor, synth
_lib.LIB_PATH = so
L = _lib.load()
seq, offs = synth.make_sequence(11, 1241, 376, 2, "blocks", max_shift=6)
ex = ORBextractor(2000, 1.2, 8, 20, 7)
for _ in range(20): ex(seq[1])
buf = (C.c_ulonglong * 24)()
L.orbx_debug_cone_ticks.argtypes = [C.c_void_p]
L.orbx_debug_cone_ticks(buf)
t = list(buf)
print("cone ticks (us since start):", [round((x - t[0]) / 100.0, 2) for x in t[:12]])
