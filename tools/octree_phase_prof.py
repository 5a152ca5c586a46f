"""Phase timing of k_octree: builds orb_extractor.hip with -DORBHIP_OCT_PROF into a scratch library, runs the bench's 256-frame
batch through it and prints, per pyramid level, the mean time a (frame, level) workgroup spends in each phase and its sweeps."""
import ctypes as C, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
out = os.path.join(ROOT, "gpurun_out", "octprof"); os.makedirs(out, exist_ok=True)
so = os.path.join(out, "liborbx_octp.so")
csrc = os.path.join(ROOT, "ceres_mono_orb_slam2_amd", "csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-DORBHIP_OCT_PROF", "-mllvm", "-amdgpu-mfma-vgpr-form", "-I" + os.path.join(ROOT, "include"),
                       "-shared", "-o", so, os.path.join(csrc, "orb_extractor.hip"), os.path.join(csrc, "capi_common.hip")])
import bench
L = C.CDLL(so)
vp, i32 = C.c_void_p, C.c_int
L.orbx_create.argtypes = [i32, C.c_float, i32, i32, i32, i32, C.POINTER(vp)]
L.orbx_extract_batch_device.argtypes = [vp, vp, i32, i32, i32, C.c_size_t, i32, vp, vp, i32, vp, vp]
L.orbx_max_keypoints.argtypes = [vp]
h = vp(); assert L.orbx_create(2000, 1.2, 8, 20, 7, 0, C.byref(h)) == 0
B = 256
fr = torch.from_numpy(bench.make_frames(B, 0)).cuda()
cap = L.orbx_max_keypoints(h)
kps = torch.empty((B, cap, 7), dtype=torch.float32, device="cuda"); desc = torch.empty((B, cap, 32), dtype=torch.uint8, device="cuda"); cnt = torch.empty((B,), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run():
    assert L.orbx_extract_batch_device(h, fr.data_ptr(), 1241, 376, fr.stride(1), fr.stride(0), B, kps.data_ptr(), desc.data_ptr(), cap, cnt.data_ptr(), vp(st)) == 0
buf = (C.c_ulonglong * (16 * 16))()
for _ in range(3): run()
L.orbx_debug_oct_prof(buf, 1)
run()
L.orbx_debug_oct_prof(buf, 0)
names = ["gather+init", "A scan", "B key loop", "C plain", "C final sort", "D-G lists", "H key loop", "best+out"]
res = {}
for l in range(8):
    row = [buf[16 * l + k] for k in range(16)]
    n = max(row[15], 1)
    d = {names[k]: round(row[k] / n * 0.01, 2) for k in range(8)}      # us per workgroup
    d["sweeps"] = round(row[14] / n, 2); d["total_us"] = round(sum(row[:8]) / n * 0.01, 1)
    d["slowest_workgroup_us"] = round(row[13] * 0.01, 1); d["max_keys"] = int(row[12])
    res["level %d" % l] = d
print(json.dumps(res, indent=1))
