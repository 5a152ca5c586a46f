"""Which levels bound k_octree?  Builds orb_extractor.hip with -DORBHIP_OCT_LEVEL_EXPERIMENT into a scratch library (the kernel then
skips the levels not in ORBHIP_OCT_LEVELS) and times the octree stage of the bench's 256-frame batch for several level sets."""
import ctypes as C, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
out = os.path.join(ROOT, "gpurun_out", "octprof"); os.makedirs(out, exist_ok=True)
so = os.path.join(out, "liborbx_oct.so")
csrc = os.path.join(ROOT, "ceres_mono_orb_slam2_amd", "csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-DORBHIP_OCT_LEVEL_EXPERIMENT",
                       "-shared", "-o", so, os.path.join(csrc, "orb_extractor.hip"), os.path.join(csrc, "capi_common.hip")])
import bench
L = C.CDLL(so)
vp, i32 = C.c_void_p, C.c_int
L.orbx_create.argtypes = [i32, C.c_float, i32, i32, i32, i32, C.POINTER(vp)]
L.orbx_extract_batch_device.argtypes = [vp, vp, i32, i32, i32, C.c_size_t, i32, vp, vp, i32, vp, vp]
L.orbx_max_keypoints.argtypes = [vp]
L.orbx_set_profiling.argtypes = [vp, i32]
L.orbx_get_stage_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(i32)]
B = 256
fr = torch.from_numpy(bench.make_frames(B, 0)).cuda()
st = torch.cuda.current_stream().cuda_stream
res = {}
for name, mask in [("all", 0xFF), ("0", 1), ("1", 2), ("2", 4), ("3", 8), ("0-1", 3), ("2-7", 0xFC), ("4-7", 0xF0), ("none", 0)]:
    os.environ["ORBHIP_OCT_LEVELS"] = str(mask)
    h = vp(); assert L.orbx_create(2000, 1.2, 8, 20, 7, 0, C.byref(h)) == 0
    cap = L.orbx_max_keypoints(h)
    kps = torch.empty((B, cap, 7), dtype=torch.float32, device="cuda"); desc = torch.empty((B, cap, 32), dtype=torch.uint8, device="cuda"); cnt = torch.empty((B,), dtype=torch.int32, device="cuda")
    def run():
        assert L.orbx_extract_batch_device(h, fr.data_ptr(), 1241, 376, fr.stride(1), fr.stride(0), B, kps.data_ptr(), desc.data_ptr(), cap, cnt.data_ptr(), vp(st)) == 0
    for _ in range(3): run()
    torch.cuda.synchronize()
    L.orbx_set_profiling(h, 1)
    for _ in range(10): run()
    torch.cuda.synchronize()
    ms = (C.c_float * 8)(); n = i32(0)
    L.orbx_get_stage_ms(h, ms, C.byref(n))
    res[name] = round(ms[2] / max(n.value, 1), 4)
print(json.dumps(res))
