"""The bench's front-end step in isolation, for rocprofv3 passes: `reps` times (extract + match) of ONE batch of B of the
bench's frames on one stream.  usage: python tools/frontend_only.py [B=256] [reps=3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ceres_mono_orb_slam2_amd import ORBextractor, ORBmatcher
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = torch.from_numpy(bench.make_frames(B, seed=0)).cuda()
ex = ORBextractor(bench.NFEAT, 1.2, 8, 20, 7)
mt = ORBmatcher(0.9, True)
a = torch.arange(B, dtype=torch.int32, device="cuda"); b = (a + B - 1) % B
for _ in range(reps):
    kps, desc, counts = ex.extract_batch(d)
    m12, nm = mt.match_frames_batch(kps, desc, counts, a, b)
torch.cuda.synchronize()
print("frames", B, "mean keypoints %.1f" % counts.float().mean().item(), "mean matches %.1f" % nm.float().mean().item())
