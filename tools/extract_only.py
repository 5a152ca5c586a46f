import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ceres_mono_orb_slam2_amd import ORBextractor
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = bench.make_frames(B, seed=0)
d = torch.from_numpy(frames).cuda()
ex = ORBextractor(2000, 1.2, 8, 20, 7)
for _ in range(3):
    out = ex.extract_batch(d)
torch.cuda.synchronize()
print(out[2][:4].cpu().numpy())
