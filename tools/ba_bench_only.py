import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ceres_mono_orb_slam2_amd import ba_bench
dev = torch.device("cuda", 0)
for i in range(2):
    r = ba_bench.run(dev, cpu=False)
    print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items() if 'note' not in k})
