"""Random launches of orbm_match_frames_batch_device (matrix-core kernel by default; ORBHIP_MATCH_MFMA=0: the VALU kernel) against
the oracle, pair by pair: random pair counts (both launch shapes: one workgroup per pair / split pairs), capacities, ragged
keypoint counts, tie-heavy descriptor pools, thresholds and ratios.  python tools/fuzz_match.py [N] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ceres_mono_orb_slam2_amd import ORBmatcher
from oracle import pyoracle as oracle
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0; pairs = 0
for it in range(N):
    cap = int(rng.choice([16, 65, 300, 512, 513, 1100, 2048, 3000]))
    npairs = int(rng.choice([1, 2, 7, 40, 200, 330])) if cap <= 600 else int(rng.choice([1, 3, 12, 40]))
    nf = npairs + 1
    pool = rng.integers(0, 256, (int(rng.choice([3, 30, 300])), 32), dtype=np.uint8)
    if rng.random() < 0.3: pool[0] = 0; pool[-1] = 255
    sizes = rng.integers(0, cap + 1, nf).astype(np.int32)
    if rng.random() < 0.5: sizes[rng.integers(0, nf)] = cap
    desc = np.zeros((nf, cap, 32), np.uint8); ang = np.zeros((nf, cap), np.float32)
    for f in range(nf):
        d = pool[rng.integers(0, len(pool), sizes[f])].copy()
        flip = rng.random(sizes[f]) < 0.7
        d[flip, rng.integers(0, 32, flip.sum())] ^= (1 << rng.integers(0, 8, flip.sum())).astype(np.uint8)
        desc[f, :sizes[f]] = d
        ang[f, :sizes[f]] = rng.choice([0.0, 30.0, 191.0, 359.8], sizes[f]).astype(np.float32) + rng.uniform(0, 0.3, sizes[f]).astype(np.float32)
    kps = np.zeros((nf, cap, 7), np.float32); kps[:, :, 3] = ang
    ratio = float(rng.choice([0.6, 0.9, 1.0])); th = int(rng.choice([3, 50, 100, 256])); ori = bool(rng.integers(0, 2))
    pa = torch.from_numpy(rng.integers(0, nf, npairs).astype(np.int32)).cuda(); pb = torch.from_numpy(rng.integers(0, nf, npairs).astype(np.int32)).cuda()
    m, nm = ORBmatcher(ratio, ori).match_frames_batch(torch.from_numpy(kps).cuda(), torch.from_numpy(desc).cuda(), torch.from_numpy(sizes).cuda(), pa, pb, th=th)
    torch.cuda.synchronize()
    m = m.cpu().numpy(); nm = nm.cpu().numpy(); A = pa.cpu().numpy(); Bv = pb.cpu().numpy()
    for p in range(npairs):
        a, b = int(A[p]), int(Bv[p])
        om, on = oracle.match_frames(desc[a, :sizes[a]], ang[a, :sizes[a]], desc[b, :sizes[b]], ang[b, :sizes[b]], ratio, th, ori)
        ok = nm[p] == on and np.array_equal(m[p, :sizes[a]], om) and (m[p, sizes[a]:] == -1).all()
        pairs += 1
        if not ok:
            bad += 1
            print("MISMATCH launch %d pair %d: cap %d npairs %d n1 %d n2 %d ratio %.1f th %d ori %d" % (it, p, cap, npairs, sizes[a], sizes[b], ratio, th, ori))
print("fuzz_match: %d launches, %d pairs, %d differences" % (N, pairs, bad))
sys.exit(1 if bad else 0)
