"""The fused frame-pair matcher on the matrix cores against the VALU kernel (ORBHIP_MATCH_MFMA=1 / 0, one subprocess each): the
match lists must be identical; prints the time per launch.  python tools/match_ab.py [npairs n cap]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np, torch, hashlib
    from ceres_mono_orb_slam2_amd import ORBmatcher, _lib
    npairs, n, cap = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rng = np.random.default_rng(3)
    nf = npairs + 1
    base = rng.integers(0, 256, (cap, 32), dtype=np.uint8)
    desc = np.empty((nf, cap, 32), np.uint8)
    for f in range(nf):                                           # every frame: the base descriptors, a few bits flipped, shuffled
        flip = (rng.random((cap, 256)) < 0.06)
        d = np.unpackbits(base, axis=1) ^ flip.astype(np.uint8)
        desc[f] = np.packbits(d, axis=1)[rng.permutation(cap)]
    kdt = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"), ("octave", "i4"), ("class_id", "i4")])
    kps = np.zeros((nf, cap), kdt); kps["angle"] = rng.uniform(0, 360, (nf, cap)).astype(np.float32)
    counts = rng.integers(max(1, n - 200), n + 1, nf).astype(np.int32); counts[0] = n
    dev = torch.device("cuda:0")
    d_k = torch.from_numpy(kps.view(np.uint8).reshape(nf, cap, -1)).to(dev); d_d = torch.from_numpy(desc).to(dev); d_c = torch.from_numpy(counts).to(dev)
    pa = torch.arange(1, nf, dtype=torch.int32, device=dev); pb = torch.arange(0, nf - 1, dtype=torch.int32, device=dev)
    M = ORBmatcher(0.9, True)
    for _ in range(int(os.environ.get("WARM", "300"))): m, nm = M.match_frames_batch(d_k, d_d, d_c, pa, pb)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    R = int(os.environ.get("REPS", "100"))
    e0.record()
    for _ in range(R): m, nm = M.match_frames_batch(d_k, d_d, d_c, pa, pb)
    e1.record(); torch.cuda.synchronize()
    mh = m.cpu().numpy()
    for p in range(npairs): mh[p, counts[p + 1]:] = -1
    print(json.dumps({"ms": e0.elapsed_time(e1) / R, "sha": hashlib.sha256(mh.tobytes()).hexdigest()[:16], "matches": int(nm.sum().item())}))
    sys.exit(0)
a = [str(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else [256, 1816, 2048])]
res = {}
for mode in ("1", "0"):
    env = dict(os.environ, ORBHIP_MATCH_MFMA=mode)
    out = subprocess.run([sys.executable, __file__, "--child"] + a, env=env, capture_output=True, text=True)
    if out.returncode: print(out.stderr[-2000:]); sys.exit(1)
    res[mode] = json.loads(out.stdout.strip().splitlines()[-1])
    print("ORBHIP_MATCH_MFMA=%s" % mode, res[mode])
assert res["1"]["sha"] == res["0"]["sha"], "match lists differ"
print("identical match lists; matrix cores %.3f ms, VALU %.3f ms per launch of %s pairs" % (res["1"]["ms"], res["0"]["ms"], a[0]))
