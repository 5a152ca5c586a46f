"""The extractor + matcher in the configuration bench.py times, checked against the CPU oracle.

bench.py's timed pass: batches of >= 8 frames (side_mode 1: k_blur7 on the extractor's side stream beside FAST + octree,
csrc/orb_extractor.hip run_batch), batch m on HIP stream m % S with its own extractor context (S = 2), every batch extracted
and matched (frame i against frame i - 1 of its batch).  Reference path: src/ORBextractor.cc:1043-1105 per frame; the match is
the brute-force pass of BASELINE.json (oracle.match_frames).

`check(nframes, rounds)` drives exactly that launch pattern on `rounds` x S distinct batches - all enqueued before the first
synchronisation, so that the two contexts' kernels and both side streams really overlap - and compares EVERY frame's keypoints
and descriptors and EVERY pair's match list with the oracle, bit for bit.  Run as a script it prints one JSON line (the test
suite runs it in a subprocess under ORBHIP_POISON=ff: every fresh device buffer filled with 0xff, so that a kernel reading
something another stream has not written yet cannot pass by luck)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H, NFEAT = 1241, 376, 2000


def make_batches(nbatches, nframes, seed=70):
    from ceres_mono_orb_slam2_amd import synth
    fams = ["blocks", "checker", "blocks", "flat"]
    out = []
    for b in range(nbatches):
        per = (nframes + 1) // 2
        a, _ = synth.make_sequence(seed + 10 * b, W, H, per, fams[b % len(fams)], max_shift=8)
        c, _ = synth.make_sequence(seed + 10 * b + 1, W, H, nframes - per, fams[(b + 1) % len(fams)], max_shift=8)
        out.append(np.concatenate([a, c]))
    return out


def check(nframes=16, rounds=2, streams=2, oracle=None):
    import torch
    from ceres_mono_orb_slam2_amd import ORBextractor, ORBmatcher
    if oracle is None:
        from oracle import pyoracle as oracle
    dev = torch.device("cuda", 0)
    S = streams
    host = make_batches(rounds * S, nframes)
    batches = [torch.from_numpy(b).to(dev) for b in host]
    exs = [ORBextractor(NFEAT, 1.2, 8, 20, 7) for _ in range(S)]
    mt = ORBmatcher(0.9, True)
    cap = exs[0].max_keypoints
    st = [torch.cuda.Stream(device=dev) for _ in range(S)]
    pa = torch.arange(nframes, dtype=torch.int32, device=dev)
    pb = (pa + nframes - 1) % nframes
    outs = []
    torch.cuda.synchronize()
    # warm-up on the same contexts (allocations, geometry) and then the checked launches, all enqueued back to back
    for m in range(S):
        with torch.cuda.stream(st[m]):
            exs[m].extract_batch(batches[m])
    for m in range(rounds * S):
        k = m % S
        with torch.cuda.stream(st[k]):
            o = (torch.empty((nframes, cap, 7), dtype=torch.float32, device=dev), torch.empty((nframes, cap, 32), dtype=torch.uint8, device=dev),
                 torch.empty((nframes,), dtype=torch.int32, device=dev), torch.empty((nframes, cap), dtype=torch.int32, device=dev),
                 torch.empty((nframes,), dtype=torch.int32, device=dev))
            exs[k].extract_batch(batches[m], out=o[:3])
            mt.match_frames_batch(o[0], o[1], o[2], pa, pb, out=o[3:])
            outs.append(o)
    torch.cuda.synchronize()
    E = oracle.OracleExtractor(NFEAT)
    nkp = nmatch = 0
    for m, o in enumerate(outs):
        kps, desc, counts, m12, nm = (t.cpu().numpy() for t in o)
        ref = [E.extract(host[m][f]) for f in range(nframes)]
        for f in range(nframes):
            ok, od = ref[f]
            n = int(counts[f])
            assert n == len(ok), "batch %d frame %d: %d keypoints, oracle %d" % (m, f, n, len(ok))
            assert np.array_equal(kps[f, :n].view(np.uint8).reshape(n, 28), ok.view(np.uint8).reshape(n, 28)), "keypoints of batch %d frame %d" % (m, f)
            assert np.array_equal(desc[f, :n], od), "descriptors of batch %d frame %d" % (m, f)
            nkp += n
        for f in range(nframes):
            p = (f + nframes - 1) % nframes
            om, on = oracle.match_frames(ref[f][1], ref[f][0]["angle"], ref[p][1], ref[p][0]["angle"], 0.9, 50, True)
            n = int(counts[f])
            assert int(nm[f]) == on, "match count of batch %d pair %d" % (m, f)
            assert np.array_equal(m12[f, :n], om), "match list of batch %d pair %d" % (m, f)
            nmatch += on
    return {"batches": len(outs), "frames_per_batch": nframes, "streams": S, "keypoints": nkp, "matches": nmatch,
            "poison": os.environ.get("ORBHIP_POISON")}


if __name__ == "__main__":
    print(json.dumps(check(int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 2)))
