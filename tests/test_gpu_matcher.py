"""GPU parity: Hamming matching through the C ABI vs the CPU oracle (bit-exact integer work)."""
import os

import numpy as np
import pytest

from ceres_mono_orb_slam2_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _m(r=0.9, ori=True):
    from ceres_mono_orb_slam2_amd import ORBmatcher
    return ORBmatcher(r, ori)


def test_descriptor_distance_identities():
    from ceres_mono_orb_slam2_amd import ORBmatcher
    rng = np.random.default_rng(0)
    z = np.zeros(32, np.uint8)
    assert ORBmatcher.DescriptorDistance(z, ~z) == 256
    for _ in range(50):
        a = rng.integers(0, 256, 32, dtype=np.uint8); b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert ORBmatcher.DescriptorDistance(a, a) == 0
        assert ORBmatcher.DescriptorDistance(a, b) == int(np.unpackbits(a ^ b).sum())


@pytest.mark.parametrize("nq,nt", [(1, 1), (50, 70), (257, 1025), (2000, 2000), (3, 0)])
def test_best2_brute_force(oracle, nq, nt):
    rng = np.random.default_rng(nq * 7 + nt)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    if nt > 30:
        t[10] = q[0]; t[20] = q[0]                        # exact tie -> first index wins
    bi, bd, sd = _m().hamming_best2(q, t)
    obi, obd, osd = oracle.hamming_best2(q, t)
    assert np.array_equal(bi, obi) and np.array_equal(bd, obd) and np.array_equal(sd, osd)


def test_best2_csr_ragged(oracle):
    rng = np.random.default_rng(3)
    q = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (400, 32), dtype=np.uint8)
    lens = rng.integers(0, 40, 300); lens[:5] = 0
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    idx = rng.integers(0, 400, int(off[-1])).astype(np.uint32)
    bi, bd, sd = _m().hamming_best2(q, t, off, idx)
    obi, obd, osd = oracle.hamming_best2(q, t, off, idx)
    assert np.array_equal(bi, obi) and np.array_equal(bd, obd) and np.array_equal(sd, osd)
    assert (bi[:5] == -1).all() and (bd[:5] == 256).all()
    # all-empty candidate lists
    off0 = np.zeros(301, np.uint32)
    bi, bd, sd = _m().hamming_best2(q, t, off0, np.zeros(0, np.uint32))
    assert (bi == -1).all() and (bd == 256).all() and (sd == 256).all()


def test_golden_match_fixture():
    import torch
    g = np.load(os.path.join(GOLD, "match_320x240.npz"))
    bi, bd, sd = _m().hamming_best2(g["d1"], g["d2"])
    assert np.array_equal(bi, g["best_idx"]) and np.array_equal(bd, g["best_d"]) and np.array_equal(sd, g["second_d"])
    m, n = _match_pair(g["d1"], g["a1"], g["d2"], g["a2"], 0.9, True)
    assert n == int(g["nmatch"]) and np.array_equal(m, g["match12"])


def _match_pair(d1, a1, d2, a2, ratio, ori, th=None):
    import torch
    cap = max(len(d1), len(d2)) + 5
    kps = torch.zeros((2, cap, 7), dtype=torch.float32)
    kps[0, :len(a1), 3] = torch.from_numpy(np.asarray(a1, np.float32)); kps[1, :len(a2), 3] = torch.from_numpy(np.asarray(a2, np.float32))
    desc = torch.zeros((2, cap, 32), dtype=torch.uint8)
    desc[0, :len(d1)] = torch.from_numpy(d1); desc[1, :len(d2)] = torch.from_numpy(d2)
    counts = torch.tensor([len(d1), len(d2)], dtype=torch.int32)
    pa = torch.tensor([0], dtype=torch.int32).cuda(); pb = torch.tensor([1], dtype=torch.int32).cuda()
    m, nm = _m(ratio, ori).match_frames_batch(kps.cuda(), desc.cuda(), counts.cuda(), pa, pb, th=th)
    torch.cuda.synchronize()
    return m[0, :len(d1)].cpu().numpy(), int(nm[0])


@pytest.mark.parametrize("ratio,ori", [(0.9, True), (0.7, True), (0.9, False)])
def test_match_frames_vs_oracle(oracle, ratio, ori):
    E = oracle.OracleExtractor(1000)
    seq, _ = synth.make_sequence(21, 640, 480, 2, "blocks", max_shift=8)
    k1, d1 = E.extract(seq[0]); k2, d2 = E.extract(seq[1])
    m, n = _match_pair(d1, k1["angle"], d2, k2["angle"], ratio, ori)
    om, on = oracle.match_frames(d1, k1["angle"], d2, k2["angle"], ratio, 50, ori)
    assert n == on and np.array_equal(m, om)
    assert n > 100


def test_match_frames_empty_frames():
    d = np.zeros((0, 32), np.uint8); a = np.zeros(0, np.float32)
    rng = np.random.default_rng(1)
    d2 = rng.integers(0, 256, (10, 32), dtype=np.uint8); a2 = np.zeros(10, np.float32)
    m, n = _match_pair(d2, a2, d, a, 0.9, True)
    assert n == 0 and (m == -1).all()


def test_search_for_initialization_vs_oracle(oracle):
    E = oracle.OracleExtractor(1000)
    seq, _ = synth.make_sequence(31, 640, 480, 2, "blocks", max_shift=6)
    k1, d1 = E.extract(seq[0]); k2, d2 = E.extract(seq[1])
    f = lambda k: np.stack([k["x"], k["y"], k["octave"].astype(np.float32), k["angle"]], 1).astype(np.float32)
    bounds = np.array([0, 640, 0, 480], np.float32)
    for window, ratio in [(100, 0.9), (10, 0.9), (30, 0.6)]:
        prev = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], 1), np.float32)
        om, on, opm = oracle.search_for_initialization(f(k1), d1, f(k2), d2, bounds, prev, window, ratio, True)
        pm = prev.copy()
        n, m = _m(ratio, True).SearchForInitialization(f(k1), d1, f(k2), d2, bounds, pm, window)
        assert n == on and np.array_equal(m, om) and np.array_equal(pm, opm)


def test_full_size_match_properties():
    """2000x2000 brute force at BASELINE size: symmetric-input and shuffle-invariance properties."""
    rng = np.random.default_rng(9)
    d = rng.integers(0, 256, (2000, 32), dtype=np.uint8)
    bi, bd, sd = _m().hamming_best2(d, d)
    assert np.array_equal(bi, np.arange(2000)) and (bd == 0).all() and (sd > 0).all()
    perm = rng.permutation(2000)
    bi2, bd2, sd2 = _m().hamming_best2(d, d[perm])
    assert np.array_equal(perm[bi2], np.arange(2000)) and np.array_equal(sd2, sd)


# ---------------------------------------------------------------- guided searches (M4-M7, M9-M12 families)
def _two_frames(oracle, seed=41):
    E = oracle.OracleExtractor(1000)
    seq, offs = synth.make_sequence(seed, 640, 480, 2, "blocks", max_shift=6)
    k1, d1 = E.extract(seq[0]); k2, d2 = E.extract(seq[1])
    f = lambda k: np.stack([k["x"], k["y"], k["octave"].astype(np.float32), k["angle"]], 1).astype(np.float32)
    return f(k1), d1, f(k2), d2, (offs[1] - offs[0]).astype(np.float32), E


@pytest.mark.parametrize("case", ["mappoints_best2", "lastframe_ori", "reloc_kf", "sim3_pred", "fuse_chi2",
                                  "mappoints_best2+noclaim", "lastframe_ori+noclaim"])
def test_search_by_projection_families_vs_oracle(oracle, case):
    K1, D1, K2, D2, shift, E = _two_frames(oracle)
    noclaim = case.endswith("+noclaim")
    case = case.split("+")[0]
    if noclaim:                         # every query twice; see below
        K1 = np.concatenate([K1, K1]); D1 = np.concatenate([D1, D1])
    rng = np.random.default_rng(5)
    bounds = np.array([0, 640, 0, 480], np.float32)
    nq = len(K1)
    # "map points" = frame-1 keypoints projected into frame 2 by the known shift (+ a little noise)
    q_uv = (K1[:, :2] - shift + rng.normal(0, 0.7, (nq, 2))).astype(np.float32)
    lvl = K1[:, 2].astype(np.int32)
    scale = E.scale[lvl]
    valid = (rng.random(nq) > 0.1).astype(np.uint8)
    taken0 = (rng.random(len(K2)) < 0.15).astype(np.uint8)
    if noclaim:
        # query points without observations (q_valid = 3, src/ORBmatcher.cc:83-84, :1220-1221) leave the feature they are
        # assigned to open: the duplicate of the query (same projection, second half) is then assigned to it again
        h = nq // 2
        q_uv[h:] = q_uv[:h]; valid[h:] = valid[:h]
        valid[:h][(rng.random(h) < 0.33) & (valid[:h] > 0)] = 3
    kw = dict(q_valid=valid)
    if case == "mappoints_best2":      # src/ORBmatcher.cc:42-119
        kw.update(q_radius=4.0 * scale, q_min_level=lvl - 1, q_max_level=lvl, taken=taken0, mode_best2=True, th=100)
        ratio, ori = 0.8, False
    elif case == "lastframe_ori":      # :1161-1271
        kw.update(q_radius=15 * scale, q_min_level=lvl - 1, q_max_level=lvl + 1, taken=taken0, th=100, q_angle=K1[:, 3])
        ratio, ori = 0.9, True
    elif case == "reloc_kf":           # :1273-1384
        kw.update(q_radius=10 * scale, q_min_level=lvl - 1, q_max_level=lvl + 1, taken=taken0, th=64, q_angle=K1[:, 3])
        ratio, ori = 0.75, True
    elif case == "sim3_pred":          # :258-361
        kw.update(q_radius=10 * scale, q_pred_level=np.minimum(lvl + rng.integers(0, 2, nq), 7), taken=taken0, th=50)
        ratio, ori = 0.75, False
    else:                              # Fuse candidate selection, :724-842
        kw.update(q_radius=3 * scale, q_pred_level=np.minimum(lvl + rng.integers(0, 2, nq), 7), th=50,
                  inv_level_sigma2=E.inv_sigma2, chi2_gate=5.99)
        ratio, ori = 0.6, False
    n, m, bd, tk = _m(ratio, ori).search_by_projection(K2, D2, bounds, q_uv, kw.pop("q_radius"), D1, **kw)
    okw = dict(kw); okw.pop("th")
    on, om, obd, otk = oracle.search_by_projection(K2, D2, bounds, q_uv, (4.0 * scale if case == "mappoints_best2" else
                                                   15 * scale if case == "lastframe_ori" else 10 * scale if case in ("reloc_kf", "sim3_pred") else 3 * scale),
                                                   D1, ratio=ratio, th=kw["th"], check_ori=ori, **okw)
    assert n == on and np.array_equal(m, om) and np.array_equal(bd, obd)
    if tk is not None:
        assert np.array_equal(tk, otk)
    if ori:
        assert (m <= -2).any()          # matches the rotation histogram removed keep their slot: -2 - target
    if noclaim:
        slot = np.where(m >= 0, m, np.where(m <= -2, -2 - m, -1)); slot = slot[slot >= 0]
        assert len(slot) > len(set(slot.tolist()))      # some feature really was assigned twice
        return
    assert n > 50
    hit = m >= 0
    assert len(set(m[hit].tolist())) == hit.sum() or case == "fuse_chi2"      # one map point per keypoint when `taken` is tracked


def test_search_by_sim3_vs_oracle(oracle):
    """M11: ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:956-1159) end to end: both window searches on the device grid, the
    mutual-agreement pass (:1145-1157) on the host; same generator as the CPU test against the numpy restatement."""
    from tests.test_oracle_matcher_npref import sim3_queries, BOUNDS
    fr = _two_frames(oracle)
    K1, D1, K2, D2, shift, E = fr
    for seed in (9, 10):
        q = sim3_queries(fr, seed)
        n, m = _m().SearchBySim3(K1, D1, K2, D2, BOUNDS, *q)
        on, om = oracle.search_by_sim3(K1, D1, K2, D2, BOUNDS, *q)
        assert n == on and np.array_equal(m, om) and n > 50
        rng = np.random.default_rng(seed)
        M1 = D1.copy(); M2 = D2.copy()                                   # map-point descriptors != the keyframes' own rows
        for M in (M1, M2):
            M[np.arange(len(M)), rng.integers(0, 32, len(M))] ^= (1 << rng.integers(0, 8, len(M))).astype(np.uint8)
        n, m = _m().SearchBySim3(K1, D1, K2, D2, BOUNDS, *q, q12_desc=M1, q21_desc=M2)
        on, om = oracle.search_by_sim3(K1, D1, K2, D2, BOUNDS, *q, q12_desc=M1, q21_desc=M2)
        assert n == on and np.array_equal(m, om) and n > 50
        B2 = (BOUNDS[0] - 7.0, BOUNDS[1] + 19.0, BOUNDS[2] - 3.0, BOUNDS[3] + 11.0)      # keyframe 2 from another camera: its own grid
        n, m = _m().SearchBySim3(K1, D1, K2, D2, BOUNDS, *q, bounds2=B2)
        on, om = oracle.search_by_sim3(K1, D1, K2, D2, BOUNDS, *q, bounds2=B2)
        assert n == on and np.array_equal(m, om) and n > 50


def _fake_feature_vector(rng, n, nnodes=40):
    node_of = rng.integers(0, nnodes, n)
    nodes = np.unique(node_of)
    off = [0]; idx = []
    for nd in nodes:
        ii = np.nonzero(node_of == nd)[0]
        idx.extend(ii.tolist()); off.append(len(idx))
    return (nodes.astype(np.uint32) * 7 + 3, np.array(off, np.uint32), np.array(idx, np.uint32)), node_of


@pytest.mark.parametrize("strict,ratio", [(False, 0.7), (True, 0.75)])
def test_search_by_bow_vs_oracle(oracle, strict, ratio):
    K1, D1, K2, D2, shift, E = _two_frames(oracle, seed=43)
    rng = np.random.default_rng(6)
    # a synthetic vocabulary level: true correspondences mostly share a node (the vocabulary blob is absent)
    fv2, node2 = _fake_feature_vector(rng, len(K2))
    bi, _, _ = oracle.hamming_best2(D1, D2)
    node1 = np.where(rng.random(len(K1)) < 0.8, node2[bi], rng.integers(0, 40, len(K1)))
    nodes = np.unique(node1); off = [0]; idx = []
    for nd in nodes:
        ii = np.nonzero(node1 == nd)[0]; idx.extend(ii.tolist()); off.append(len(idx))
    fv1 = (nodes.astype(np.uint32) * 7 + 3, np.array(off, np.uint32), np.array(idx, np.uint32))
    valid1 = (rng.random(len(K1)) > 0.2).astype(np.uint8)
    valid2 = (rng.random(len(K2)) > 0.1).astype(np.uint8) if strict else None
    n, m = _m(ratio, True).SearchByBoW(D1, valid1, K1[:, 3], D2, valid2, K2[:, 3], fv1, fv2, strict=strict)
    on, om = oracle.search_by_bow(D1, valid1, K1[:, 3], D2, valid2, K2[:, 3], fv1, fv2, ratio=ratio, th=50, strict=strict, check_ori=True)
    assert n == on and np.array_equal(m, om) and n > 30
    assert (m[valid1 == 0] == -1).all()


def test_search_for_triangulation_vs_oracle(oracle):
    K1, D1, K2, D2, shift, E = _two_frames(oracle, seed=47)
    rng = np.random.default_rng(8)
    fv2, node2 = _fake_feature_vector(rng, len(K2), nnodes=25)
    bi, _, _ = oracle.hamming_best2(D1, D2)
    node1 = np.where(rng.random(len(K1)) < 0.8, node2[bi], rng.integers(0, 25, len(K1)))
    nodes = np.unique(node1); off = [0]; idx = []
    for nd in nodes:
        ii = np.nonzero(node1 == nd)[0]; idx.extend(ii.tolist()); off.append(len(idx))
    fv1 = (nodes.astype(np.uint32) * 7 + 3, np.array(off, np.uint32), np.array(idx, np.uint32))
    um1 = (rng.random(len(K1)) > 0.3).astype(np.uint8); um2 = (rng.random(len(K2)) > 0.3).astype(np.uint8)
    # pure-translation fundamental matrix F12 = [t]x (up to scale) for the known image shift: epipolar lines are
    # parallel to the shift, so true correspondences pass the 3.84 sigma^2 gate; epipole far outside the image
    t = np.array([shift[0], shift[1], 0.0]); t = t / (np.linalg.norm(t) + 1e-12)
    F12 = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]], np.float64)
    for ori in (False, True):
        n, m = _m(0.6, ori).SearchForTriangulation(K1, D1, um1, K2, D2, um2, fv1, fv2, F12, (-5000.0, 240.0), E.scale, E.sigma2)
        on, om = oracle.search_for_triangulation(K1, D1, um1, K2, D2, um2, fv1, fv2, F12, (-5000.0, 240.0), E.scale, E.sigma2,
                                                 check_ori=ori)
        assert n == on and np.array_equal(m, om) and n > 20
        assert (m[um1 == 0] == -1).all() and (um2[m[m >= 0]] == 1).all()


@pytest.mark.parametrize("seed", range(12))
def test_match_frames_random_sizes_and_ties(oracle, seed):
    """Random frame sizes (1 .. 4100 keypoints, more targets than the LDS tile holds at once included) with descriptors drawn
    from a small pool so that equal distances are everywhere: the first-minimum / second-distance bookkeeping and the
    rotation histogram must follow the reference's scan order exactly."""
    rng = np.random.default_rng(500 + seed)
    n1 = int(rng.choice([1, 2, 63, 64, 65, 500, 1999, 2048, 4100])); n2 = int(rng.choice([1, 3, 64, 127, 1000, 2049, 4100]))
    pool = rng.integers(0, 256, (int(rng.choice([4, 40, 400])), 32), dtype=np.uint8)
    def draw(n):
        d = pool[rng.integers(0, len(pool), n)].copy()
        flip = rng.random(n) < 0.5                                      # half of them one or two bits away -> distances 0, 1, 2 tie a lot
        d[flip, rng.integers(0, 32, flip.sum())] ^= (1 << rng.integers(0, 8, flip.sum())).astype(np.uint8)
        return d
    d1, d2 = draw(n1), draw(n2)
    a1 = rng.choice([0.0, 12.0, 90.0, 359.5], n1).astype(np.float32) + rng.uniform(0, 0.4, n1).astype(np.float32)
    a2 = rng.choice([0.0, 12.0, 90.0, 359.5], n2).astype(np.float32) + rng.uniform(0, 0.4, n2).astype(np.float32)
    ratio = float(rng.choice([0.6, 0.9, 1.0])); ori = bool(rng.integers(0, 2)); th = int(rng.choice([50, 100, 3]))
    m, n = _match_pair(d1, a1, d2, a2, ratio, ori, th)
    om, on = oracle.match_frames(d1, a1, d2, a2, ratio, th, ori)
    assert n == on and np.array_equal(m, om)
    bi, bd, sd = _m().hamming_best2(d1, d2)
    obi, obd, osd = oracle.hamming_best2(d1, d2)
    assert np.array_equal(bi, obi) and np.array_equal(bd, obd) and np.array_equal(sd, osd)


def test_match_frames_batch_one_workgroup_per_pair(oracle):
    """A launch with a pair per CU takes the matrix-core matcher's nsplit = 1 shape (one workgroup walks all query blocks of its
    pair, no merge through global memory): 330 ragged pairs (empty frames, one keypoint, sizes around the 512-query block and
    the 128-target chunk) against the oracle, pair by pair."""
    import torch
    rng = np.random.default_rng(77)
    nf, cap = 331, 1100
    sizes = rng.choice([0, 1, 15, 16, 17, 127, 128, 129, 511, 512, 513, 700, 1024, 1025, 1100], nf).astype(np.int32)
    pool = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    desc = np.zeros((nf, cap, 32), np.uint8); ang = np.zeros((nf, cap), np.float32)
    for f in range(nf):
        d = pool[rng.integers(0, len(pool), sizes[f])].copy()
        flip = rng.random(sizes[f]) < 0.6
        d[flip, rng.integers(0, 32, flip.sum())] ^= (1 << rng.integers(0, 8, flip.sum())).astype(np.uint8)
        desc[f, :sizes[f]] = d
        ang[f, :sizes[f]] = rng.choice([0.0, 45.0, 200.0], sizes[f]).astype(np.float32) + rng.uniform(0, 0.5, sizes[f]).astype(np.float32)
    kps = np.zeros((nf, cap, 7), np.float32); kps[:, :, 3] = ang
    pa = torch.arange(1, nf, dtype=torch.int32).cuda(); pb = torch.arange(0, nf - 1, dtype=torch.int32).cuda()
    m, nm = _m(0.9, True).match_frames_batch(torch.from_numpy(kps).cuda(), torch.from_numpy(desc).cuda(), torch.from_numpy(sizes).cuda(), pa, pb)
    torch.cuda.synchronize()
    m = m.cpu().numpy(); nm = nm.cpu().numpy()
    for p in range(nf - 1):
        a, b = p + 1, p
        om, on = oracle.match_frames(desc[a, :sizes[a]], ang[a, :sizes[a]], desc[b, :sizes[b]], ang[b, :sizes[b]], 0.9, 50, True)
        assert nm[p] == on and np.array_equal(m[p, :sizes[a]], om), p
        assert (m[p, sizes[a]:] == -1).all(), p


@pytest.mark.parametrize("shape", [(40, 900, 1000), (330, 500, 520), (3, 1816, 2048), (2, 4000, 4080)])
def test_match_frames_matrix_cores_equal_valu_kernel(shape):
    """ORBHIP_MATCH_MFMA=1 / 0 (tools/match_ab.py, one process each: the switch is read once): identical match lists."""
    import subprocess, sys, json
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "match_ab.py")
    sha = {}
    for mode in ("1", "0"):
        env = dict(os.environ, ORBHIP_MATCH_MFMA=mode, WARM="1", REPS="2")
        out = subprocess.run([sys.executable, tool, "--child"] + [str(x) for x in shape], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        sha[mode] = json.loads(out.stdout.strip().splitlines()[-1])
    assert sha["1"]["sha"] == sha["0"]["sha"] and sha["1"]["matches"] == sha["0"]["matches"] > 0


def test_match_frames_extreme_weights_and_sizes(oracle):
    """The matrix-core matcher computes d - |a| = |b| - 2 |a & b| in the accumulator: descriptors of weight 0 and 256 (d - |a| from
    -256 to +256), exact duplicates (distance 0 ties: the first target wins, the second-best distance is 0 too), a frame at the
    kernel's capacity limit (4080 keypoints: 255 target tiles) and one just above it (the VALU kernel takes over)."""
    rng = np.random.default_rng(123)
    zeros = np.zeros((1, 32), np.uint8); ones = np.full((1, 32), 255, np.uint8)
    rnd = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    half = np.concatenate([np.full((1, 16), 255, np.uint8), np.zeros((1, 16), np.uint8)], 1)
    pool = np.concatenate([zeros, ones, half, 255 - half, rnd, rnd[:10], zeros, ones])
    for n1, n2 in [(len(pool), len(pool)), (700, 4080), (4080, 33), (4081, 600)]:
        d1 = pool[rng.integers(0, len(pool), n1)] if n1 != len(pool) else pool.copy()
        d2 = pool[rng.integers(0, len(pool), n2)] if n2 != len(pool) else pool[::-1].copy()
        a1 = rng.uniform(0, 360, n1).astype(np.float32); a2 = rng.uniform(0, 360, n2).astype(np.float32)
        for ratio, th, ori in [(0.9, 50, True), (1.0, 256, False)]:
            import torch
            cap = max(n1, n2)                                   # (cap itself at the limit, unlike _match_pair's + 5)
            kps = torch.zeros((2, cap, 7), dtype=torch.float32)
            kps[0, :n1, 3] = torch.from_numpy(a1); kps[1, :n2, 3] = torch.from_numpy(a2)
            desc = torch.zeros((2, cap, 32), dtype=torch.uint8)
            desc[0, :n1] = torch.from_numpy(d1); desc[1, :n2] = torch.from_numpy(d2)
            counts = torch.tensor([n1, n2], dtype=torch.int32)
            pa = torch.tensor([0], dtype=torch.int32).cuda(); pb = torch.tensor([1], dtype=torch.int32).cuda()
            m, nm = _m(ratio, ori).match_frames_batch(kps.cuda(), desc.cuda(), counts.cuda(), pa, pb, th=th)
            torch.cuda.synchronize()
            om, on = oracle.match_frames(d1, a1, d2, a2, ratio, th, ori)
            assert int(nm[0]) == on and np.array_equal(m[0, :n1].cpu().numpy(), om), (n1, n2, ratio, th, ori)


def test_rotation_pass_keeps_the_three_maxima_bins(oracle):
    """The construction tools/pin/pin_matcher.cpp uses to pin ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:1386-1418) through the
    library's fused rotation-consistency pass, checked here against the oracle's three_maxima: one vocabulary node, every query an exact
    copy of one target, angle differences with prescribed bin counts (clear maxima, the `< 0.1 max1` cut-offs, ties).  The reference's
    bins are 30 DEGREES wide (factor = 1.0f / HISTO_LENGTH, bin = round(rot * factor): only bins 0 .. 12 of the 30 are ever used)."""
    rng = np.random.default_rng(21)
    cases = []
    for spec in ({3: 40, 4: 30, 7: 20, 10: 5}, {0: 50, 11: 4, 10: 4}, {5: 50, 6: 30, 7: 4}, {2: 10, 9: 10, 1: 10, 5: 10}, {8: 12, 1: 12, 11: 7, 4: 7}):
        c = np.zeros(30, np.int32)
        for b, v in spec.items(): c[b] = v
        cases.append(c)
    for _ in range(20):
        c = np.zeros(30, np.int32); c[:12] = np.where(rng.random(12) < 0.5, rng.integers(0, 25, 12), 0)
        cases.append(c)
    for c in cases:
        n = int(c.sum())
        if n == 0: continue
        desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        bin_of = np.repeat(np.arange(30), c)
        rot = 30.0 * bin_of + np.where(bin_of == 0, rng.integers(0, 11, n), rng.integers(-10, 11, n))      # (bin 0: -10 degrees would wrap to 350 = bin 12)
        a2 = rng.integers(0, 360, n).astype(np.float32)
        a1 = np.mod(a2 + rot, 360.0).astype(np.float32)
        fv = (np.array([7], np.uint32), np.array([0, n], np.uint32), np.arange(n, dtype=np.uint32))
        nm, m = _m(0.99, True).SearchByBoW(desc, None, a1, desc, None, a2, fv, fv, strict=False, th=50)
        assert ((m == np.arange(n)) | (m == -1)).all()
        kept = set(bin_of[m >= 0].tolist())
        realised = np.bincount([oracle.rot_bin(x, y) for x, y in zip(a1, a2)], minlength=30).astype(np.int32)
        assert np.array_equal(realised, c)                          # the angles really fall into the prescribed bins
        want = set(int(b) for b in oracle.three_maxima(c) if b >= 0 and c[b] > 0)
        assert kept == want, (c.tolist(), sorted(kept), sorted(want))
