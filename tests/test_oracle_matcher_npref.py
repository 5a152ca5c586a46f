"""De-twinned matcher evidence (CPU): the C++ oracle (oracle/match_oracle.cpp) against the independent numpy restatements
in tests/npmatch.py, which were written from the reference lines they cite and share no text with the oracle.  The GPU
parity tests (tests/test_gpu_matcher.py) then compare the HIP path with the oracle on the same families."""
import numpy as np
import pytest

from ceres_mono_orb_slam2_amd import synth
from tests import npmatch as NP


@pytest.fixture(scope="module")
def frames(oracle):
    E = oracle.OracleExtractor(1000)
    seq, offs = synth.make_sequence(41, 640, 480, 2, "blocks", max_shift=6)
    k1, d1 = E.extract(seq[0]); k2, d2 = E.extract(seq[1])
    f = lambda k: np.stack([k["x"], k["y"], k["octave"].astype(np.float32), k["angle"]], 1).astype(np.float32)
    return f(k1), d1, f(k2), d2, (offs[1] - offs[0]).astype(np.float32), E


BOUNDS = np.array([0, 640, 0, 480], np.float32)


def test_descriptor_distance_and_three_maxima(oracle):
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (64, 32), dtype=np.uint8); b = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    for i in range(64):
        assert NP.descriptor_distance(a[i], b[i]) == oracle.descriptor_distance(a[i], b[i])
    for _ in range(200):
        h = rng.integers(0, rng.choice([3, 20, 400]), 30).astype(np.int32)
        assert NP.three_maxima(h.tolist()) == tuple(oracle.three_maxima(h))
    for _ in range(500):
        a1, a2 = np.float32(rng.uniform(0, 360)), np.float32(rng.uniform(0, 360))
        assert NP.rot_bin(a1, a2) == oracle.rot_bin(a1, a2)


def test_features_in_area(oracle, frames):
    K1, D1, K2, D2, shift, E = frames
    rng = np.random.default_rng(1)
    g = NP.Grid(K2, BOUNDS)
    nq = 300
    q = np.stack([rng.uniform(-30, 670, nq), rng.uniform(-30, 510, nq)], 1).astype(np.float32)
    r = rng.choice([1.0, 7.5, 40.0, 100.0, 900.0], nq).astype(np.float32)
    mn = rng.integers(-1, 4, nq).astype(np.int32); mx = (mn + rng.integers(-1, 3, nq)).astype(np.int32)
    off, idx = oracle.features_in_area(K2, BOUNDS, q, r, mn, mx)
    for i in range(nq):
        assert g.features_in_area(q[i, 0], q[i, 1], r[i], int(mn[i]), int(mx[i])) == idx[off[i]:off[i + 1]].tolist(), i
    m1 = np.full(nq, -1, np.int32)
    off, idx = oracle.features_in_area(K2, BOUNDS, q, r, m1, m1)              # KeyFrame::GetFeaturesInArea
    for i in range(nq):
        assert g.features_in_area(q[i, 0], q[i, 1], r[i]) == idx[off[i]:off[i + 1]].tolist(), i


@pytest.mark.parametrize("window,ratio", [(100, 0.9), (10, 0.9), (30, 0.6)])
def test_search_for_initialization(oracle, frames, window, ratio):
    K1, D1, K2, D2, shift, E = frames
    prev = np.ascontiguousarray(K1[:, :2], np.float32)
    om, on, opm = oracle.search_for_initialization(K1, D1, K2, D2, BOUNDS, prev, window, ratio, True)
    m, n, pm = NP.search_for_initialization(K1, D1, K2, D2, BOUNDS, prev, window, ratio, True)
    assert n == on and np.array_equal(m, om) and np.array_equal(pm, opm) and n > 20


def _queries(frames, seed=5):
    K1, D1, K2, D2, shift, E = frames
    rng = np.random.default_rng(seed)
    nq = len(K1)
    q_uv = (K1[:, :2] - shift + rng.normal(0, 0.7, (nq, 2))).astype(np.float32)
    lvl = K1[:, 2].astype(np.int32)
    valid = (rng.random(nq) > 0.1).astype(np.uint8)
    taken0 = (rng.random(len(K2)) < 0.15).astype(np.uint8)
    pred = np.minimum(lvl + rng.integers(0, 2, nq), 7).astype(np.int32)
    return q_uv, lvl, E.scale[lvl].astype(np.float32), valid, taken0, pred


def _no_claim_variant(valid, q_uv, rad, lvl, desc, seed=11):
    """Every query twice (second copy right after the first pass), a third of the valid ones marked "no observations"
    (q_valid = 3): their match leaves the feature open, so the duplicate can be assigned to it again."""
    rng = np.random.default_rng(seed)
    v = valid.copy()
    v[(rng.random(len(v)) < 0.33) & (valid > 0)] = 3
    two = lambda a: np.concatenate([a, a])
    return two(v), two(q_uv), two(rad), two(lvl), two(desc)


def test_projection_family_map_points(oracle, frames):                     # M4, src/ORBmatcher.cc:42-119
    K1, D1, K2, D2, shift, E = frames
    q_uv, lvl, scale, valid, taken0, pred = _queries(frames)
    rad = (4.0 * scale).astype(np.float32)
    on, om, obd, otk = oracle.search_by_projection(K2, D2, BOUNDS, q_uv, rad, D1, q_min_level=lvl - 1, q_max_level=lvl, q_valid=valid,
                                                   taken=taken0, mode_best2=True, ratio=0.8, th=100, check_ori=False)
    n, m, tk = NP.search_by_projection_mappoints(K2, D2, BOUNDS, q_uv, rad, lvl, D1, valid, taken0, 0.8)
    assert n == on and np.array_equal(m, om) and np.array_equal(tk, otk) and n > 50
    # query points without observations do not close the feature they are assigned to (:83-84): q_valid = 3; with duplicated
    # queries the second copy then lands on the same feature
    v3, q2, r2, l2, d2 = _no_claim_variant(valid, q_uv, rad, lvl, D1)
    on3, om3, _, otk3 = oracle.search_by_projection(K2, D2, BOUNDS, q2, r2, d2, q_min_level=l2 - 1, q_max_level=l2, q_valid=v3,
                                                    taken=taken0, mode_best2=True, ratio=0.8, th=100, check_ori=False)
    n3, m3, tk3 = NP.search_by_projection_mappoints(K2, D2, BOUNDS, q2, r2, l2, d2, v3, taken0, 0.8)
    assert n3 == on3 and np.array_equal(m3, om3) and np.array_equal(tk3, otk3)
    hit = om3[om3 >= 0]
    assert len(hit) > len(set(hit.tolist()))                      # some feature really was assigned twice


@pytest.mark.parametrize("th,mult", [(100, 15), (64, 10)])
def test_projection_family_frame(oracle, frames, th, mult):               # M5 :1161-1271, M7 :1273-1384
    K1, D1, K2, D2, shift, E = frames
    q_uv, lvl, scale, valid, taken0, pred = _queries(frames, seed=6)
    rad = (mult * scale).astype(np.float32)
    on, om, obd, otk = oracle.search_by_projection(K2, D2, BOUNDS, q_uv, rad, D1, q_min_level=lvl - 1, q_max_level=lvl + 1, q_valid=valid,
                                                   taken=taken0, q_angle=K1[:, 3], ratio=0.9, th=th, check_ori=True)
    n, m, tk = NP.search_by_projection_frame(K2, D2, BOUNDS, q_uv, rad, lvl, D1, valid, K1[:, 3], taken0, th, True)
    assert n == on and np.array_equal(m, om) and np.array_equal(tk, otk) and n > 50
    assert (om <= -2).any()                                       # the rotation histogram removed something: encoded -2 - slot
    v3, q2, r2, l2, d2 = _no_claim_variant(valid, q_uv, rad, lvl, D1)
    a2 = np.concatenate([K1[:, 3], K1[:, 3]])
    on3, om3, _, otk3 = oracle.search_by_projection(K2, D2, BOUNDS, q2, r2, d2, q_min_level=l2 - 1, q_max_level=l2 + 1, q_valid=v3,
                                                    taken=taken0, q_angle=a2, ratio=0.9, th=th, check_ori=True)
    n3, m3, tk3 = NP.search_by_projection_frame(K2, D2, BOUNDS, q2, r2, l2, d2, v3, a2, taken0, th, True)
    assert n3 == on3 and np.array_equal(m3, om3) and np.array_equal(tk3, otk3)


def test_projection_family_sim3(oracle, frames):                          # M12 :258-361
    K1, D1, K2, D2, shift, E = frames
    q_uv, lvl, scale, valid, taken0, pred = _queries(frames, seed=7)
    rad = (10 * E.scale[pred]).astype(np.float32)
    on, om, obd, otk = oracle.search_by_projection(K2, D2, BOUNDS, q_uv, rad, D1, q_pred_level=pred, q_valid=valid, taken=taken0,
                                                   ratio=0.75, th=50, check_ori=False)
    n, m, tk = NP.search_by_projection_sim3(K2, D2, BOUNDS, q_uv, rad, pred, D1, valid, taken0)
    assert n == on and np.array_equal(m, om) and np.array_equal(tk, otk) and n > 50


def test_projection_family_fuse(oracle, frames):                          # M10 :775-811
    K1, D1, K2, D2, shift, E = frames
    q_uv, lvl, scale, valid, taken0, pred = _queries(frames, seed=8)
    rad = (3 * E.scale[pred]).astype(np.float32)
    on, om, obd, _ = oracle.search_by_projection(K2, D2, BOUNDS, q_uv, rad, D1, q_pred_level=pred, q_valid=valid, th=50,
                                                 inv_level_sigma2=E.inv_sigma2, chi2_gate=5.99, ratio=0.6, check_ori=False)
    n, m = NP.fuse_candidates(K2, D2, BOUNDS, q_uv, rad, pred, D1, valid, E.inv_sigma2)
    assert n == on and np.array_equal(m, om) and n > 50


def _fv(rng, node_of):
    nodes = np.unique(node_of); off = [0]; idx = []
    for nd in nodes:
        ii = np.nonzero(node_of == nd)[0]; idx.extend(ii.tolist()); off.append(len(idx))
    return nodes.astype(np.uint32) * 7 + 3, np.array(off, np.uint32), np.array(idx, np.uint32)


def _bow_setup(oracle, frames, seed, nnodes):
    K1, D1, K2, D2, shift, E = frames
    rng = np.random.default_rng(seed)
    node2 = rng.integers(0, nnodes, len(K2))
    bi, _, _ = oracle.hamming_best2(D1, D2)
    node1 = np.where(rng.random(len(K1)) < 0.8, node2[bi], rng.integers(0, nnodes + 3, len(K1)))     # (+3: nodes only set 1 has)
    return rng, _fv(rng, node1), _fv(rng, node2)


@pytest.mark.parametrize("strict,ratio,ori", [(False, 0.7, True), (True, 0.75, True), (False, 0.9, False)])
def test_search_by_bow(oracle, frames, strict, ratio, ori):              # M6 :151-256, M9 :470-580
    K1, D1, K2, D2, shift, E = frames
    rng, fv1, fv2 = _bow_setup(oracle, frames, 6, 40)
    valid1 = (rng.random(len(K1)) > 0.2).astype(np.uint8)
    valid2 = (rng.random(len(K2)) > 0.1).astype(np.uint8) if strict else None
    on, om = oracle.search_by_bow(D1, valid1, K1[:, 3], D2, valid2, K2[:, 3], fv1, fv2, ratio=ratio, th=50, strict=strict, check_ori=ori)
    n, m = NP.search_by_bow(D1, valid1, K1[:, 3], D2, valid2, K2[:, 3], fv1, fv2, ratio, strict, ori)
    assert n == on and np.array_equal(m, om) and n > 30


@pytest.mark.parametrize("ori", [False, True])
def test_search_for_triangulation(oracle, frames, ori):                  # M8 :582-722
    K1, D1, K2, D2, shift, E = frames
    rng, fv1, fv2 = _bow_setup(oracle, frames, 8, 25)
    um1 = (rng.random(len(K1)) > 0.3).astype(np.uint8); um2 = (rng.random(len(K2)) > 0.3).astype(np.uint8)
    t = np.array([shift[0], shift[1], 0.0]); t = t / (np.linalg.norm(t) + 1e-12)
    F12 = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]], np.float64) + rng.normal(0, 1e-4, (3, 3))
    for epi in ((-5000.0, 240.0), (300.0, 200.0)):                                                  # far / inside the image (epipole gate fires)
        on, om = oracle.search_for_triangulation(K1, D1, um1, K2, D2, um2, fv1, fv2, F12, epi, E.scale, E.sigma2, check_ori=ori)
        n, m = NP.search_for_triangulation(K1, D1, um1, K2, D2, um2, fv1, fv2, F12, epi, E.scale, E.sigma2, ori)
        assert n == on and np.array_equal(m, om)
    assert n > 10


def sim3_queries(frames, seed=9):
    """Both directions of SearchBySim3 on the two test frames: keyframe-1 features projected into keyframe 2 by the known
    shift (and back), predicted levels around the true octave, a validity mask standing in for the reference's gates."""
    K1, D1, K2, D2, shift, E = frames
    rng = np.random.default_rng(seed)
    def one(Ka, sgn):
        n = len(Ka)
        uv = (Ka[:, :2] - sgn * shift + rng.normal(0, 0.8, (n, 2))).astype(np.float32)
        pred = np.clip(Ka[:, 2].astype(np.int32) + rng.integers(0, 2, n), 0, 7).astype(np.int32)
        rad = (7.5 * E.scale[pred]).astype(np.float32)                                   # th = 7.5 (src/LoopClosing.cc:319)
        return uv, rad, pred, (rng.random(n) > 0.15).astype(np.uint8)
    return one(K1, 1.0) + one(K2, -1.0)


def test_search_by_sim3(oracle, frames):                                  # M11 :956-1159
    K1, D1, K2, D2, shift, E = frames
    q = sim3_queries(frames)
    on, om = oracle.search_by_sim3(K1, D1, K2, D2, BOUNDS, *q)
    n, m = NP.search_by_sim3(K1, D1, K2, D2, BOUNDS, *q)
    assert n == on and np.array_equal(m, om) and n > 50
    # map-point descriptors that are NOT the keyframes' own rows (a map point's representative descriptor comes from any of
    # its observations): a few bits flipped
    rng = np.random.default_rng(3)
    M1 = D1.copy(); M2 = D2.copy()
    for M in (M1, M2):
        M[np.arange(len(M)), rng.integers(0, 32, len(M))] ^= (1 << rng.integers(0, 8, len(M))).astype(np.uint8)
    on2, om2 = oracle.search_by_sim3(K1, D1, K2, D2, BOUNDS, *q, q12_desc=M1, q21_desc=M2)
    n2, m2 = NP.search_by_sim3(K1, D1, K2, D2, BOUNDS, *q, q12_desc=M1, q21_desc=M2)
    assert n2 == on2 and np.array_equal(m2, om2) and n2 > 50
    # keyframes of two cameras: each direction searches the TARGET keyframe's own grid (:1022 pKF2, :1102 pKF1)
    B2 = (BOUNDS[0] - 7.0, BOUNDS[1] + 19.0, BOUNDS[2] - 3.0, BOUNDS[3] + 11.0)
    on3, om3 = oracle.search_by_sim3(K1, D1, K2, D2, BOUNDS, *q, bounds2=B2)
    n3, m3 = NP.search_by_sim3(K1, D1, K2, D2, BOUNDS, *q, bounds2=B2)
    assert n3 == on3 and np.array_equal(m3, om3) and n3 > 50
    # agreement really filters: one-directional matches outnumber the mutual ones
    n12, m12, _, _ = oracle.search_by_projection(K2, D2, BOUNDS, q[0], q[1], D1, q_pred_level=q[2], q_valid=q[3], th=100, ratio=1.0)
    assert n12 > on
