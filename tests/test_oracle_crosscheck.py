"""Cross-checks of the C++ oracle against independent numpy / pure-Python restatements
(no reference golden vectors exist: SURVEY.md section 4 and 8(c))."""
import numpy as np
import pytest

from ceres_mono_orb_slam2_amd import synth
from tests import npref


@pytest.mark.parametrize("shape,dst", [((376, 1241), (1034, 313)), ((480, 640), (533, 400)), ((61, 97), (81, 51)),
                                       ((40, 50), (50, 40)), ((33, 47), (120, 90))])
def test_resize_matches_numpy(oracle, shape, dst):
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, shape, dtype=np.uint8)
    got = oracle.resize_linear_u8(src, dst[0], dst[1])
    exp = npref.np_resize_linear_u8(src, dst[0], dst[1])
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("shape", [(376, 1241), (105, 346), (7, 9), (16, 5)])
def test_blur_matches_numpy(oracle, shape):
    rng = np.random.default_rng(2)
    src = rng.integers(0, 256, shape, dtype=np.uint8)
    if min(shape) < 4:
        pytest.skip("reflect pad needs >3")
    assert np.array_equal(oracle.gaussian_blur7(src), npref.np_gaussian_blur7(src))
    flat = np.full(shape, 255, np.uint8)
    assert (oracle.gaussian_blur7(flat) == 255).all()     # taps sum to 257: saturates, does not wrap


@pytest.mark.parametrize("seed,t", [(0, 20), (1, 7), (2, 7), (3, 20)])
def test_fast_matches_literal_definition(oracle, seed, t):
    fam = ["blocks", "checker", "flat", "blocks"][seed]
    img = synth.make_frame(seed, 64, 48, fam)
    got = oracle.fast(img, t)
    exp = npref.py_fast_nms(img, t)
    assert np.array_equal(got, exp)


def test_fast_threshold_monotone_property(oracle):
    """K20 == {p in K7 : score >= 20}   -- the single-score-map identity the HIP kernel uses."""
    for seed in range(4):
        img = synth.make_frame(10 + seed, 96, 64, "blocks")
        k7 = oracle.fast(img, 7)
        k20 = oracle.fast(img, 20)
        assert np.array_equal(k7[k7[:, 2] >= 20], k20)


def test_fast_tiny_subimage_is_empty(oracle):
    img = np.random.default_rng(0).integers(0, 256, (6, 40), dtype=np.uint8)
    assert len(oracle.fast(img, 7)) == 0


def test_ic_angle_and_atan2(oracle):
    rng = np.random.default_rng(3)
    img = synth.make_frame(5, 80, 80, "blocks")
    um = oracle.OracleExtractor(1000).umax
    for _ in range(50):
        x, y = rng.integers(19, 60, 2)
        m01, m10 = npref.np_ic_angle_moments(img, x, y, um)
        a = oracle.ic_angle(img, x, y)
        assert a == oracle.fast_atan2(np.float32(m01), np.float32(m10))
        ref = np.degrees(np.arctan2(m01, m10)) % 360.0
        d = abs(a - ref); d = min(d, 360 - d)
        assert d < 0.3                          # fastAtan2's documented accuracy (~0.3 deg)
    assert oracle.fast_atan2(0, 0) == 0.0
    assert oracle.fast_atan2(0, -1) == 180.0
    assert abs(oracle.fast_atan2(1, 0) - 90.0) < 1e-4
    assert abs(oracle.fast_atan2(-1, 0) - 270.0) < 1e-4


def test_det_sincos_matches_libm_float(oracle):
    """The canonical sincos must be indistinguishable from cosf/sinf at float precision."""
    ang = np.linspace(0, 360, 200001, dtype=np.float32)
    rad = (ang * np.float32(np.pi / 180.0)).astype(np.float32)
    s = np.empty(len(rad), np.float32); c = np.empty(len(rad), np.float32)
    for i, r in enumerate(rad[::97]):
        sd, cd = oracle.det_sincos(float(r))
        assert abs(sd - np.sin(np.float64(r))) < 4e-16 and abs(cd - np.cos(np.float64(r))) < 4e-16
        assert np.float32(sd) == np.sin(np.float64(r)).astype(np.float32)
        assert np.float32(cd) == np.cos(np.float64(r)).astype(np.float32)


def test_brief_matches_numpy(oracle):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    pat = oracle.pattern().reshape(512, 2).astype(np.float32)
    for ang in [0.0, 33.3, 90.0, 181.7, 359.9]:
        d = oracle.brief(img, 32, 31, ang)
        rad = np.float32(ang) * np.float32(np.pi / 180.0)
        a = np.float32(np.cos(np.float64(rad))); b = np.float32(np.sin(np.float64(rad)))
        fy = (pat[:, 0] * b).astype(np.float32) + (pat[:, 1] * a).astype(np.float32)
        fx = (pat[:, 0] * a).astype(np.float32) - (pat[:, 1] * b).astype(np.float32)
        vals = img[31 + np.rint(fy).astype(int), 32 + np.rint(fx).astype(int)].astype(int)
        bits = (vals[0::2] < vals[1::2]).astype(np.uint8)
        exp = np.packbits(bits.reshape(32, 8), axis=1, bitorder="little").ravel()
        assert np.array_equal(d, exp)


def test_octree_properties(oracle):
    rng = np.random.default_rng(5)
    for trial in range(6):
        W, H = 1209, 344
        n = [0, 1, 5, 300, 3000, 9000][trial]
        xy = set()
        while len(xy) < n:
            xy.add((int(rng.integers(3, W - 3)), int(rng.integers(3, H - 3))))
        c = np.array([(x, y, int(rng.integers(7, 200))) for x, y in sorted(xy, key=lambda p: (p[1], p[0]))],
                     np.int32).reshape(-1, 3)
        N = 434
        out = oracle.octree(c, 16, 16 + W, 16, 16 + H, N)
        cs = {tuple(r) for r in c.tolist()}
        assert all(tuple(r) in cs for r in out.tolist())
        assert len({(r[0], r[1]) for r in out.tolist()}) == len(out)
        assert len(out) <= min(n, N + 3)
        if n >= N * 4:
            assert len(out) >= N
        if n <= 5:
            assert len(out) == n


def test_extract_invariants(oracle):
    E = oracle.OracleExtractor(1000)
    img = synth.make_frame(6, 640, 480, "blocks")
    kps, desc = E.extract(img)
    assert len(kps) == len(desc) and len(kps) > 500
    assert (np.diff(kps["octave"]) >= 0).all()                  # levels concatenated 0..7 (:1075-1104)
    assert (kps["class_id"] == -1).all()
    assert ((kps["angle"] >= 0) & (kps["angle"] < 360.0001)).all()
    for l in range(8):
        lk = E.level_keypoints(l)
        w, h = E.level_image(l).shape[::-1]
        assert ((lk["x"] >= 19) & (lk["x"] <= w - 20) & (lk["y"] >= 19) & (lk["y"] <= h - 20)).all()
        assert len(lk) <= E.quota[l] + 3
    kps2, desc2 = E.extract(img)
    assert np.array_equal(kps, kps2) and np.array_equal(desc, desc2)     # deterministic / stateless per call


def test_hamming_best2_vs_numpy(oracle):
    rng = np.random.default_rng(7)
    q = rng.integers(0, 256, (50, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (70, 32), dtype=np.uint8)
    t[10] = q[3]; t[20] = q[3]                                    # exact tie: first index must win
    bi, bd, sd = oracle.hamming_best2(q, t)
    D = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2)
    assert np.array_equal(bd, D.min(1))
    assert np.array_equal(bi, D.argmin(1))
    assert bi[3] == 10 and bd[3] == 0 and sd[3] == 0
    srt = np.sort(D, axis=1)
    assert np.array_equal(sd, srt[:, 1])
    # CSR candidate form, ragged incl. empty rows
    off = np.array([0, 0, 3, 3, 10], np.uint32); idx = np.array([5, 1, 5, 9, 8, 7, 6, 5, 4, 3], np.uint32)
    bi, bd, sd = oracle.hamming_best2(q[:4], t, off, idx)
    assert bi[0] == -1 and bd[0] == 256 and sd[0] == 256 and bi[2] == -1
    assert bd[1] == min(D[1, 5], D[1, 1]) and sd[1] == sorted([D[1, 5], D[1, 1], D[1, 5]])[1]


def test_three_maxima_and_rot_bin(oracle):
    assert oracle.three_maxima(np.array([0] * 30)).tolist() == [-1, -1, -1]
    c = np.zeros(30, np.int32); c[4] = 100; c[7] = 9; c[9] = 50
    assert oracle.three_maxima(c).tolist() == [4, 9, -1]            # 9 < 0.1*100 dropped
    c[7] = 10
    assert oracle.three_maxima(c).tolist() == [4, 9, 7]
    c[:] = 5
    assert oracle.three_maxima(c).tolist() == [0, 1, 2]             # ties: earlier bin wins
    assert oracle.rot_bin(10.0, 350.0) == 1 and oracle.rot_bin(0.0, 0.0) == 0
    assert oracle.rot_bin(359.0, 0.0) == 12                          # round(359/30)=12
    assert oracle.rot_bin(355.0, 0.0) == 12 and oracle.rot_bin(0.0, 1.0) == 12


def test_search_for_initialization_basic(oracle):
    E = oracle.OracleExtractor(1000)
    frames, offs = synth.make_sequence(11, 640, 480, 2, "blocks", max_shift=6)
    k1, d1 = E.extract(frames[0]); k2, d2 = E.extract(frames[1])
    f = lambda k: np.stack([k["x"], k["y"], k["octave"].astype(np.float32), k["angle"]], 1)
    bounds = np.array([0, 640, 0, 480], np.float32)
    prev = np.stack([k1["x"], k1["y"]], 1)
    m, n, pm = oracle.search_for_initialization(f(k1), d1, f(k2), d2, bounds, prev)
    assert n == int((m >= 0).sum()) and n > 50
    mi = np.nonzero(m >= 0)[0]
    assert (k1["octave"][mi] == 0).all()
    assert len(set(m[mi].tolist())) == len(mi)                       # one-to-one after stealing
    shift = offs[1] - offs[0]
    dxy = np.stack([k2["x"][m[mi]] - k1["x"][mi], k2["y"][m[mi]] - k1["y"][mi]], 1)
    good = (np.abs(dxy + shift) <= 1.5).all(1).mean()
    assert good > 0.9
