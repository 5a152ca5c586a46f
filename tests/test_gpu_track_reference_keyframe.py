"""GPU parity of the device-resident TrackReferenceKeyFrame step (csrc/orb_track.hip, include/orbslam_hip.h::
orbt_track_reference_keyframe; reference src/Tracking.cc:566-615) against the CPU oracle's COMPOSITION at 1241 x 376: the oracle's
extractor, Frame::ComputeBoW = TemplatedVocabulary::transform (oracle bow_transform on a synthetic k-ary tree), ORBmatcher::SearchByBoW
(KeyFrame*, Frame&) (oracle search_by_bow: per-node greedy pass, nnratio 0.7, TH_LOW 50, rotation histogram) and PoseOptimization
on the matched features in feature order.  Keypoints, descriptors, BowVector (bit patterns), FeatureVector, matches, slot owners
and outlier flags identical, pose 1e-7; also with the frame left on the device by an earlier call (img == NULL)."""
import numpy as np
import pytest

from ceres_mono_orb_slam2_amd import synth
from tests.test_gpu_track import _scenario, K4, BOUNDS

pytestmark = pytest.mark.gpu
KEYS = ("node_desc", "child_off", "children", "word_id", "weight", "L")


def _expected(oracle, voc, S, kf_valid, ratio=0.7, check_ori=True):
    E = S["E"]
    kps, desc = E.extract(S["img"])
    bw, bv, fn, fo, fi = oracle.bow_transform(voc, desc, 4)
    kbw, kbv, kfn, kfo, kfi = oracle.bow_transform(voc, S["desc"], 4)
    nm, m = oracle.search_by_bow(S["desc"], kf_valid, S["angle"], desc, None, kps["angle"].astype(np.float32), (kfn, kfo, kfi), (fn, fo, fi), ratio=ratio, th=50,
                                 strict=False, check_ori=check_ori)
    owner = np.full(len(kps), -1, np.int32)
    for q in range(len(m)):
        if m[q] >= 0: owner[m[q]] = q
    feat = np.nonzero(owner >= 0)[0]
    pose0 = oracle.matrix4d_to_pose7(S["T"])
    outl = np.zeros(len(kps), bool)
    if len(feat) >= 3:
        ninl, pose, out, _ = oracle.pose_optimization(K4.astype(np.float64), pose0, S["X"][owner[feat]], np.stack([kps["x"][feat], kps["y"][feat]], 1).astype(np.float64),
                                                      E.inv_sigma2[kps["octave"][feat].astype(int)])
        outl[feat] = out.astype(bool)
    else:
        ninl, pose = 0, pose0
    return dict(kps=kps, desc=desc, bow=(bw, bv), fv=(fn, fo, fi), kf_fv=(kfn, kfo, kfi), match=m, nmatches=nm, owner=owner, outlier=outl, pose7=pose, n_inliers=int(ninl), ncorr=len(feat))


@pytest.mark.parametrize("seed,k,L,check_ori", [(3, 6, 6, True), (4, 10, 5, True), (5, 6, 6, False), (6, 4, 6, True)])
def test_track_reference_keyframe_vs_oracle_composition(oracle, seed, k, L, check_ori):
    from ceres_mono_orb_slam2_amd import ORBextractor, tracking
    from ceres_mono_orb_slam2_amd.vocabulary import ORBVocabulary
    S = _scenario(oracle, seed)
    voc = synth.make_vocabulary(seed, k=k, L=L)
    V = ORBVocabulary(*[voc[x] for x in KEYS])
    kf_valid = (S["valid"] != 0).astype(np.uint8)
    want = _expected(oracle, voc, S, kf_valid, check_ori=check_ori)
    ex = ORBextractor(2000, 1.2, 8, 20, 7)
    got = tracking.track_reference_keyframe(ex, V, S["img"], K4, BOUNDS, S["T"], S["desc"], kf_valid, S["angle"], S["X"], want["kf_fv"], 0.7, check_ori)
    assert want["nmatches"] > 100, want["nmatches"]
    assert np.array_equal(got["kps"], want["kps"]) and np.array_equal(got["desc"], want["desc"])
    assert np.array_equal(got["bow"][0], want["bow"][0]) and np.array_equal(got["bow"][1].view(np.uint64), want["bow"][1].view(np.uint64))
    assert all(np.array_equal(a, b) for a, b in zip(got["fv"], want["fv"]))
    assert np.array_equal(got["match"], want["match"]) and got["nmatches"] == want["nmatches"]
    assert np.array_equal(got["owner"], want["owner"]) and got["n_correspondences"] == want["ncorr"]
    assert np.array_equal(got["outlier"], want["outlier"]) and got["n_inliers"] == want["n_inliers"]
    assert np.allclose(got["pose7"], want["pose7"], rtol=0, atol=1e-7)
    # the same on the frame that call left on the device
    again = tracking.track_reference_keyframe(ex, V, None, K4, BOUNDS, S["T"], S["desc"], kf_valid, S["angle"], S["X"], want["kf_fv"], 0.7, check_ori)
    assert np.array_equal(again["match"], got["match"]) and np.array_equal(again["owner"], got["owner"]) and np.array_equal(again["pose7"], got["pose7"])
    print("TrackReferenceKeyFrame: %d keypoints, %d keyframe features in %d nodes, %d matches, %d inliers" % (len(got["kps"]), len(kf_valid), len(want["kf_fv"][0]), got["nmatches"], got["n_inliers"]))


def test_relocalization_search_by_bow_all_candidates_in_one_call(oracle):
    """Tracking::Relocalization's first stage (src/Tracking.cc:979-1029): ComputeBoW + SearchByBoW(keyframe, frame) with ORBmatcher(0.75,
    true) for five candidate keyframes in ONE call (orbt_relocalization_search_by_bow) against the oracle, candidate by candidate; the
    candidates differ (other validity masks, perturbed descriptors, one without usable points, one empty)."""
    from ceres_mono_orb_slam2_amd import ORBextractor, tracking
    from ceres_mono_orb_slam2_amd.vocabulary import ORBVocabulary
    S = _scenario(oracle, 8)
    voc = synth.make_vocabulary(8, k=6, L=6)
    V = ORBVocabulary(*[voc[x] for x in KEYS])
    rng = np.random.default_rng(8)
    E = S["E"]
    kps, desc = E.extract(S["img"])
    bw, bv, fn, fo, fi = oracle.bow_transform(voc, desc, 4)
    cands = []
    for i in range(5):
        d = S["desc"].copy(); valid = (S["valid"] != 0).astype(np.uint8); ang = S["angle"].copy()
        if i == 1: valid &= (rng.random(len(valid)) < 0.6).astype(np.uint8)
        if i == 2:
            flip = rng.random(len(d)) < 0.5
            d[flip, rng.integers(0, 32, flip.sum())] ^= (1 << rng.integers(0, 8, flip.sum())).astype(np.uint8); ang = (ang + 40.0 * (rng.random(len(ang)) < 0.2)).astype(np.float32) % 360
        if i == 3: valid[:] = 0
        if i == 4: d = d[:0]; valid = valid[:0]; ang = ang[:0]
        _, _, kfn, kfo, kfi = oracle.bow_transform(voc, d, 4) if len(d) else (None, None, np.zeros(0, np.uint32), np.zeros(1, np.uint32), np.zeros(0, np.uint32))
        cands.append(dict(desc=d, valid=valid, angle=ang, fv=(kfn, kfo, kfi)))
    ex = ORBextractor(2000, 1.2, 8, 20, 7)
    got = tracking.relocalization_search_by_bow(ex, V, S["img"], K4, BOUNDS, cands, 0.75, True)
    assert np.array_equal(got["kps"], kps) and np.array_equal(got["desc"], desc)
    assert np.array_equal(got["bow"][0], bw) and np.array_equal(got["bow"][1].view(np.uint64), bv.view(np.uint64))
    assert all(np.array_equal(a, b) for a, b in zip(got["fv"], (fn, fo, fi)))
    total = 0
    for i, q in enumerate(cands):
        if len(q["desc"]):
            nm, m = oracle.search_by_bow(q["desc"], q["valid"], q["angle"], desc, None, kps["angle"].astype(np.float32), q["fv"], (fn, fo, fi), ratio=0.75, th=50, strict=False, check_ori=True)
        else:
            nm, m = 0, np.zeros(0, np.int32)
        owner = np.full(len(kps), -1, np.int32)
        for k in range(len(m)):
            if m[k] >= 0: owner[m[k]] = k
        assert int(got["nmatches"][i]) == nm, (i, int(got["nmatches"][i]), nm)
        assert np.array_equal(got["owner"][i], owner), "candidate %d" % i
        total += nm
    assert got["nmatches"][0] > 100 and got["nmatches"][3] == 0 and got["nmatches"][4] == 0 and got["nmatches"][1] < got["nmatches"][0]
    # the same on the resident frame
    again = tracking.relocalization_search_by_bow(ex, V, None, K4, BOUNDS, cands, 0.75, True)
    assert np.array_equal(again["owner"], got["owner"]) and np.array_equal(again["nmatches"], got["nmatches"])
