"""GPU parity of the LocalMapping thread's device-resident steps (round 5) against the oracle's COMPOSITION of
the stages the reference runs one after the other:

  orbl_create_new_map_points = for every neighbour in order: SearchForTriangulation (oracle.search_for_triangulation,
      src/ORBmatcher.cc:582-722) on the current keyframe's keypoints that hold no map point yet, the per-match triangulation and
      gates (oracle.triangulate_matches, src/LocalMapping.cc:267-378), AddMapPoint on the accepted ones (:383) - which the next
      neighbour's search must see;
  orbl_fuse_batch = ORBmatcher::Fuse's candidate selection (oracle.search_by_projection, chi-square form, src/ORBmatcher.cc:724-842)
      per target keyframe.

Matches, accept flags and candidate indices / distances must be identical; triangulated points agree to 1e-9 relative."""
import numpy as np
import pytest

from ceres_mono_orb_slam2_amd import synth

pytestmark = pytest.mark.gpu

W, H = 1241, 376
SF = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
LS = (SF * SF).astype(np.float32)
BOUNDS = np.array([0, W, 0, H], np.float32)


def _fv(node_of):
    nodes = np.unique(node_of); off = [0]; idx = []
    for nd in nodes:
        ii = np.nonzero(node_of == nd)[0]; idx.extend(ii.tolist()); off.append(len(idx))
    return nodes.astype(np.uint32), np.array(off, np.uint32), np.array(idx, np.uint32)


def make_scene(seed, n_nb=6, npts=2600, clutter=500, nnodes=60, p_unmapped=0.6):
    """A current keyframe and n_nb neighbours looking at one cloud of 3-D points: every keyframe sees a random subset (its
    keypoints = projections + level-scaled noise, descriptors = the point's with a few bits flipped) plus clutter; the BoW node of a
    keypoint is a function of its point (so true correspondences share a node), F12 / epipoles from the poses (ComputeF12)."""
    rng = np.random.default_rng(seed)
    K = synth.KITTI_K4.astype(np.float32)
    fx, fy, cx, cy = [np.float64(v) for v in K]
    P = np.stack([rng.uniform(-25, 25, npts), rng.uniform(-4, 4, npts), rng.uniform(6, 70, npts)], 1)
    base_desc = rng.integers(0, 256, (npts, 32), dtype=np.uint8)
    node_pt = rng.integers(0, nnodes, npts) * 11 + 5

    def keyframe(C, rv):
        R = synth.quat_to_R(synth.quat_from_rotvec(rv)); t = -R @ C
        Pc = P @ R.T + t
        uv = np.stack([fx * Pc[:, 0] / Pc[:, 2] + cx, fy * Pc[:, 1] / Pc[:, 2] + cy], 1)
        vis = (Pc[:, 2] > 1) & (uv[:, 0] > 20) & (uv[:, 0] < W - 20) & (uv[:, 1] > 20) & (uv[:, 1] < H - 20) & (rng.random(npts) < 0.75)
        ids = np.nonzero(vis)[0]; rng.shuffle(ids)
        octv = rng.integers(0, 6, len(ids))
        kp = np.zeros((len(ids) + clutter, 4), np.float32)
        kp[:len(ids), :2] = uv[ids] + rng.normal(0, 0.35, (len(ids), 2)) * SF[octv][:, None]
        kp[:len(ids), 2] = octv; kp[:len(ids), 3] = rng.uniform(0, 360, len(ids))
        kp[len(ids):, 0] = rng.uniform(20, W - 20, clutter); kp[len(ids):, 1] = rng.uniform(20, H - 20, clutter)
        kp[len(ids):, 2] = rng.integers(0, 8, clutter); kp[len(ids):, 3] = rng.uniform(0, 360, clutter)
        d = np.concatenate([base_desc[ids], rng.integers(0, 256, (clutter, 32), dtype=np.uint8)])
        nflip = rng.integers(0, 7, len(d))
        for j in range(6):
            m = nflip > j
            d[m, rng.integers(0, 32, m.sum())] ^= (1 << rng.integers(0, 8, m.sum())).astype(np.uint8)
        node = np.concatenate([node_pt[ids], rng.integers(0, nnodes, clutter) * 11 + 5])
        stray = rng.random(len(node)) < 0.1                         # a tenth of the keypoints fell into another word's node
        node[stray] = rng.integers(0, nnodes, stray.sum()) * 11 + 5
        perm = rng.permutation(len(kp))                             # keypoint order is not point order
        return dict(kps=kp[perm], desc=d[perm], fv=_fv(node[perm]), unmapped=(rng.random(len(kp)) < p_unmapped).astype(np.uint8),
                    Tcw=np.hstack([R, t[:, None]]), K4=K, R=R, t=t, C=C)
    cur = keyframe(np.zeros(3), rng.normal(0, 0.01, 3))
    nbs = []
    for k in range(n_nb):
        C = np.array([0.6 + 0.5 * k, rng.normal(0, 0.05), rng.normal(0, 0.25)]) * (1 if k % 2 == 0 else -1)
        q = keyframe(C, rng.normal(0, 0.02, 3))
        # LocalMapping::ComputeF12 (src/LocalMapping.cc:507-523) and the epipole of SearchForTriangulation (src/ORBmatcher.cc:588-595)
        R12 = cur["R"] @ q["R"].T; t12 = -R12 @ q["t"] + cur["t"]
        tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
        Km = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
        q["F12"] = np.linalg.inv(Km.T) @ tx @ R12 @ np.linalg.inv(Km)
        C2 = q["R"] @ cur["C"] + q["t"]
        q["epipole"] = (np.float32(fx * C2[0] / C2[2] + cx), np.float32(fy * C2[1] / C2[2] + cy))
        nbs.append(q)
    return cur, nbs


def oracle_create_new_map_points(oracle, cur, nbs, ratio):
    n1 = len(cur["kps"])
    mask = cur["unmapped"].copy()
    M, OK, X = [], [], []
    for q in nbs:
        _, m = oracle.search_for_triangulation(cur["kps"], cur["desc"], mask, q["kps"], q["desc"], q["unmapped"], cur["fv"], q["fv"], q["F12"], q["epipole"],
                                               SF, LS, check_ori=False)
        ok = np.zeros(n1, bool); x = np.zeros((n1, 3))
        hit = np.nonzero(m >= 0)[0]
        if len(hit):
            kp1 = cur["kps"][hit][:, :3]; kp2 = q["kps"][m[hit]][:, :3]
            xx, oo = oracle.triangulate_matches(cur["Tcw"], q["Tcw"], cur["K4"], q["K4"], kp1, kp2, LS, SF, ratio)
            ok[hit] = oo.astype(bool); x[hit] = np.where(oo[:, None].astype(bool), xx, 0.0)
        mask = mask.copy(); mask[ok] = 0                               # AddMapPoint(map_point, idx1)
        M.append(m); OK.append(ok); X.append(x)
    return np.array(M), np.array(OK), np.array(X)


@pytest.mark.parametrize("seed,n_nb", [(1, 6), (2, 20), (3, 1)])
def test_create_new_map_points_vs_oracle_composition(oracle, seed, n_nb):
    from ceres_mono_orb_slam2_amd import localmapping
    cur, nbs = make_scene(seed, n_nb=n_nb)
    ratio = np.float32(1.5) * np.float32(1.2)
    m, ok, X, npr = localmapping.create_new_map_points(cur, nbs, SF, LS, ratio)
    om, ook, oX = oracle_create_new_map_points(oracle, cur, nbs, ratio)
    assert npr == n_nb
    assert np.array_equal(m, om), "SearchForTriangulation partners"
    assert np.array_equal(ok, ook), "triangulation accept flags"
    assert np.abs(X - oX).max() <= 1e-9 * max(1.0, np.abs(oX).max())
    # the scene really exercises the dependence between neighbours: later neighbours find fewer candidates because earlier ones
    # gave keypoints a map point, and a keypoint is triangulated at most once over the whole call
    assert ok.sum() > 25 * min(n_nb, 3) and (ok.sum(0) <= 1).all()
    if n_nb >= 6:
        first = ok[0]
        assert (m[1:, first] == -1).all()
        # without the mask hand-over the second neighbour would have matched some of them again
        _, m_free = oracle.search_for_triangulation(cur["kps"], cur["desc"], cur["unmapped"], nbs[1]["kps"], nbs[1]["desc"], nbs[1]["unmapped"], cur["fv"], nbs[1]["fv"],
                                                    nbs[1]["F12"], nbs[1]["epipole"], SF, LS, check_ori=False)
        assert (m_free[first] >= 0).sum() > 10


def test_create_new_map_points_degenerate_inputs(oracle):
    from ceres_mono_orb_slam2_amd import localmapping
    cur, nbs = make_scene(5, n_nb=3)
    ratio = np.float32(1.8)
    # no neighbours; a neighbour without keypoints / without feature vector; all keypoints of the current keyframe mapped
    m, ok, X, npr = localmapping.create_new_map_points(cur, [], SF, LS, ratio)
    assert m.shape[0] == 0 and npr == 0
    e = dict(nbs[0]); e["kps"] = np.zeros((0, 4), np.float32); e["desc"] = np.zeros((0, 32), np.uint8); e["unmapped"] = np.zeros(0, np.uint8)
    e["fv"] = (np.zeros(0, np.uint32), np.zeros(1, np.uint32), np.zeros(0, np.uint32))
    m, ok, X, npr = localmapping.create_new_map_points(cur, [e, nbs[1]], SF, LS, ratio)
    om, ook, oX = oracle_create_new_map_points(oracle, cur, [e, nbs[1]], ratio)
    assert npr == 2 and (m[0] == -1).all() and np.array_equal(m, om) and np.array_equal(ok, ook)
    c2 = dict(cur); c2["unmapped"] = np.zeros(len(cur["kps"]), np.uint8)
    m, ok, X, npr = localmapping.create_new_map_points(c2, nbs, SF, LS, ratio)
    assert (m == -1).all() and not ok.any()
    c3 = dict(cur); c3["unmapped"] = None                               # NULL = every keypoint is a candidate
    m, ok, X, npr = localmapping.create_new_map_points(c3, nbs, SF, LS, ratio)
    c3o = dict(cur); c3o["unmapped"] = np.ones(len(cur["kps"]), np.uint8)
    om, ook, oX = oracle_create_new_map_points(oracle, c3o, nbs, ratio)
    assert np.array_equal(m, om) and np.array_equal(ok, ook)


def test_create_new_map_points_stop_flag(oracle):
    """CheckNewKeyFrames (src/LocalMapping.cc:227): a flag that is already up stops the call after the FIRST neighbour (the check is
    skipped for i = 0), whose results are complete; nothing of the later neighbours is produced."""
    from ceres_mono_orb_slam2_amd import localmapping
    cur, nbs = make_scene(7, n_nb=5)
    ratio = np.float32(1.8)
    stop = np.ones(1, np.uint8)
    m, ok, X, npr = localmapping.create_new_map_points(cur, nbs, SF, LS, ratio, stop=stop)
    om, ook, oX = oracle_create_new_map_points(oracle, cur, nbs[:1], ratio)
    assert npr == 1
    assert np.array_equal(m[0], om[0]) and np.array_equal(ok[0], ook[0]) and ok[0].sum() > 15
    assert (m[1:] == -1).all() and not ok[1:].any()
    stop[0] = 0
    m2, ok2, X2, npr2 = localmapping.create_new_map_points(cur, nbs, SF, LS, ratio, stop=stop)
    assert npr2 == 5 and ok2[1:].any()


@pytest.mark.parametrize("seed", [11, 12])
def test_fuse_batch_vs_oracle(oracle, seed):
    """ORBmatcher::Fuse's candidate selection for 12 target keyframes x ~1200 map points in one call, against the oracle run keyframe
    by keyframe (different grids: two of the keyframes come from a camera with other image bounds)."""
    from ceres_mono_orb_slam2_amd import localmapping
    cur, nbs = make_scene(seed, n_nb=12, npts=2200)
    rng = np.random.default_rng(seed)
    Mq = 1200
    inv_ls = (1.0 / LS).astype(np.float32)
    kfs = []; uv = []; rad = []; lvl = []
    # "map points": keypoints of the current keyframe carried into every target by a per-target shift of the target's own keypoints
    # (what matters here is the window / level / chi-square / distance selection, not the projection, which stays with the caller)
    src = rng.choice(len(cur["kps"]), Mq, replace=False)
    mp_desc = cur["desc"][src].copy()
    for t, q in enumerate(nbs):
        b = BOUNDS.copy()
        if t % 5 == 3: b = np.array([-12.5, W + 9.0, -7.0, H + 4.5], np.float32)
        pick = rng.integers(0, len(q["kps"]), Mq)
        u = q["kps"][pick, :2] + rng.normal(0, 1.2, (Mq, 2)).astype(np.float32) * SF[q["kps"][pick, 2].astype(int)][:, None]
        l = np.minimum(q["kps"][pick, 2].astype(np.int32) + rng.integers(0, 2, Mq), 7).astype(np.int32)
        l[rng.random(Mq) < 0.15] = -1                                  # a gate failed on the caller's side
        # half of the queries look for the keypoint's own descriptor (so that distances <= TH_LOW exist), half for the map point's
        own = rng.random(Mq) < 0.5
        if t == 0: mp_desc[own] = q["desc"][pick[own]]
        kfs.append(dict(kps=q["kps"], desc=q["desc"], bounds=b)); uv.append(u.astype(np.float32)); rad.append((3.0 * SF[np.maximum(l, 0)]).astype(np.float32)); lvl.append(l)
    uv = np.array(uv); rad = np.array(rad); lvl = np.array(lvl)
    bi, bd = localmapping.fuse_batch(kfs, uv, rad, lvl, mp_desc, inv_ls)
    hits = 0
    for t, q in enumerate(kfs):
        n, om, obd, _ = oracle.search_by_projection(q["kps"], q["desc"], q["bounds"], uv[t], rad[t], mp_desc, q_pred_level=lvl[t], q_valid=(lvl[t] >= 0).astype(np.uint8),
                                                    inv_level_sigma2=inv_ls, chi2_gate=5.99, th=256)
        assert np.array_equal(bi[t], om), "keyframe %d: candidate indices" % t
        assert np.array_equal(bd[t], obd), "keyframe %d: candidate distances" % t
        hits += int((om >= 0).sum())
    assert hits > 12 * 300 and (bd[0] <= 50).sum() > 100
    # the Sim(3) form of LoopClosing::SearchAndFuse (src/ORBmatcher.cc:844-954): the same selection WITHOUT the chi-square gate
    bi3, bd3 = localmapping.fuse_batch(kfs, uv, rad, lvl, mp_desc, None, sim3=True)
    more = 0
    for t, q in enumerate(kfs):
        n, om, obd, _ = oracle.search_by_projection(q["kps"], q["desc"], q["bounds"], uv[t], rad[t], mp_desc, q_pred_level=lvl[t], q_valid=(lvl[t] >= 0).astype(np.uint8), th=256)
        assert np.array_equal(bi3[t], om) and np.array_equal(bd3[t], obd), "keyframe %d: Sim(3) form" % t
        more += int((bi3[t] != bi[t]).sum())
    assert more > 50                                                    # (the gate does decide some of the candidates)
    # empty inputs
    bi0, bd0 = localmapping.fuse_batch([], np.zeros((0, 5, 2), np.float32), np.zeros((0, 5), np.float32), np.zeros((0, 5), np.int32), mp_desc[:5], inv_ls)
    assert bi0.shape[0] == 0
