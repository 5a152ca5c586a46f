"""CPU checks of the covisibility-structured BA graph generator (synth.make_ba_graph_covis, round 6): the reduced-system structures the
bench's `c4_covis` / `c4_dense` / `c5_loop` legs and tests/test_gpu_ba_structures.py rely on are what the docstring says, the geometry is
valid (every observation in front of and inside the image of its keyframe), and the oracle descends on them.
Reference: src/CeresOptimizer.cc:353-363 (who is in a local map), src/LoopClosing.cc:656 (GlobalBA after a loop closure)."""
import numpy as np
import pytest

from ceres_mono_orb_slam2_amd import synth


def _share(g):
    ncam, npts = len(g["cam_fixed"]), len(g["pts0"])
    M = np.zeros((npts, ncam), np.float32); M[g["obs_pt"], g["obs_cam"]] = 1
    return M.T @ M


def _envelope(g):
    import bench_ba
    return bench_ba.skyline_tiles(g)


@pytest.mark.parametrize("structure,ncam,npts,nobs", [("covis", 100, 6000, 30000), ("dense", 60, 3000, 15000), ("loop", 120, 4000, 20000)])
def test_structures_are_what_they_claim(structure, ncam, npts, nobs):
    g = synth.make_ba_graph_covis(7, ncam, npts, nobs, structure=structure)
    assert len(g["obs_cam"]) == nobs and len(g["pts0"]) == npts and g["current"] == ncam - 1
    assert np.bincount(g["obs_pt"], minlength=npts).min() >= 2                       # every landmark has a track
    assert len(set(zip(g["obs_cam"].tolist(), g["obs_pt"].tolist()))) == nobs         # no keyframe sees a landmark twice
    # geometry: positive depth, inside the 1241 x 376 image (ground truth, before noise)
    for c in (0, ncam // 2, ncam - 1):
        m = g["obs_cam"] == c
        uv, z = synth.project(g["K4"][c], g["poses_gt"][c], g["pts_gt"][g["obs_pt"][m]])
        assert z.min() > 4.0 and uv[:, 0].min() > 0 and uv[:, 0].max() < 1241 and uv[:, 1].min() > 0 and uv[:, 1].max() < 376
    S = _share(g)
    inside, lower, upd, dense_upd = _envelope(g)
    if structure == "covis":
        assert S[ncam - 1][: ncam - 1].min() >= 15                                    # the current keyframe shares >= 15 landmarks with EVERY local keyframe
        far = np.array([S[a, b] for a in range(ncam - 1) for b in range(a + 31, ncam - 1)])
        assert far.max() == 0                                                          # otherwise a window of <= 30 keyframes: a band
        assert inside < lower and upd < dense_upd // 2
    elif structure == "dense":
        assert (S > 0).mean() > 0.95 and inside == lower and upd == dense_upd          # a full reduced system
    else:
        assert S[0, ncam - 1] > 0 and S[1, ncam - 2] > 0                               # the ends of the chain are tied
        mid = np.array([S[a, b] for a in range(ncam) for b in range(a + 31, ncam) if (a + ncam - b) > 31])
        assert mid.max() == 0                                                          # nothing else leaves the window
        assert inside < lower


def test_oracle_descends_on_every_structure(oracle):
    for st in synth.BA_STRUCTURES:
        g = synth.make_ba_graph_covis(11, 14, 300, 1500, structure=st, window=(4, 8), cur_share=15)
        a = (g["K4"], g["poses0"], g["cam_fixed"], np.ones(14, np.uint8), g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
        rc, poses, pts, er, s1, s2 = oracle.local_ba(*a)
        assert rc == 0 and s1["iterations"] == 5 and s1["final_cost"] < 0.6 * s1["initial_cost"]
        assert 0 < er.sum() < 0.2 * len(er)                                            # the gross outliers go, the rest stays


def test_generator_is_deterministic():
    a = synth.make_ba_graph_covis(5, 20, 200, 1000, structure="covis", window=(4, 8))
    b = synth.make_ba_graph_covis(5, 20, 200, 1000, structure="covis", window=(4, 8))
    assert all(np.array_equal(a[k], b[k]) for k in ("obs_cam", "obs_pt", "obs_uv", "poses0", "pts0"))


def test_shuffle_keyframes_is_a_relabelling():
    g = synth.make_ba_graph(3, ncam=40, npts=800, nobs=4000, n_fixed=1)
    h = synth.shuffle_keyframes(g, 7)
    perm = h["kf_perm"]
    assert sorted(perm.tolist()) == list(range(40)) and not np.array_equal(perm, np.arange(40))
    assert np.array_equal(h["poses0"], g["poses0"][perm]) and np.array_equal(h["cam_fixed"], g["cam_fixed"][perm])
    assert np.array_equal(perm[h["obs_cam"]], g["obs_cam"])                       # every observation still names the same keyframe
    assert np.array_equal(h["obs_pt"], g["obs_pt"]) and np.array_equal(h["obs_uv"], g["obs_uv"])
    assert int(h["cam_fixed"].sum()) == 1
