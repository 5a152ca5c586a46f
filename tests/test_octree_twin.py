"""The level-synchronous array formulation of DistributeOctTree (what the HIP kernel k_octree runs)
must equal the literal std::list restatement in the oracle on arbitrary candidate sets."""
import numpy as np

from ceres_mono_orb_slam2_amd import synth
from tests.octree_twin import octree_twin


def _rand_cands(rng, n, W, H, mode):
    xy = set(); tries = 0
    while len(xy) < n and tries < 20 * n + 100:
        tries += 1
        if mode == 0:
            x, y = rng.integers(3, W - 3), rng.integers(3, H - 3)
        elif mode == 1:
            cx, cy = rng.choice([0.1, 0.5, 0.9]) * W, rng.choice([0.2, 0.7]) * H
            x = int(np.clip(rng.normal(cx, W / 20), 3, W - 4)); y = int(np.clip(rng.normal(cy, H / 10), 3, H - 4))
        else:
            x = int(rng.integers(3, min(60, W - 3))); y = int(rng.integers(3, H - 3))
        xy.add((int(x), int(y)))
    pts = sorted(xy, key=lambda p: (p[1] // 32, p[0] // 31, p[1], p[0]))
    return np.array([(x, y, int(rng.integers(7, 60))) for x, y in pts], np.int32).reshape(-1, 3)


def test_twin_equals_oracle_on_random_sets(oracle):
    rng = np.random.default_rng(0)
    for trial in range(150):
        W, H = [(1209, 344), (608, 448), (314, 73), (100, 300)][trial % 4]
        n = int(rng.choice([0, 1, 2, 3, 7, 50, 200, 433, 434, 435, 800, 2000]))
        n = min(n, (W - 6) * (H - 6) // 3)
        N = int(rng.choice([5, 60, 122, 217, 434]))
        c = _rand_cands(rng, n, W, H, trial % 3)
        a = oracle.octree(c, 16, 16 + W, 16, 16 + H, N)
        b = octree_twin(c, 16, 16 + W, 16, 16 + H, N)
        assert np.array_equal(a, b), (trial, W, H, len(c), N)


def test_twin_equals_oracle_on_real_candidates(oracle):
    E = oracle.OracleExtractor(1000)
    for seed, fam in [(0, "blocks"), (1, "checker")]:
        E.extract(synth.make_frame(seed, 640, 480, fam))
        for l in range(8):
            c = E.level_candidates(l)
            h, w = E.level_image(l).shape
            k = E.level_keypoints(l)
            exp = np.stack([k["x"] - 16, k["y"] - 16, k["response"]], 1).astype(np.int32).reshape(-1, 3)
            assert np.array_equal(octree_twin(c, 16, w - 16, 16, h - 16, int(E.quota[l])), exp)
