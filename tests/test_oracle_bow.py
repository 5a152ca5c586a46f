"""Pins oracle/bow_oracle.cpp (DBoW2 transform, reference lib/DBoW2/DBoW2/TemplatedVocabulary.h:1124-1260) with a numpy
brute-force descent and a dictionary accumulation."""
import numpy as np

from oracle import pyoracle as po
from ceres_mono_orb_slam2_amd import synth


def _descend_np(voc, f, levelsup):
    cur, level, nid = 0, 0, 0
    while voc["child_off"][cur + 1] > voc["child_off"][cur]:
        ch = voc["children"][voc["child_off"][cur]:voc["child_off"][cur + 1]]
        d = np.unpackbits(voc["node_desc"][ch] ^ f, axis=1).sum(1)
        cur = int(ch[int(np.argmin(d))])                               # argmin = first minimum = strict '<' scan
        level += 1
        if level == voc["L"] - levelsup:
            nid = cur
    return int(voc["word_id"][cur]), float(voc["weight"][cur]), nid


def test_transform_against_numpy():
    for seed, k, L, ragged, levelsup in [(0, 10, 3, 0.0, 1), (1, 6, 4, 0.3, 2), (2, 10, 3, 0.0, 4)]:
        voc = synth.make_vocabulary(seed, k=k, L=L, ragged=ragged)
        rng = np.random.default_rng(seed)
        leaves = np.nonzero(voc["word_id"] >= 0)[0]
        d = voc["node_desc"][rng.choice(leaves, 300)] ^ rng.integers(0, 256, (300, 32), dtype=np.uint8) & rng.integers(0, 256, (300, 32), dtype=np.uint8) & 0x11
        bw, bv, fn, fo, fi = po.bow_transform(voc, d, levelsup)
        acc, fv = {}, {}
        for i, f in enumerate(d):
            w, wt, nid = _descend_np(voc, f, levelsup)
            if wt > 0:
                acc[w] = acc.get(w, 0.0) + wt
                fv.setdefault(nid, []).append(i)
        ws = sorted(acc)
        vals = np.array([acc[w] for w in ws]); vals = vals / np.abs(vals).sum()
        assert list(bw) == ws and np.allclose(bv, vals, rtol=1e-14, atol=0)
        assert list(fn) == sorted(fv)
        for m, nid in enumerate(fn):
            assert list(fi[fo[m]:fo[m + 1]]) == fv[nid]
        assert (voc["weight"][leaves] == 0).any()                      # stopped words exist and are skipped


def test_l1_score_properties():
    voc = synth.make_vocabulary(3, k=10, L=3)
    rng = np.random.default_rng(5)
    d1 = rng.integers(0, 256, (500, 32), dtype=np.uint8); d2 = rng.integers(0, 256, (500, 32), dtype=np.uint8)
    a = po.bow_transform(voc, d1, 1); b = po.bow_transform(voc, d2, 1)
    assert abs(po.bow_score_l1(a[0], a[1], a[0], a[1]) - 1.0) < 1e-12
    s = po.bow_score_l1(a[0], a[1], b[0], b[1])
    da = dict(zip(a[0], a[1])); db = dict(zip(b[0], b[1]))
    l1 = sum(abs(da.get(w, 0) - db.get(w, 0)) for w in set(da) | set(db))
    assert abs(s - (1 - 0.5 * l1)) < 1e-12 and s == po.bow_score_l1(b[0], b[1], a[0], a[1])
