"""The merge half of Frame::ComputeBoW pinned against the REFERENCE's own code.

oracle/_ref/libdbow2_ref.so = /root/reference/lib/DBoW2/DBoW2/{BowVector,FeatureVector}.cpp compiled where they lie
(+ oracle/ref_dbow2_shim.cpp; `make -C oracle ref`).  tests/golden/bow_merge_ref.npz holds inputs and the outputs THAT library
produced (tests/golden/make_golden_bow_ref.py), so the check also runs where the reference tree is absent."""
import os

import numpy as np
import pytest

from oracle import pyoracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bow_merge_ref.npz")
OUT = ("bow_word", "bow_value", "fv_node", "fv_off", "fv_idx")


def _same(a, b):
    return all(x.dtype == y.dtype and x.shape == y.shape and np.array_equal(x.view(np.uint8), y.view(np.uint8)) for x, y in zip(a, b))


def test_oracle_merge_equals_reference_golden_vectors(oracle):
    g = np.load(GOLD)
    for i in range(int(g["n_triple_cases"])):
        got = pyoracle.bow_merge(g["t%d_wid" % i], g["t%d_w" % i], g["t%d_nid" % i])
        assert _same(got, [g["t%d_%s" % (i, k)] for k in OUT]), "triple case %d" % i


def test_oracle_transform_equals_reference_merge_of_its_descents(oracle):
    g = np.load(GOLD)
    voc = {k: g["voc_" + k] for k in ("node_desc", "child_off", "children", "word_id", "weight")}
    voc["L"] = int(g["voc_L"])
    wid, w, nid = pyoracle.bow_descend(voc, g["voc_desc"], int(g["voc_levelsup"]))
    assert np.array_equal(wid, g["voc_wid"]) and np.array_equal(w, g["voc_w"]) and np.array_equal(nid, g["voc_nid"])
    got = pyoracle.bow_transform(voc, g["voc_desc"], int(g["voc_levelsup"]))
    assert _same(got, [g["voc_" + k] for k in OUT])
    assert abs(got[1].sum() - 1.0) < 1e-12


@pytest.mark.skipif(pyoracle.build_ref() is None, reason="oracle/_ref not built and no reference tree to build it from")
def test_oracle_merge_equals_reference_library_live(oracle):
    """300 random triple sets through the reference's BowVector / FeatureVector and through the oracle: bit-identical."""
    rng = np.random.default_rng(5)
    for it in range(300):
        n = int(rng.integers(0, 2500)); nw = int(rng.integers(1, 400))
        wid = rng.integers(0, nw, n).astype(np.int32)
        w = np.exp(rng.uniform(-10, 5, n)); w[rng.random(n) < 0.1] = 0.0
        nid = rng.integers(0, max(nw // 4, 1), n).astype(np.uint32)
        assert _same(pyoracle.bow_merge(wid, w, nid), pyoracle.ref_bow_merge(wid, w, nid, 0)), it


@pytest.mark.skipif(pyoracle.build_ref() is None, reason="oracle/_ref not built and no reference tree to build it from")
def test_golden_vectors_are_what_the_reference_library_produces(oracle):
    g = np.load(GOLD)
    for i in range(int(g["n_triple_cases"])):
        assert _same(pyoracle.ref_bow_merge(g["t%d_wid" % i], g["t%d_w" % i], g["t%d_nid" % i], 0), [g["t%d_%s" % (i, k)] for k in OUT])
