#!/usr/bin/env python3
"""Generate tests/golden/bow_merge_ref.npz with the REFERENCE's own code (run from the repo root, in the container that
holds /root/reference).

The expected outputs come from oracle/_ref/libdbow2_ref.so = the reference's lib/DBoW2/DBoW2/BowVector.cpp and
FeatureVector.cpp compiled where they lie (`make -C oracle ref`): BowVector::addWeight / normalize(L1) and
FeatureVector::addFeature applied to per-feature (word, idf weight, node) triples - the merge half of Frame::ComputeBoW
(src/Frame.cc:322-327 -> TemplatedVocabulary::transform, lib/DBoW2/DBoW2/TemplatedVocabulary.h:1124-1200).  The fixture is
data only (inputs + expected outputs); it pins the oracle's restatement (tests/test_oracle_bow_ref.py) and the product's host
merge (tests/test_gpu_bow.py) on machines without the reference tree.

Two kinds of cases:
  * `t<i>_*`: raw triples with heavy word collisions, stopped words (weight 0), weights of very different magnitude (so the
    accumulation ORDER shows in the last bits) and unsorted node ids;
  * `voc_*`: a synthetic k = 6, L = 4 vocabulary + 400 descriptors; the triples are the oracle's descents (that half stays
    "parity unpinned": TemplatedVocabulary.h needs OpenCV), merged by the reference."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po            # noqa: E402
from ceres_mono_orb_slam2_amd import synth   # noqa: E402

assert po.build_ref() and po.ref_lib() is not None, "oracle/_ref needs /root/reference"
out = {}
rng = np.random.default_rng(20260929)
cases = [(0, 1), (1, 1), (17, 3), (400, 40), (2000, 150), (2000, 1999), (3000, 7)]
for i, (n, nwords) in enumerate(cases):
    wid = rng.integers(0, max(nwords, 1), n).astype(np.int32) * 7 + 3
    w = np.exp(rng.uniform(-12, 6, n))                      # 8 decades: sums depend on the order of accumulation
    w[rng.random(n) < 0.12] = 0.0                           # stopped words
    nid = rng.integers(0, max(nwords // 3, 1), n).astype(np.uint32) * 11 + 1
    bw, bv, fn, fo, fi = po.ref_bow_merge(wid, w, nid, 0)
    for k, v in (("wid", wid), ("w", w), ("nid", nid), ("bow_word", bw), ("bow_value", bv), ("fv_node", fn), ("fv_off", fo), ("fv_idx", fi)):
        out["t%d_%s" % (i, k)] = v
out["n_triple_cases"] = np.int32(len(cases))

voc = synth.make_vocabulary(11, k=6, L=4, ragged=0.2)
leaves = np.nonzero(voc["word_id"] >= 0)[0]
d = voc["node_desc"][rng.choice(leaves, 400)].copy()
bits = np.unpackbits(d, axis=1)
for r in range(len(bits)):
    bits[r, rng.choice(256, 20, replace=False)] ^= 1
d = np.packbits(bits, axis=1)
levelsup = 2
wid, w, nid = po.bow_descend(voc, d, levelsup)
bw, bv, fn, fo, fi = po.ref_bow_merge(wid, w, nid, 0)
for k in ("node_desc", "child_off", "children", "word_id", "weight"):
    out["voc_" + k] = voc[k]
out["voc_L"] = np.int32(voc["L"]); out["voc_levelsup"] = np.int32(levelsup); out["voc_desc"] = d
for k, v in (("wid", wid), ("w", w), ("nid", nid), ("bow_word", bw), ("bow_value", bv), ("fv_node", fn), ("fv_off", fo), ("fv_idx", fi)):
    out["voc_" + k] = v
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bow_merge_ref.npz")
np.savez_compressed(path, **out)
print("bow_merge_ref.npz:", os.path.getsize(path), "bytes;", len(bw), "words,", len(fn), "nodes")
