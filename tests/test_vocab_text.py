"""ORBvoc text loader (reference lib/DBoW2/DBoW2/TemplatedVocabulary.h:1338 loadFromTextFile, :1428 saveToTextFile): a synthetic
tree is written in the reference's text format, parsed back by the library's host parser (no GPU) and compared array by array;
on the GPU the tree loaded from text must transform descriptors exactly like the tree created from the arrays."""
import numpy as np
import pytest

from ceres_mono_orb_slam2_amd import synth


def save_to_text_file(voc, path, scoring=0, weighting=0):
    """TemplatedVocabulary::saveToTextFile (:1428-1450): nodes 1.. in id order as "parent leaf d0 .. d31 weight"."""
    n = len(voc["word_id"])
    parent = np.zeros(n, np.int64)
    for i in range(n):
        parent[voc["children"][voc["child_off"][i]:voc["child_off"][i + 1]]] = i
    with open(path, "w") as f:
        f.write("%d %d  %d %d\n" % (voc["k"], voc["L"], scoring, weighting))
        for i in range(1, n):
            leaf = voc["child_off"][i + 1] == voc["child_off"][i]
            f.write("%d %d %s %r\n" % (parent[i], 1 if leaf else 0, " ".join(str(int(b)) for b in voc["node_desc"][i]) + " ", float(voc["weight"][i])))


@pytest.mark.parametrize("k,L,ragged", [(10, 3, 0.0), (6, 4, 0.3)])
def test_parse_text_round_trip(tmp_path, k, L, ragged):
    from ceres_mono_orb_slam2_amd import vocabulary
    voc = synth.make_vocabulary(3, k=k, L=L, ragged=ragged)
    path = tmp_path / "voc.txt"
    save_to_text_file(voc, path)
    got = vocabulary.parse_text(path)
    assert (got["k"], got["L"]) == (k, L)
    assert np.array_equal(got["node_desc"][1:], voc["node_desc"][1:])            # (the root's descriptor is not in the file)
    assert np.array_equal(got["child_off"], voc["child_off"]) and np.array_equal(got["children"], voc["children"])
    assert np.array_equal(got["word_id"], voc["word_id"]) and np.array_equal(got["weight"][1:], voc["weight"][1:])
    with open(path, "a") as f:
        f.write("\n\n")                                                              # trailing blank lines add no node
    assert len(vocabulary.parse_text(path)["word_id"]) == len(voc["word_id"])


def test_parse_text_rejects_garbage(tmp_path):
    from ceres_mono_orb_slam2_amd import vocabulary, _lib
    p = tmp_path / "bad.txt"; p.write_text("hello world\n")
    with pytest.raises(_lib.OrbHipError):
        vocabulary.parse_text(p)
    p.write_text("10 6 0 0\n7 0 " + "1 " * 32 + "0.5\n")                          # parent 7 does not exist yet
    with pytest.raises(_lib.OrbHipError):
        vocabulary.parse_text(p)
    p.write_text("10 6 0 0\n0 0 " + "1 " * 31 + "0.5\n")                          # 34 tokens: a byte is missing
    with pytest.raises(_lib.OrbHipError, match="35 tokens"):
        vocabulary.parse_text(p)
    p.write_text("10 6 0 0\n0 0 " + "1 " * 32 + "\n")                             # the weight is missing
    with pytest.raises(_lib.OrbHipError, match="35 tokens"):
        vocabulary.parse_text(p)
    p.write_text("10 6 0 0\n0 0 " + "300 " * 32 + "0.5\n")                        # not bytes
    with pytest.raises(_lib.OrbHipError):
        vocabulary.parse_text(p)
    p.write_text("10 6 0\n")                                                       # header too short
    with pytest.raises(_lib.OrbHipError):
        vocabulary.parse_text(p)
    p.write_text("2 1 0 0\n" + ("0 1 " + "1 " * 32 + "0.5\n") * 3)                # 4 nodes in a binary tree of depth 1
    with pytest.raises(_lib.OrbHipError, match="more than"):
        vocabulary.parse_text(p)


def test_load_text_rejects_other_scoring_before_touching_a_device(tmp_path):
    """orbv_transform / orbv_score_l1 implement TF-IDF weights with L1 scores (ORBvoc.txt: "10 6 0 0"); a vocabulary saved with
    another scoring / weighting type must not silently get those semantics."""
    from ceres_mono_orb_slam2_amd import vocabulary, _lib
    voc = synth.make_vocabulary(3, k=4, L=2)
    for sc, wt in ((1, 0), (0, 2)):
        path = tmp_path / ("voc_%d_%d.txt" % (sc, wt))
        save_to_text_file(voc, path, scoring=sc, weighting=wt)
        assert vocabulary.parse_text(path)["scoring"] == sc and vocabulary.parse_text(path)["weighting"] == wt
        with pytest.raises(_lib.OrbHipError, match="L1_NORM"):
            vocabulary.ORBVocabulary.loadFromTextFile(path)


@pytest.mark.gpu
def test_loaded_vocabulary_transforms_like_the_array_one(tmp_path):
    from ceres_mono_orb_slam2_amd import vocabulary
    voc = synth.make_vocabulary(5, k=8, L=4, ragged=0.2)
    path = tmp_path / "voc.txt"
    save_to_text_file(voc, path)
    A = vocabulary.ORBVocabulary(voc["node_desc"], voc["child_off"], voc["children"], voc["word_id"], voc["weight"], voc["L"])
    B = vocabulary.ORBVocabulary.loadFromTextFile(path)
    assert B.depth == A.depth == voc["L"]
    desc = np.random.default_rng(1).integers(0, 256, (1500, 32), dtype=np.uint8)
    a, b = A.transform(desc, 2), B.transform(desc, 2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(x, y)
