"""Host mirror of ORBVocabulary::transform as Frame::ComputeBoW uses it (reference src/Frame.cc:322-327,
lib/DBoW2/DBoW2/TemplatedVocabulary.h:1124-1260) over the HIP C ABI.  The vocabulary is held in flattened form
(see include/orbslam_hip.h::orbv_create); loading ORBvoc.txt into that form is host-side control plane."""
import ctypes as C

import numpy as np

from . import _lib


class ORBVocabulary:
    def __init__(self, node_desc, child_off, children, word_id, weight, L, device=0):
        self._L = _lib.load()
        nd = np.ascontiguousarray(node_desc, np.uint8).reshape(-1, 32); co = np.ascontiguousarray(child_off, np.uint32)
        ch = np.ascontiguousarray(children, np.uint32); wi = np.ascontiguousarray(word_id, np.int32); wt = np.ascontiguousarray(weight, np.float64)
        self.depth = int(L)
        self._h = C.c_void_p()
        _lib.check(self._L.orbv_create(_lib.ptr(nd), _lib.ptr(co), _lib.ptr(ch), _lib.ptr(wi), _lib.ptr(wt), len(wi), int(L), int(device),
                                       C.byref(self._h)), "orbv_create")

    @classmethod
    def loadFromTextFile(cls, path, device=0):
        """ORBVocabulary::loadFromTextFile (lib/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1423): ORBvoc.txt -> device-resident tree."""
        self = cls.__new__(cls)
        self._L = _lib.load()
        self._h = C.c_void_p()
        _lib.check(self._L.orbv_load_text(str(path).encode(), int(device), C.byref(self._h)), "orbv_load_text")
        with open(path, "rb") as f:                     # header line "k L scoring weighting" (orbv_load_text validated it)
            self.depth = int(f.readline().split()[1])
        return self

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orbv_destroy(self._h); self._h = None

    def transform(self, desc, levelsup=4):
        """-> (bow_word, bow_value, (fv_node, fv_off, fv_idx)): BowVector (ascending word ids, L1-normalised tf-idf) and the
        FeatureVector in the CSR layout ORBmatcher.SearchByBoW takes."""
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); n = len(d)
        bw = np.zeros(max(n, 1), np.uint32); bv = np.zeros(max(n, 1), np.float64); nw = C.c_int(0)
        fn = np.zeros(max(n, 1), np.uint32); fo = np.zeros(n + 2, np.uint32); fi = np.zeros(max(n, 1), np.uint32); nf = C.c_int(0)
        _lib.check(self._L.orbv_transform(self._h, _lib.ptr(d), n, int(levelsup), _lib.ptr(bw), _lib.ptr(bv), C.byref(nw), _lib.ptr(fn),
                                          _lib.ptr(fo), _lib.ptr(fi), C.byref(nf)), "orbv_transform")
        return bw[:nw.value], bv[:nw.value], (fn[:nf.value], fo[:nf.value + 1], fi[:fo[nf.value]])

    def descend_device(self, d_desc, levelsup=4, stream=None):
        """Tree descent for device-resident descriptors (torch uint8 [n, 32]); returns (word int32[n], weight f64[n], node uint32-as-int32[n])."""
        import torch
        n = d_desc.shape[0]; dev = d_desc.device
        word = torch.empty((max(n, 1),), dtype=torch.int32, device=dev); wt = torch.empty((max(n, 1),), dtype=torch.float64, device=dev)
        node = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        _lib.check(self._L.orbv_descend_device(self._h, _lib.ptr(d_desc), n, int(levelsup), _lib.ptr(word), _lib.ptr(wt), _lib.ptr(node),
                                               C.c_void_p(st)), "orbv_descend_device")
        return word[:n], wt[:n], node[:n]

    def score(self, bow1, bow2):
        w1 = np.ascontiguousarray(bow1[0], np.uint32); v1 = np.ascontiguousarray(bow1[1], np.float64)
        w2 = np.ascontiguousarray(bow2[0], np.uint32); v2 = np.ascontiguousarray(bow2[1], np.float64)
        return float(self._L.orbv_score_l1(_lib.ptr(w1), _lib.ptr(v1), len(w1), _lib.ptr(w2), _lib.ptr(v2), len(w2)))


def parse_text(path):
    """Host half of loadFromTextFile: the flattened arrays orbv_create takes.  No device needed."""
    L = _lib.load()
    ints = [C.c_int32() for _ in range(6)]
    ptrs = [C.c_void_p() for _ in range(5)]
    _lib.check(L.orbv_parse_text(str(path).encode(), *[C.byref(x) for x in ints], *[C.byref(x) for x in ptrs]), "orbv_parse_text")
    k, depth, scoring, weighting, n, nc = [x.value for x in ints]
    def take(ptr, dt, count):
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(max(count, 1),))[:count].copy()
        L.orbv_free_parsed(ptr)
        return a
    return dict(k=k, L=depth, scoring=scoring, weighting=weighting, node_desc=take(ptrs[0], np.uint8, 32 * n).reshape(n, 32),
                child_off=take(ptrs[1], np.uint32, n + 1), children=take(ptrs[2], np.uint32, nc), word_id=take(ptrs[3], np.int32, n),
                weight=take(ptrs[4], np.float64, n))
