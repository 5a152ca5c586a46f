"""Host mirror of the flattened ORB_SLAM2::ORBmatcher entry points (reference include/ORBmatcher.h,
src/ORBmatcher.cc) over the HIP C ABI."""
import ctypes as C

import numpy as np

from . import _lib


class ORBmatcher:
    TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30          # src/ORBmatcher.cc:35-37

    def __init__(self, nnratio=0.6, checkOri=True):       # include/ORBmatcher.h:38
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)
        self._L = _lib.load()

    @staticmethod
    def DescriptorDistance(a, b):
        """src/ORBmatcher.cc:1422-1437"""
        a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
        assert a.size == 32 and b.size == 32
        return _lib.load().orbm_descriptor_distance(_lib.ptr(a), _lib.ptr(b))

    def hamming_best2(self, q, t, cand_offsets=None, cand_idx=None):
        """best/second-best distance of each query over its candidate list (host arrays)."""
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32); t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        nq, nt = len(q), len(t)
        bi = np.full(nq, -1, np.int32); bd = np.full(nq, 256, np.int32); sd = np.full(nq, 256, np.int32)
        co = None if cand_offsets is None else np.ascontiguousarray(cand_offsets, np.uint32)
        ci = None if cand_idx is None else np.ascontiguousarray(cand_idx, np.uint32)
        if ci is not None and ci.size == 0:
            ci = np.zeros(1, np.uint32)
        _lib.check(self._L.orbm_hamming_best2(_lib.ptr(q), nq, _lib.ptr(t), nt, _lib.ptr(co), _lib.ptr(ci),
                                              _lib.ptr(bi), _lib.ptr(bd), _lib.ptr(sd)), "orbm_hamming_best2")
        return bi, bd, sd

    def hamming_best2_device(self, d_q, d_t, d_off=None, d_idx=None, stream=None):
        import torch
        nq, nt = d_q.shape[0], d_t.shape[0]
        out = torch.empty((3, max(nq, 1)), dtype=torch.int32, device=d_q.device)
        st = torch.cuda.current_stream(d_q.device).cuda_stream if stream is None else stream
        _lib.check(self._L.orbm_hamming_best2_device(_lib.ptr(d_q), nq, _lib.ptr(d_t), nt, _lib.ptr(d_off),
                                                     _lib.ptr(d_idx), _lib.ptr(out[0]), _lib.ptr(out[1]),
                                                     _lib.ptr(out[2]), C.c_void_p(st)), "orbm_hamming_best2_device")
        return out[0, :nq], out[1, :nq], out[2, :nq]

    def match_frames_batch(self, kps, desc, counts, pair_a, pair_b, th=None, stream=None, out=None):
        """Brute-force frame-to-frame matching of extract_batch outputs (device tensors):
        ratio test (mfNNratio), threshold (default TH_LOW) and rotation consistency (mbCheckOrientation)."""
        import torch
        cap = kps.shape[1]
        npairs = pair_a.shape[0]
        if out is None:
            m = torch.empty((npairs, cap), dtype=torch.int32, device=kps.device)
            nm = torch.empty((npairs,), dtype=torch.int32, device=kps.device)
        else:
            m, nm = out
        st = torch.cuda.current_stream(kps.device).cuda_stream if stream is None else stream
        _lib.check(self._L.orbm_match_frames_batch_device(_lib.ptr(kps), _lib.ptr(desc), _lib.ptr(counts), cap,
                                                          _lib.ptr(pair_a), _lib.ptr(pair_b), npairs, self.mfNNratio,
                                                          self.TH_LOW if th is None else int(th),
                                                          int(self.mbCheckOrientation), _lib.ptr(m), _lib.ptr(nm),
                                                          C.c_void_p(st)), "orbm_match_frames_batch_device")
        return m, nm

    def SearchForInitialization(self, kps1, desc1, kps2, desc2, bounds2, vbPrevMatched, windowSize=10):
        """src/ORBmatcher.cc:363-468 on flattened frames.  kps = (n,4) float32 [x,y,octave,angle];
        vbPrevMatched (n1,2) float32 is updated in place.  Returns (nmatches, vnMatches12)."""
        kps1 = np.ascontiguousarray(kps1, np.float32); kps2 = np.ascontiguousarray(kps2, np.float32)
        desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
        b = np.ascontiguousarray(bounds2, np.float32)
        assert vbPrevMatched.dtype == np.float32 and vbPrevMatched.flags.c_contiguous
        m = np.full(len(kps1), -1, np.int32)
        n = C.c_int(0)
        _lib.check(self._L.orbm_search_for_initialization(_lib.ptr(kps1), _lib.ptr(desc1), len(kps1), _lib.ptr(kps2),
                                                          _lib.ptr(desc2), len(kps2), _lib.ptr(b),
                                                          _lib.ptr(vbPrevMatched), int(windowSize), self.mfNNratio,
                                                          int(self.mbCheckOrientation), _lib.ptr(m), C.byref(n)),
                   "orbm_search_for_initialization")
        return n.value, m

    # ---- guided searches on flattened data (see include/orbslam_hip.h for the reference entry points covered) ----
    def search_by_projection(self, kps4, desc, bounds, q_uv, q_radius, q_desc, q_min_level=None, q_max_level=None,
                             q_pred_level=None, q_valid=None, q_angle=None, inv_level_sigma2=None, chi2_gate=0.0,
                             taken=None, mode_best2=False, th=None):
        """Returns (nmatches, q_match[nq], q_best_dist[nq], taken_out or None).  ratio = mfNNratio, rotation check =
        mbCheckOrientation (needs q_angle); th defaults to TH_HIGH."""
        f32, u8, i32 = np.float32, np.uint8, np.int32
        opt = lambda a, dt: None if a is None else np.ascontiguousarray(a, dt)
        kps4 = np.ascontiguousarray(kps4, f32); desc = np.ascontiguousarray(desc, u8); b = np.ascontiguousarray(bounds, f32)
        q_uv = np.ascontiguousarray(q_uv, f32); q_radius = np.ascontiguousarray(q_radius, f32); q_desc = np.ascontiguousarray(q_desc, u8)
        nq = len(q_radius)
        mn, mx, pl = opt(q_min_level, i32), opt(q_max_level, i32), opt(q_pred_level, i32)
        qv, qa, isg = opt(q_valid, u8), opt(q_angle, f32), opt(inv_level_sigma2, f32)
        tk = None if taken is None else np.ascontiguousarray(taken, u8).copy()
        m = np.full(max(nq, 1), -1, i32); bd = np.full(max(nq, 1), 256, i32); n = C.c_int(0)
        check_ori = int(self.mbCheckOrientation and qa is not None)
        _lib.check(self._L.orbm_search_by_projection(_lib.ptr(kps4), _lib.ptr(desc), len(kps4), _lib.ptr(b), _lib.ptr(q_uv),
                                                     _lib.ptr(q_radius), _lib.ptr(mn), _lib.ptr(mx), _lib.ptr(pl), _lib.ptr(q_desc),
                                                     _lib.ptr(qv), _lib.ptr(qa), nq, _lib.ptr(isg), float(chi2_gate), _lib.ptr(tk),
                                                     int(mode_best2), self.mfNNratio, self.TH_HIGH if th is None else int(th),
                                                     check_ori, _lib.ptr(m), _lib.ptr(bd), C.byref(n)), "orbm_search_by_projection")
        return n.value, m[:nq], bd[:nq], tk

    def SearchByBoW(self, desc1, valid1, angle1, desc2, valid2, angle2, fv1, fv2, strict=False, th=None):
        """src/ORBmatcher.cc:151-256 (strict=False) / :470-580 (strict=True) on flattened data.
        fv = (ascending node ids, CSR offsets, keypoint indices).  Returns (nmatches, match12[n1])."""
        u8, f32, u32 = np.uint8, np.float32, np.uint32
        opt = lambda a, dt: None if a is None else np.ascontiguousarray(a, dt)
        d1 = np.ascontiguousarray(desc1, u8); d2 = np.ascontiguousarray(desc2, u8)
        v1, v2, a1, a2 = opt(valid1, u8), opt(valid2, u8), opt(angle1, f32), opt(angle2, f32)
        f1 = [np.ascontiguousarray(x, u32) for x in fv1]; f2 = [np.ascontiguousarray(x, u32) for x in fv2]
        m = np.full(max(len(d1), 1), -1, np.int32); n = C.c_int(0)
        _lib.check(self._L.orbm_search_by_bow(_lib.ptr(d1), len(d1), _lib.ptr(v1), _lib.ptr(a1), _lib.ptr(d2), len(d2), _lib.ptr(v2),
                                              _lib.ptr(a2), _lib.ptr(f1[0]), _lib.ptr(f1[1]), _lib.ptr(f1[2]), len(f1[0]),
                                              _lib.ptr(f2[0]), _lib.ptr(f2[1]), _lib.ptr(f2[2]), len(f2[0]), self.mfNNratio,
                                              self.TH_LOW if th is None else int(th), int(strict), int(self.mbCheckOrientation),
                                              _lib.ptr(m), C.byref(n)), "orbm_search_by_bow")
        return n.value, m[:len(d1)]

    def SearchForTriangulation(self, kps1, desc1, unmapped1, kps2, desc2, unmapped2, fv1, fv2, F12, epipole, scale_factors,
                               level_sigma2):
        """src/ORBmatcher.cc:582-722 (mono) on flattened data.  Returns (nmatches, match12[n1])."""
        f32, u8, u32 = np.float32, np.uint8, np.uint32
        opt = lambda a, dt: None if a is None else np.ascontiguousarray(a, dt)
        k1 = np.ascontiguousarray(kps1, f32); k2 = np.ascontiguousarray(kps2, f32)
        d1 = np.ascontiguousarray(desc1, u8); d2 = np.ascontiguousarray(desc2, u8)
        u1, u2 = opt(unmapped1, u8), opt(unmapped2, u8)
        f1 = [np.ascontiguousarray(x, u32) for x in fv1]; f2 = [np.ascontiguousarray(x, u32) for x in fv2]
        F = np.ascontiguousarray(F12, np.float64).reshape(9); sf = np.ascontiguousarray(scale_factors, f32)
        ls = np.ascontiguousarray(level_sigma2, f32)
        m = np.full(max(len(k1), 1), -1, np.int32); n = C.c_int(0)
        _lib.check(self._L.orbm_search_for_triangulation(_lib.ptr(k1), _lib.ptr(d1), _lib.ptr(u1), len(k1), _lib.ptr(k2), _lib.ptr(d2),
                                                         _lib.ptr(u2), len(k2), _lib.ptr(f1[0]), _lib.ptr(f1[1]), _lib.ptr(f1[2]),
                                                         len(f1[0]), _lib.ptr(f2[0]), _lib.ptr(f2[1]), _lib.ptr(f2[2]), len(f2[0]),
                                                         _lib.ptr(F), float(epipole[0]), float(epipole[1]), _lib.ptr(sf), _lib.ptr(ls),
                                                         int(self.mbCheckOrientation), _lib.ptr(m), C.byref(n)),
                   "orbm_search_for_triangulation")
        return n.value, m[:len(k1)]

    def SearchBySim3(self, kps1, desc1, kps2, desc2, bounds, q12_uv, q12_radius, q12_pred, q12_valid, q21_uv, q21_radius,
                     q21_pred, q21_valid, q12_desc=None, q21_desc=None, bounds2=None):
        """src/ORBmatcher.cc:956-1159 from the two window searches on (see orbm_search_by_sim3).  Returns (nFound, match12[n1]).
        bounds = image bounds of keyframe 1, bounds2 = of keyframe 2 (default: the same camera)."""
        f32, u8, i32 = np.float32, np.uint8, np.int32
        c = np.ascontiguousarray
        k1, k2, d1, d2, b = c(kps1, f32), c(kps2, f32), c(desc1, u8), c(desc2, u8), c(bounds, f32)
        b2 = b if bounds2 is None else c(bounds2, f32)
        od = lambda d: None if d is None else c(d, u8)
        a = [c(q12_uv, f32), c(q12_radius, f32), c(q12_pred, i32), c(q12_valid, u8), od(q12_desc), c(q21_uv, f32), c(q21_radius, f32),
             c(q21_pred, i32), c(q21_valid, u8), od(q21_desc)]
        m = np.full(max(len(k1), 1), -1, i32); n = C.c_int(0)
        _lib.check(self._L.orbm_search_by_sim3(_lib.ptr(k1), _lib.ptr(d1), len(k1), _lib.ptr(k2), _lib.ptr(d2), len(k2), _lib.ptr(b), _lib.ptr(b2),
                                               *[_lib.ptr(x) for x in a], _lib.ptr(m), C.byref(n)), "orbm_search_by_sim3")
        return n.value, m[:len(k1)]
