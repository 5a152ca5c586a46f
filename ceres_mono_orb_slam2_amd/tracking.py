"""Tracking::TrackWithMotionModel's data-parallel core as ONE device-resident call (reference src/Tracking.cc:616-646):
Frame construction (ORBextractor::operator(), grid), ORBmatcher::SearchByProjection(current_frame_, last_frame_, th) and
CeresOptimizer::PoseOptimization - include/orbslam_hip.h::orbt_track_with_motion_model, csrc/orb_track.hip."""
import ctypes as C
import threading

import numpy as np

from . import _lib
from .extractor import KP_DTYPE


class TrackResult(C.Structure):                # orbt_result
    _fields_ = [("n_keypoints", C.c_int32), ("nmatches", C.c_int32), ("n_correspondences", C.c_int32), ("n_inliers", C.c_int32),
                ("greedy_rounds", C.c_int32), ("reserved", C.c_int32), ("pose7", C.c_double * 7)]


def _c(a, dt):
    """C-contiguous array of dtype dt without a copy (or a call into numpy) when it already is one"""
    if type(a) is np.ndarray and a.dtype == dt and a.flags.c_contiguous:
        return a
    return np.ascontiguousarray(a, dt)


def _addr(a):
    return a.__array_interface__["data"][0]


def track_with_motion_model(extractor, image, K4, bounds, Tcw_pred, last_Xw, last_desc, last_octave, last_angle, last_valid, th=15.0,
                            check_ori=True, copy=True):
    """extractor: ORBextractor; image (H, W) uint8; Tcw_pred (3 or 4, 4); the last frame's per-feature arrays (see the header).
    Returns dict(kps, desc, match, owner, outlier, pose7, nmatches, n_inliers, n_correspondences, greedy_rounds).  copy=True (the
    default) returns private arrays.  copy=False is for the latency-critical caller: the arrays are then READ-ONLY VIEWS of two
    alternating buffer sets kept with the extractor - frame N's arrays are overwritten by frame N + 2 - so anything kept longer
    (keyframe bookkeeping) must be copied by the caller.  Calls on one extractor are serialised by a lock kept with it."""
    L = _lib.load()
    img = _c(image, np.uint8)
    h, w = img.shape
    K4 = _c(K4, np.float32); bounds = _c(bounds, np.float32)
    T = np.ascontiguousarray(np.asarray(Tcw_pred, np.float64).reshape(-1)[:12])
    X = _c(last_Xw, np.float64).reshape(-1, 3); n = len(X)
    D = _c(last_desc, np.uint8).reshape(-1, 32); O = _c(last_octave, np.int32)
    A = _c(last_angle, np.float32); V = _c(last_valid, np.uint8)
    assert len(D) == n and len(O) == n and len(A) == n and len(V) == n
    # The output buffers (and their addresses) live with the extractor, two sets used alternately: a per-frame call must not spend
    # its time in allocations, page faults and copies (190 us of Python per call at first, ~25 now).
    with _extractor_lock(extractor):
        return _track_locked(L, extractor, img, w, h, K4, bounds, T, X, D, O, A, V, n, th, check_ori, copy)


def _extractor_lock(extractor):
    """One lock per extractor: every entry point that runs an extraction on it (pyramid / blur buffers, side stream and events of
    the context are per-extractor state) takes it, so calls on one extractor from several Python threads are serialised."""
    lock = extractor.__dict__.get("_track_lock")
    if lock is None:
        lock = extractor.__dict__.setdefault("_track_lock", threading.Lock())
    return lock


def _track_locked(L, extractor, img, w, h, K4, bounds, T, X, D, O, A, V, n, th, check_ori, copy):
    cap = extractor.max_keypoints
    S = getattr(extractor, "_track_bufs", None)
    if S is None or S["cap"] != cap or S["nq"] < n:
        nq = max(n, 1, S["nq"] if S else 0)
        def mk():
            b = dict(kps=np.zeros(cap, KP_DTYPE), desc=np.zeros((cap, 32), np.uint8), match=np.full(nq, -1, np.int32),
                     owner=np.full(cap, -1, np.int32), outl=np.zeros(cap, np.uint8), res=TrackResult())
            b["p"] = tuple(_addr(b[k]) for k in ("kps", "desc", "match", "owner", "outl"))
            b["pres"] = C.byref(b["res"])
            return b
        S = dict(cap=cap, nq=nq, sets=(mk(), mk()), turn=0)
        extractor._track_bufs = S
    S["turn"] ^= 1
    B = S["sets"][S["turn"]]
    pk, pd, pm, po, pl = B["p"]
    res = B["res"]
    _lib.check(L.orbt_track_with_motion_model(extractor._h, _addr(img), w, h, img.strides[0], _addr(K4), _addr(bounds), _addr(T), _addr(X),
                                              _addr(D), _addr(O), _addr(A), _addr(V), n, float(th), int(bool(check_ori)), pk, pd, cap, pm, po, pl,
                                              B["pres"]), "orbt_track_with_motion_model")
    k = res.n_keypoints
    out = dict(kps=B["kps"][:k], desc=B["desc"][:k], match=B["match"][:n], owner=B["owner"][:k], outlier=B["outl"][:k].view(np.bool_),
               pose7=np.array(res.pose7[:], np.float64), nmatches=res.nmatches, n_inliers=res.n_inliers,
               n_correspondences=res.n_correspondences, greedy_rounds=res.greedy_rounds)
    for key in ("kps", "desc", "match", "owner", "outlier"):
        if copy:
            out[key] = out[key].copy()
        else:
            v = out[key].view(); v.flags.writeable = False; out[key] = v
    return out


def track_local_map(extractor, K4, bounds, Tcw, log_scale_factor, mp_Xw, mp_normal, mp_min_dist, mp_max_dist, mp_desc, mp_state, slot_Xw, slot_state,
                    th=1.0, nnratio=0.8):
    """Tracking::TrackLocalMap's data-parallel core on the frame the last track_with_motion_model call of THIS thread left on the
    device (include/orbslam_hip.h::orbt_track_local_map; reference src/Tracking.cc:673-750, :793-842, src/ORBmatcher.cc:42-119).
    Returns dict(in_view, match, owner, outlier, pose7, nmatches, n_inliers, n_correspondences, n_in_view, greedy_rounds)."""
    L = _lib.load()
    K4 = _c(K4, np.float32); bounds = _c(bounds, np.float32)
    T = np.ascontiguousarray(np.asarray(Tcw, np.float64).reshape(-1)[:12])
    X = _c(mp_Xw, np.float64).reshape(-1, 3); n = len(X)
    N = _c(mp_normal, np.float64).reshape(-1, 3); mn = _c(mp_min_dist, np.float32); mx = _c(mp_max_dist, np.float32)
    D = _c(mp_desc, np.uint8).reshape(-1, 32); S = _c(mp_state, np.uint8)
    SX = _c(slot_Xw, np.float64).reshape(-1, 3); SS = _c(slot_state, np.uint8); nk = len(SS)
    assert len(N) == n and len(mn) == n and len(mx) == n and len(D) == n and len(S) == n and len(SX) == nk
    in_view = np.zeros(max(n, 1), np.uint8); match = np.full(max(n, 1), -1, np.int32)
    owner = np.full(max(nk, 1), -1, np.int32); outl = np.zeros(max(nk, 1), np.uint8)
    res = TrackResult()
    _lib.check(L.orbt_track_local_map(extractor._h, _addr(K4), _addr(bounds), _addr(T), float(log_scale_factor), _addr(X), _addr(N), _addr(mn), _addr(mx), _addr(D),
                                      _addr(S), n, _addr(SX), _addr(SS), nk, float(th), float(nnratio), _addr(in_view), _addr(match), _addr(owner), _addr(outl),
                                      C.byref(res)), "orbt_track_local_map")
    return dict(in_view=in_view[:n].view(np.bool_), match=match[:n], owner=owner[:nk], outlier=outl[:nk].view(np.bool_), pose7=np.array(res.pose7[:], np.float64),
                nmatches=res.nmatches, n_inliers=res.n_inliers, n_correspondences=res.n_correspondences, n_in_view=res.reserved, greedy_rounds=res.greedy_rounds)


def track_reference_keyframe(extractor, vocabulary, image, K4, bounds, Tcw_last, kf_desc, kf_valid, kf_angle, kf_Xw, kf_fv, nnratio=0.7, check_ori=True):
    """Tracking::TrackReferenceKeyFrame's data-parallel core (include/orbslam_hip.h::orbt_track_reference_keyframe; reference
    src/Tracking.cc:566-615).  image None: the frame of the last orbt_* call of this thread.  kf_fv = (node ids, offsets, indices).
    Returns dict(kps, desc (None without image), bow=(words, values), fv=(nodes, offsets, indices), match, owner, outlier, pose7, ...)."""
    L = _lib.load()
    K4 = _c(K4, np.float32); bounds = _c(bounds, np.float32)
    T = np.ascontiguousarray(np.asarray(Tcw_last, np.float64).reshape(-1)[:12])
    D = _c(kf_desc, np.uint8).reshape(-1, 32); n = len(D)
    V = _c(kf_valid, np.uint8); A = _c(kf_angle, np.float32); X = _c(kf_Xw, np.float64).reshape(-1, 3)
    fn, fo, fi = [_c(x, np.uint32) for x in kf_fv]
    assert len(V) == n and len(A) == n and len(X) == n and len(fo) == len(fn) + 1
    cap = extractor.max_keypoints
    kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
    bw = np.zeros(cap, np.uint32); bv = np.zeros(cap, np.float64); nw = C.c_int(0)
    on = np.zeros(cap, np.uint32); oo = np.zeros(cap + 2, np.uint32); oi = np.zeros(cap, np.uint32); nf = C.c_int(0)
    match = np.full(max(n, 1), -1, np.int32); owner = np.full(cap, -1, np.int32); outl = np.zeros(cap, np.uint8)
    res = TrackResult()
    if image is not None:
        img = _c(image, np.uint8); h, w = img.shape; ip, st = _addr(img), img.strides[0]
    else:
        img, h, w, ip, st = None, 0, 0, None, 0
    import contextlib
    with (_extractor_lock(extractor) if img is not None else contextlib.nullcontext()):      # (with an image the call runs an extraction on this extractor)
        _lib.check(L.orbt_track_reference_keyframe(extractor._h, vocabulary._h, ip, w, h, st, _addr(K4), _addr(bounds), _addr(T), _addr(D), _addr(V), _addr(A), _addr(X), n,
                                                   _addr(fn), _addr(fo), _addr(fi), len(fn), float(nnratio), int(bool(check_ori)), _addr(kps), _addr(desc), cap,
                                                   _addr(bw), _addr(bv), C.byref(nw), _addr(on), _addr(oo), _addr(oi), C.byref(nf), _addr(match), _addr(owner), _addr(outl),
                                                   C.byref(res)), "orbt_track_reference_keyframe")
    k = res.n_keypoints
    return dict(kps=kps[:k] if img is not None else None, desc=desc[:k] if img is not None else None, bow=(bw[:nw.value], bv[:nw.value]),
                fv=(on[:nf.value], oo[:nf.value + 1], oi[:oo[nf.value]]), match=match[:n], owner=owner[:k], outlier=outl[:k].view(np.bool_),
                pose7=np.array(res.pose7[:], np.float64), nmatches=res.nmatches, n_inliers=res.n_inliers, n_correspondences=res.n_correspondences, n_keypoints=k)


def last_call_ms():
    """orbt_last_call_ms: wall time of this thread's most recent track_* call inside the library (a Python caller adds its own
    interpreter-lock waits around the call when other threads are busy)."""
    return float(_lib.load().orbt_last_call_ms())


class _RelocKF(C.Structure):         # orbt_reloc_keyframe
    _fields_ = [("desc", C.c_void_p), ("valid", C.c_void_p), ("angle", C.c_void_p), ("n", C.c_int),
                ("fv_node", C.c_void_p), ("fv_off", C.c_void_p), ("fv_idx", C.c_void_p), ("fv_n", C.c_int)]


def relocalization_search_by_bow(extractor, vocabulary, image, K4, bounds, candidates, nnratio=0.75, check_ori=True):
    """Tracking::Relocalization, first stage (include/orbslam_hip.h::orbt_relocalization_search_by_bow; reference src/Tracking.cc:979-1029):
    ComputeBoW of the frame + SearchByBoW(keyframe, frame) for every candidate keyframe in one call.  candidates: dicts(desc[n,32],
    valid[n], angle[n], fv=(nodes, offsets, indices)).  image None: the frame of the last orbt_* call of this thread.
    Returns dict(kps, desc (None without image), bow, fv, owner[n_cand, n_keypoints], nmatches[n_cand])."""
    import contextlib
    L = _lib.load()
    K4 = _c(K4, np.float32); bounds = _c(bounds, np.float32)
    keep = []
    nc = len(candidates)
    cs = (_RelocKF * max(nc, 1))()
    for i, q in enumerate(candidates):
        D = _c(q["desc"], np.uint8).reshape(-1, 32); V = _c(q["valid"], np.uint8); A = _c(q["angle"], np.float32)
        fn, fo, fi = [_c(x, np.uint32) for x in q["fv"]]
        assert len(V) == len(D) and len(A) == len(D) and len(fo) == len(fn) + 1
        keep += [D, V, A, fn, fo, fi]
        cs[i].desc, cs[i].valid, cs[i].angle, cs[i].n = D.ctypes.data, V.ctypes.data, A.ctypes.data, len(D)
        cs[i].fv_node, cs[i].fv_off, cs[i].fv_idx, cs[i].fv_n = fn.ctypes.data, fo.ctypes.data, fi.ctypes.data, len(fn)
    cap = extractor.max_keypoints
    kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
    bw = np.zeros(cap, np.uint32); bv = np.zeros(cap, np.float64); nw = C.c_int(0)
    on = np.zeros(cap, np.uint32); oo = np.zeros(cap + 2, np.uint32); oi = np.zeros(cap, np.uint32); nf = C.c_int(0)
    owner = np.full((max(nc, 1), cap), -1, np.int32); nm = np.zeros(max(nc, 1), np.int32); nk = C.c_int(0)
    if image is not None:
        img = _c(image, np.uint8); h, w = img.shape; ip, st = _addr(img), img.strides[0]
    else:
        img, h, w, ip, st = None, 0, 0, None, 0
    L.orbt_relocalization_search_by_bow.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int,
                                                    C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int),
                                                    C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    with (_extractor_lock(extractor) if img is not None else contextlib.nullcontext()):
        _lib.check(L.orbt_relocalization_search_by_bow(extractor._h, vocabulary._h, ip, w, h, st, _addr(K4), _addr(bounds), C.cast(cs, C.c_void_p), nc, float(nnratio),
                                                       int(bool(check_ori)), _addr(kps), _addr(desc), cap, _addr(bw), _addr(bv), C.byref(nw), _addr(on), _addr(oo), _addr(oi),
                                                       C.byref(nf), _addr(owner), _addr(nm), C.byref(nk)), "orbt_relocalization_search_by_bow")
    k = nk.value
    return dict(kps=kps[:k] if img is not None else None, desc=desc[:k] if img is not None else None, bow=(bw[:nw.value], bv[:nw.value]),
                fv=(on[:nf.value], oo[:nf.value + 1], oi[:oo[nf.value]]), owner=owner[:nc, :k], nmatches=nm[:nc], n_keypoints=k)
