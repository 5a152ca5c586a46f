"""Tracking::TrackWithMotionModel's data-parallel core as ONE device-resident call (reference src/Tracking.cc:616-646):
Frame construction (ORBextractor::operator(), grid), ORBmatcher::SearchByProjection(current_frame_, last_frame_, th) and
CeresOptimizer::PoseOptimization - include/orbslam_hip.h::orbt_track_with_motion_model, csrc/orb_track.hip."""
import ctypes as C

import numpy as np

from . import _lib
from .extractor import KP_DTYPE


class TrackResult(C.Structure):                # orbt_result
    _fields_ = [("n_keypoints", C.c_int32), ("nmatches", C.c_int32), ("n_correspondences", C.c_int32), ("n_inliers", C.c_int32),
                ("greedy_rounds", C.c_int32), ("reserved", C.c_int32), ("pose7", C.c_double * 7)]


def track_with_motion_model(extractor, image, K4, bounds, Tcw_pred, last_Xw, last_desc, last_octave, last_angle, last_valid, th=15.0,
                            check_ori=True):
    """extractor: ORBextractor; image (H, W) uint8; Tcw_pred (3 or 4, 4); the last frame's per-feature arrays (see the header).
    Returns dict(kps, desc, match, owner, outlier, pose7, nmatches, n_inliers, n_correspondences, greedy_rounds)."""
    L = _lib.load()
    img = np.ascontiguousarray(image, np.uint8)
    h, w = img.shape
    K4 = np.ascontiguousarray(K4, np.float32); bounds = np.ascontiguousarray(bounds, np.float32)
    T = np.ascontiguousarray(np.asarray(Tcw_pred, np.float64).reshape(-1)[:12])
    X = np.ascontiguousarray(last_Xw, np.float64).reshape(-1, 3); n = len(X)
    D = np.ascontiguousarray(last_desc, np.uint8).reshape(-1, 32); O = np.ascontiguousarray(last_octave, np.int32)
    A = np.ascontiguousarray(last_angle, np.float32); V = np.ascontiguousarray(last_valid, np.uint8)
    assert len(D) == n and len(O) == n and len(A) == n and len(V) == n
    cap = extractor.max_keypoints
    # the output buffers (and their ctypes pointers) live with the extractor: a per-frame call must not spend its time in
    # allocations (the C side writes every entry it reports; what is returned are copies of the used parts)
    B = getattr(extractor, "_track_bufs", None)
    if B is None or B["cap"] != cap or B["nq"] < n:
        nq = max(n, 1, B["nq"] if B else 0)
        B = dict(cap=cap, nq=nq, kps=np.zeros(cap, KP_DTYPE), desc=np.zeros((cap, 32), np.uint8), match=np.full(nq, -1, np.int32),
                 owner=np.full(cap, -1, np.int32), outl=np.zeros(cap, np.uint8), res=TrackResult())
        B["p"] = tuple(_lib.ptr(B[k]) for k in ("kps", "desc", "match", "owner", "outl"))
        B["pres"] = C.byref(B["res"])
        extractor._track_bufs = B
    pk, pd, pm, po, pl = B["p"]
    res = B["res"]
    _lib.check(L.orbt_track_with_motion_model(extractor._h, _lib.ptr(img), w, h, img.strides[0], _lib.ptr(K4), _lib.ptr(bounds), _lib.ptr(T), _lib.ptr(X),
                                              _lib.ptr(D), _lib.ptr(O), _lib.ptr(A), _lib.ptr(V), n, float(th), int(bool(check_ori)), pk, pd, cap, pm, po, pl,
                                              B["pres"]), "orbt_track_with_motion_model")
    k = res.n_keypoints
    return dict(kps=B["kps"][:k].copy(), desc=B["desc"][:k].copy(), match=B["match"][:n].copy(), owner=B["owner"][:k].copy(),
                outlier=B["outl"][:k].astype(bool), pose7=np.array(res.pose7[:], np.float64), nmatches=res.nmatches, n_inliers=res.n_inliers,
                n_correspondences=res.n_correspondences, greedy_rounds=res.greedy_rounds)
