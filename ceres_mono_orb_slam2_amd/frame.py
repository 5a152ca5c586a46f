"""Host mirror of the Frame-side steps around extract -> match (reference src/Frame.cc:158-173, 191-241, 243-355;
src/MapPoint.cc:406-420) over the HIP C ABI.  kps4 = n x (x, y, octave, angle) float32 of the UNDISTORTED keypoints,
bounds = (min_x, max_x, min_y, max_y)."""
import ctypes as C

import numpy as np

from . import _lib

FRAME_GRID_COLS, FRAME_GRID_ROWS = 64, 48          # include/Frame.h:44-45


def UndistortKeyPoints(xy, K4, dist5):
    """Frame::UndistortKeyPoints (src/Frame.cc:329-355); dist5 = (k1, k2, p1, p2, k3)."""
    L = _lib.load()
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2); out = np.zeros_like(xy)
    K4 = np.ascontiguousarray(K4, np.float32); d = np.ascontiguousarray(dist5, np.float32)
    _lib.check(L.orbm_undistort_keypoints(_lib.ptr(xy), len(xy), _lib.ptr(K4), _lib.ptr(d), _lib.ptr(out)), "orbm_undistort_keypoints")
    return out


def AssignFeaturesToGrid(kps4, bounds):
    """Frame::AssignFeaturesToGrid (src/Frame.cc:158-173) -> (cell_offsets[64*48+1], cell_idx), cell = x * 48 + y."""
    L = _lib.load()
    k = np.ascontiguousarray(kps4, np.float32).reshape(-1, 4); b = np.ascontiguousarray(bounds, np.float32)
    off = np.zeros(FRAME_GRID_COLS * FRAME_GRID_ROWS + 1, np.uint32); idx = np.zeros(max(len(k), 1), np.uint32)
    n = C.c_int(0)
    _lib.check(L.orbm_assign_features_to_grid(_lib.ptr(k), len(k), _lib.ptr(b), _lib.ptr(off), _lib.ptr(idx), C.byref(n)),
               "orbm_assign_features_to_grid")
    return off, idx[:n.value]


def GetFeaturesInArea(kps4, bounds, q_xy, q_radius, q_min_level=None, q_max_level=None):
    """Frame::GetFeaturesInArea (src/Frame.cc:243-307) for a list of queries -> CSR (offsets[nq+1], idx) in reference order."""
    L = _lib.load()
    k = np.ascontiguousarray(kps4, np.float32).reshape(-1, 4); b = np.ascontiguousarray(bounds, np.float32)
    q = np.ascontiguousarray(q_xy, np.float32).reshape(-1, 2); r = np.ascontiguousarray(q_radius, np.float32)
    mn = None if q_min_level is None else np.ascontiguousarray(q_min_level, np.int32)
    mx = None if q_max_level is None else np.ascontiguousarray(q_max_level, np.int32)
    nq = len(q)
    off = np.zeros(nq + 1, np.uint32); tot = C.c_int(0)
    _lib.check(L.orbm_features_in_area(_lib.ptr(k), len(k), _lib.ptr(b), _lib.ptr(q), _lib.ptr(r), _lib.ptr(mn), _lib.ptr(mx), nq, _lib.ptr(off),
                                       None, 0, C.byref(tot)), "orbm_features_in_area")
    idx = np.zeros(max(tot.value, 1), np.uint32)
    _lib.check(L.orbm_features_in_area(_lib.ptr(k), len(k), _lib.ptr(b), _lib.ptr(q), _lib.ptr(r), _lib.ptr(mn), _lib.ptr(mx), nq, _lib.ptr(off),
                                       _lib.ptr(idx), len(idx), C.byref(tot)), "orbm_features_in_area")
    return off, idx[:tot.value]


def isInFrustum(Rcw, tcw, K4, bounds, P, Pn, min_dist, max_dist, viewing_cos_limit, log_scale_factor, n_levels):
    """Frame::isInFrustum (src/Frame.cc:191-241) for n map points.  Returns (in_view, uv, level, view_cos)."""
    L = _lib.load()
    R = np.ascontiguousarray(Rcw, np.float64).reshape(9); t = np.ascontiguousarray(tcw, np.float64)
    K4 = np.ascontiguousarray(K4, np.float32); b = np.ascontiguousarray(bounds, np.float32)
    P = np.ascontiguousarray(P, np.float64).reshape(-1, 3); Pn = np.ascontiguousarray(Pn, np.float64).reshape(-1, 3)
    mn = np.ascontiguousarray(min_dist, np.float32); mx = np.ascontiguousarray(max_dist, np.float32)
    n = len(P)
    iv = np.zeros(max(n, 1), np.uint8); uv = np.zeros((max(n, 1), 2), np.float32); lv = np.zeros(max(n, 1), np.int32); vc = np.zeros(max(n, 1), np.float32)
    _lib.check(L.orbm_is_in_frustum(_lib.ptr(R), _lib.ptr(t), _lib.ptr(K4), _lib.ptr(b), _lib.ptr(P), _lib.ptr(Pn), _lib.ptr(mn), _lib.ptr(mx), n,
                                    float(viewing_cos_limit), float(log_scale_factor), int(n_levels), _lib.ptr(iv), _lib.ptr(uv), _lib.ptr(lv),
                                    _lib.ptr(vc)), "orbm_is_in_frustum")
    return iv[:n], uv[:n], lv[:n], vc[:n]


def TriangulateMatches(Tcw1, Tcw2, K1, K2, kp1, kp2, level_sigma2, scale_factors, ratio_factor):
    """Per-match body of LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:267-378).  kp = n x (x, y, octave).
    Returns (x3D [n, 3] float64, ok [n] uint8)."""
    L = _lib.load()
    T1 = np.ascontiguousarray(Tcw1, np.float64).reshape(12); T2 = np.ascontiguousarray(Tcw2, np.float64).reshape(12)
    K1 = np.ascontiguousarray(K1, np.float32); K2 = np.ascontiguousarray(K2, np.float32)
    k1 = np.ascontiguousarray(kp1, np.float32).reshape(-1, 3); k2 = np.ascontiguousarray(kp2, np.float32).reshape(-1, 3)
    ls = np.ascontiguousarray(level_sigma2, np.float32); sf = np.ascontiguousarray(scale_factors, np.float32)
    n = len(k1); X = np.zeros((max(n, 1), 3)); ok = np.zeros(max(n, 1), np.uint8)
    _lib.check(L.orbm_triangulate_matches(_lib.ptr(T1), _lib.ptr(T2), _lib.ptr(K1), _lib.ptr(K2), _lib.ptr(k1), _lib.ptr(k2), n, _lib.ptr(ls),
                                          _lib.ptr(sf), len(sf), float(ratio_factor), _lib.ptr(X), _lib.ptr(ok)), "orbm_triangulate_matches")
    return X[:n], ok[:n]
