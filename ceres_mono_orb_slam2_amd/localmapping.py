"""Host mirror of the LocalMapping thread's device-resident steps (include/orbslam_hip.h: orbl_*; reference
src/LocalMapping.cc:196-396 CreateNewMapPoints, :398-505 SearchInNeighbors).  Thin ctypes layer: arrays in, arrays out."""
import ctypes as C

import numpy as np

from . import _lib


class _KF(C.Structure):              # orbl_keyframe
    _fields_ = [("kps", C.c_void_p), ("desc", C.c_void_p), ("unmapped", C.c_void_p), ("n", C.c_int),
                ("fv_node", C.c_void_p), ("fv_off", C.c_void_p), ("fv_idx", C.c_void_p), ("fv_n", C.c_int),
                ("Tcw", C.c_double * 12), ("K4", C.c_float * 4), ("F12", C.c_double * 9), ("ex", C.c_float), ("ey", C.c_float)]


class _FuseKF(C.Structure):          # orbl_fuse_keyframe
    _fields_ = [("kps", C.c_void_p), ("desc", C.c_void_p), ("n", C.c_int), ("bounds", C.c_float * 4)]


def _c(a, dt):
    return np.ascontiguousarray(a, dt)


def _addr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def prepare_create_new_map_points(cur, neighbours, scale_factors, level_sigma2, ratio_factor, stop=None):
    """Marshals the arguments of orbl_create_new_map_points once and returns call() -> (match12, ok, x3D, n_processed): what a C++
    caller holds anyway (tools/api_latency.py times call() alone)."""
    L = _lib.load()
    keep = []                                                   # the arrays the structs point at

    def hold(a, dt):
        a = _c(a, dt); keep.append(a); return a
    k1 = hold(cur["kps"], np.float32).reshape(-1, 4); n1 = len(k1)
    d1 = hold(cur["desc"], np.uint8).reshape(-1, 32)
    u1 = hold(cur["unmapped"], np.uint8) if cur.get("unmapped") is not None else None
    f1 = [hold(x, np.uint32) for x in cur["fv"]]
    T1 = hold(np.asarray(cur["Tcw"], np.float64).reshape(-1)[:12], np.float64); K1 = hold(cur["K4"], np.float32)
    nb = (_KF * max(len(neighbours), 1))()
    for k, q in enumerate(neighbours):
        kk = hold(q["kps"], np.float32).reshape(-1, 4); dd = hold(q["desc"], np.uint8).reshape(-1, 32)
        uu = hold(q["unmapped"], np.uint8) if q.get("unmapped") is not None else None
        ff = [hold(x, np.uint32) for x in q["fv"]]
        assert len(dd) == len(kk) and len(ff[1]) == len(ff[0]) + 1
        s = nb[k]
        s.kps, s.desc, s.unmapped, s.n = kk.ctypes.data, dd.ctypes.data, (uu.ctypes.data if uu is not None else None), len(kk)
        s.fv_node, s.fv_off, s.fv_idx, s.fv_n = ff[0].ctypes.data, ff[1].ctypes.data, ff[2].ctypes.data, len(ff[0])
        s.Tcw[:] = np.asarray(q["Tcw"], np.float64).reshape(-1)[:12].tolist(); s.K4[:] = _c(q["K4"], np.float32).tolist()
        s.F12[:] = np.asarray(q["F12"], np.float64).reshape(9).tolist()
        s.ex, s.ey = float(q["epipole"][0]), float(q["epipole"][1])
    sf = hold(scale_factors, np.float32); ls = hold(level_sigma2, np.float32)
    nn = len(neighbours)
    m = np.full((max(nn, 1), max(n1, 1)), -1, np.int32); ok = np.zeros((max(nn, 1), max(n1, 1)), np.uint8); X = np.zeros((max(nn, 1), max(n1, 1), 3), np.float64)
    npr = C.c_int(0)
    st = None
    if stop is not None:
        assert stop.dtype == np.uint8 and stop.size >= 1
        st = C.c_void_p(stop.ctypes.data)
    L.orbl_create_new_map_points.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    a = (_addr(k1), _addr(d1), _addr(u1), n1, _addr(f1[0]), _addr(f1[1]), _addr(f1[2]), len(f1[0]), _addr(T1), _addr(K1), C.cast(nb, C.c_void_p), nn, _addr(sf), _addr(ls),
         len(sf), float(ratio_factor), st, _addr(m), _addr(ok), _addr(X), C.byref(npr))

    def call():
        _lib.check(L.orbl_create_new_map_points(*a), "orbl_create_new_map_points")
        return m[:nn, :n1], ok[:nn, :n1].view(np.bool_), X[:nn, :n1], npr.value
    call._keep = (keep, nb)
    return call


def create_new_map_points(cur, neighbours, scale_factors, level_sigma2, ratio_factor, stop=None):
    """LocalMapping::CreateNewMapPoints for the current keyframe and its neighbours, in order, in one call.
    cur = dict(kps[n1,4] {x, y, octave, angle}, desc[n1,32], unmapped[n1] (or None), fv=(nodes, offsets, indices), Tcw[3,4], K4);
    each neighbour the same plus F12[3,3] and epipole=(ex, ey).  stop: optional 1-element uint8 array (CheckNewKeyFrames).
    Returns (match12[nb, n1] int32, ok[nb, n1] bool, x3D[nb, n1, 3] float64, n_processed)."""
    return prepare_create_new_map_points(cur, neighbours, scale_factors, level_sigma2, ratio_factor, stop)()


def fuse_batch(keyframes, q_uv, q_radius, q_level, mp_desc, inv_level_sigma2, sim3=False, n_levels=8):
    """ORBmatcher::Fuse candidate selection for T keyframes x M map points (orbl_fuse_batch; sim3=True: orbl_fuse_batch_sim3, the form of
    LoopClosing::SearchAndFuse without the chi-square gate, inv_level_sigma2 unused).  keyframes: dicts(kps[n,4], desc[n,32],
    bounds[4]); q_uv[T,M,2], q_radius[T,M], q_level[T,M] (-1: not projected), mp_desc[M,32].  Returns (best_idx[T,M], best_dist[T,M])."""
    L = _lib.load()
    keep = []
    T = len(keyframes)
    if T == 0:
        return np.zeros((0, 0), np.int32), np.zeros((0, 0), np.int32)
    kf = (_FuseKF * max(T, 1))()
    for t, q in enumerate(keyframes):
        kk = _c(q["kps"], np.float32).reshape(-1, 4); dd = _c(q["desc"], np.uint8).reshape(-1, 32); keep += [kk, dd]
        kf[t].kps, kf[t].desc, kf[t].n = kk.ctypes.data, dd.ctypes.data, len(kk)
        kf[t].bounds[:] = _c(q["bounds"], np.float32).tolist()
    uv = _c(q_uv, np.float32).reshape(T, -1, 2); M = uv.shape[1]
    rad = _c(q_radius, np.float32).reshape(T, M); lvl = _c(q_level, np.int32).reshape(T, M)
    md = _c(mp_desc, np.uint8).reshape(M, 32)
    bi = np.full((max(T, 1), max(M, 1)), -1, np.int32); bd = np.full((max(T, 1), max(M, 1)), 256, np.int32)
    if sim3:
        L.orbl_fuse_batch_sim3.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _lib.check(L.orbl_fuse_batch_sim3(C.cast(kf, C.c_void_p), T, _addr(uv), _addr(rad), _addr(lvl), M, _addr(md), int(n_levels), _addr(bi), _addr(bd)),
                   "orbl_fuse_batch_sim3")
        return bi[:T, :M], bd[:T, :M]
    ils = _c(inv_level_sigma2, np.float32)
    L.orbl_fuse_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    _lib.check(L.orbl_fuse_batch(C.cast(kf, C.c_void_p), T, _addr(uv), _addr(rad), _addr(lvl), M, _addr(md), _addr(ils), len(ils), _addr(bi), _addr(bd)),
               "orbl_fuse_batch")
    return bi[:T, :M], bd[:T, :M]
