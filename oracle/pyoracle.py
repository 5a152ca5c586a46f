"""ctypes binding of the CPU oracle (oracle/_build/liborb_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liborb_oracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("orb_oracle.cpp", "match_oracle.cpp", "ba_oracle.cpp", "sim3_oracle.cpp", "bow_oracle.cpp", "tri_oracle.cpp",
                                             "orb_pattern_data.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _SO


CANONICAL_FLAGS = "-O3 -march=x86-64-v2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math"     # (oracle/Makefile; runs on any host)
NATIVE_FLAGS = "-O3 -march=native -std=c++17 -fPIC -ffp-contract=off -fno-fast-math"            # the CPU-baseline build


def build_native():
    """The timing build of SURVEY 8(d) (`-O3 -march=native -ffp-contract=off`): compiled ON the host that runs it (an
    -march=native object must not travel between machines), into a directory named after this host's CPU flags.
    Returns the path, or None when the compiler is missing / fails.  bench.py's cpu_baseline leg times this build after
    checking that it reproduces the canonical build bit for bit."""
    import hashlib
    try:
        flags = [l for l in open("/proc/cpuinfo") if l.startswith("flags")][0]
    except Exception:
        flags = "unknown"
    d = os.path.join(_HERE, "_build", "native_" + hashlib.sha1(flags.encode()).hexdigest()[:10])
    so = os.path.join(d, "liborb_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("orb_oracle.cpp", "match_oracle.cpp", "ba_oracle.cpp", "sim3_oracle.cpp", "bow_oracle.cpp", "tri_oracle.cpp")]
    if os.path.exists(so) and all(os.path.getmtime(s) <= os.path.getmtime(so) for s in srcs):
        return so
    try:
        os.makedirs(d, exist_ok=True)
        subprocess.check_call(["g++"] + NATIVE_FLAGS.split() + ["-Wall", "-Wno-unused-function", "-pthread", "-shared", "-o", so] + srcs,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return so
    except Exception:
        return None


_lib = None


def use_library(path=None):
    """Switch every function of this module to another build of the oracle (None = the canonical one).  Objects created
    before the switch (OracleExtractor) keep working only with the library that made them: create new ones."""
    global _lib, _SO
    _SO = path or os.path.join(_HERE, "_build", "liborb_oracle.so")
    _lib = None
    return lib()


def lib():
    global _lib
    if _lib is None:
        if _SO == os.path.join(_HERE, "_build", "liborb_oracle.so"):
            build()
        L = C.CDLL(_SO)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orc_extract.restype = C.c_int
        L.orc_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_level_dims.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_level_image.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_level_num_candidates.argtypes = [C.c_void_p, C.c_int]
        L.orc_level_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_level_num_keypoints.argtypes = [C.c_void_p, C.c_int]
        L.orc_level_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_gaussian_blur7.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_gauss7_taps.argtypes = [C.c_void_p]
        L.orc_fast.restype = C.c_int
        L.orc_fast.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_octree.restype = C.c_int
        L.orc_octree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_det_sincos.argtypes = [C.c_double, C.c_void_p, C.c_void_p]
        L.orc_ic_angle.restype = C.c_float
        L.orc_ic_angle.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_brief.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.orc_pattern.restype = C.c_void_p
        # matcher
        L.orc_descriptor_distance.restype = C.c_int
        L.orc_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_hamming_best2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_three_maxima.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_rot_bin.restype = C.c_int
        L.orc_rot_bin.argtypes = [C.c_float, C.c_float]
        L.orc_match_frames.restype = C.c_int
        L.orc_match_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_float, C.c_int, C.c_int, C.c_void_p]
        L.orc_features_in_area.restype = C.c_int
        L.orc_features_in_area.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_search_for_initialization.restype = C.c_int
        L.orc_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                    C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.orc_search_by_projection.restype = C.c_int
        L.orc_search_by_projection.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 9 + [C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_search_by_bow.restype = C.c_int
        L.orc_search_by_bow.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_search_for_triangulation.restype = C.c_int
        L.orc_search_for_triangulation.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        # BA
        L.orc_ba_solve.restype = C.c_int
        L.orc_ba_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_check_outlier.restype = C.c_int
        L.orc_check_outlier.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        L.orc_pose_optimization.restype = C.c_int
        L.orc_pose_optimization.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_void_p]
        L.orc_local_ba.restype = C.c_int
        L.orc_local_ba.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_ba_eval_obs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_double,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_matrix4d_to_pose7.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_pose7_to_matrix4d.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_quat_plus.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_quat_rotate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        # Sim3
        for f in ("orc_sim3_exp", "orc_sim3_log", "orc_sim3_inverse"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p]
        L.orc_sim3_plus.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_sim3_act.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_sim3_eval_term.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_optimize_sim3.restype = C.c_int
        L.orc_optimize_sim3.argtypes = [C.c_void_p] * 9 + [C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class BaOpts(C.Structure):
    _fields_ = [("max_iters", C.c_int), ("huber_delta", C.c_double), ("fix_points", C.c_int), ("stop", C.c_void_p)]


class BaSummary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("iterations", C.c_int),
                ("successful_steps", C.c_int), ("termination", C.c_int), ("final_radius", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class OracleExtractor:
    """ORB_SLAM2::ORBextractor restated (reference include/ORBextractor.h:45-111)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self.h = self.L.orc_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        sc = np.zeros(nlevels, np.float32); isc = sc.copy(); s2 = sc.copy(); is2 = sc.copy()
        q = np.zeros(nlevels, np.int32); um = np.zeros(16, np.int32)
        self.L.orc_tables(self.h, _p(sc), _p(isc), _p(s2), _p(is2), _p(q), _p(um))
        self.scale, self.inv_scale, self.sigma2, self.inv_sigma2, self.quota, self.umax = sc, isc, s2, is2, q, um

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    def extract(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        # per level at most max(quota + 3, 4 * nIni) keypoints (the first DistributeOctTree sweep splits all nIni = round(W/H)
        # initial nodes unconditionally), so size for the aspect ratio as well
        cap = self.nfeatures + (4 * max(1, int(round(w / max(h, 1)))) + 8) * self.nlevels + 64
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = self.L.orc_extract(self.h, _p(img), w, h, w, _p(kps), _p(desc), cap)
        assert n >= 0
        return kps[:n].copy(), desc[:n].copy()

    def level_image(self, level, blurred=False):
        w = C.c_int(); h = C.c_int()
        self.L.orc_level_dims(self.h, level, C.byref(w), C.byref(h))
        out = np.zeros((h.value, w.value), np.uint8)
        self.L.orc_level_image(self.h, level, int(blurred), _p(out))
        return out

    def level_candidates(self, level):
        n = self.L.orc_level_num_candidates(self.h, level)
        out = np.zeros((n, 3), np.int32)
        if n:
            self.L.orc_level_candidates(self.h, level, _p(out))
        return out

    def level_keypoints(self, level):
        n = self.L.orc_level_num_keypoints(self.h, level)
        out = np.zeros(n, KP_DTYPE)
        if n:
            self.L.orc_level_keypoints(self.h, level, _p(out))
        return out


def resize_linear_u8(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    out = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], _p(out), dw, dh)
    return out


def gaussian_blur7(src):
    src = np.ascontiguousarray(src, np.uint8)
    out = np.zeros_like(src)
    lib().orc_gaussian_blur7(_p(src), src.shape[1], src.shape[0], _p(out))
    return out


def gauss7_taps():
    t = np.zeros(7, np.int32)
    lib().orc_gauss7_taps(_p(t))
    return t


def fast(src, threshold):
    src = np.ascontiguousarray(src, np.uint8)
    cap = src.size // 4 + 16
    out = np.zeros((cap, 3), np.int32)
    n = lib().orc_fast(_p(src), src.shape[1], src.shape[0], threshold, _p(out), cap)
    return out[:n].copy()


def octree(cands, minX, maxX, minY, maxY, N):
    cands = np.ascontiguousarray(cands, np.int32)
    cap = len(cands) + 8
    out = np.zeros((cap, 3), np.int32)
    n = lib().orc_octree(_p(cands), len(cands), minX, maxX, minY, maxY, N, _p(out), cap)
    return out[:n].copy()


def fast_atan2(y, x):
    return lib().orc_fast_atan2(float(y), float(x))


def det_sincos(x):
    s = C.c_double(); c = C.c_double()
    lib().orc_det_sincos(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def ic_angle(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    return lib().orc_ic_angle(_p(img), img.shape[1], img.shape[0], int(x), int(y))


def brief(img, x, y, angle_deg):
    img = np.ascontiguousarray(img, np.uint8)
    d = np.zeros(32, np.uint8)
    lib().orc_brief(_p(img), img.shape[1], img.shape[0], int(x), int(y), float(angle_deg), _p(d))
    return d


def pattern():
    return np.ctypeslib.as_array(C.cast(lib().orc_pattern(), C.POINTER(C.c_int8)), shape=(1024,)).copy()


# ------------------------------- matcher -------------------------------------
def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_descriptor_distance(_p(a), _p(b))


def set_blur_variant(v):
    """0: OpenCV <= 3.4.1 GaussianBlur taps (default), 1: the ufixedpoint16 taps of later versions (process-wide)"""
    r = lib().orc_set_blur_variant(int(v))
    assert r == 0


def hamming_best2(q, t, cand_offsets=None, cand_idx=None):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    nq, nt = len(q), len(t)
    bi = np.zeros(nq, np.int32); bd = np.zeros(nq, np.int32); sd = np.zeros(nq, np.int32)
    co = None if cand_offsets is None else np.ascontiguousarray(cand_offsets, np.uint32)
    ci = None if cand_idx is None else np.ascontiguousarray(cand_idx, np.uint32)
    lib().orc_hamming_best2(_p(q), nq, _p(t), nt, _p(co), _p(ci), _p(bi), _p(bd), _p(sd))
    return bi, bd, sd


def three_maxima(cnt):
    cnt = np.ascontiguousarray(cnt, np.int32)
    ind = np.zeros(3, np.int32)
    lib().orc_three_maxima(_p(cnt), len(cnt), _p(ind))
    return ind


def rot_bin(a1, a2):
    return lib().orc_rot_bin(float(a1), float(a2))


def match_frames(d1, ang1, d2, ang2, ratio=0.9, th=50, check_ori=True):
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    a1 = np.ascontiguousarray(ang1, np.float32); a2 = np.ascontiguousarray(ang2, np.float32)
    m = np.zeros(len(d1), np.int32)
    n = lib().orc_match_frames(_p(d1), _p(a1), len(d1), _p(d2), _p(a2), len(d2), ratio, th, int(check_ori), _p(m))
    return m, n


def features_in_area(kps4, bounds, qxy, qr, qminl, qmaxl):
    kps4 = np.ascontiguousarray(kps4, np.float32); bounds = np.ascontiguousarray(bounds, np.float32)
    qxy = np.ascontiguousarray(qxy, np.float32); qr = np.ascontiguousarray(qr, np.float32)
    qminl = np.ascontiguousarray(qminl, np.int32); qmaxl = np.ascontiguousarray(qmaxl, np.int32)
    nq = len(qr)
    off = np.zeros(nq + 1, np.uint32)
    tot = lib().orc_features_in_area(_p(kps4), len(kps4), _p(bounds), _p(qxy), _p(qr), _p(qminl), _p(qmaxl), nq,
                                     _p(off), None, 0)
    idx = np.zeros(max(tot, 1), np.uint32)
    lib().orc_features_in_area(_p(kps4), len(kps4), _p(bounds), _p(qxy), _p(qr), _p(qminl), _p(qmaxl), nq,
                               _p(off), _p(idx), tot)
    return off, idx[:tot]


def search_for_initialization(kps1, d1, kps2, d2, bounds2, prev_matched, window=100, nnratio=0.9, check_ori=True):
    kps1 = np.ascontiguousarray(kps1, np.float32); kps2 = np.ascontiguousarray(kps2, np.float32)
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    b = np.ascontiguousarray(bounds2, np.float32)
    pm = np.ascontiguousarray(prev_matched, np.float32).copy()
    m = np.zeros(len(kps1), np.int32)
    n = lib().orc_search_for_initialization(_p(kps1), _p(d1), len(kps1), _p(kps2), _p(d2), len(kps2), _p(b), _p(pm),
                                            window, nnratio, int(check_ori), _p(m))
    return m, n, pm


def _opt(a, dt):
    return None if a is None else np.ascontiguousarray(a, dt)


def search_by_projection(kps4, desc, bounds, q_uv, q_radius, q_desc, q_min_level=None, q_max_level=None, q_pred_level=None,
                         q_valid=None, q_angle=None, inv_level_sigma2=None, chi2_gate=0.0, taken=None, mode_best2=False,
                         ratio=0.8, th=100, check_ori=False):
    kps4 = np.ascontiguousarray(kps4, np.float32); desc = np.ascontiguousarray(desc, np.uint8)
    b = np.ascontiguousarray(bounds, np.float32); q_uv = np.ascontiguousarray(q_uv, np.float32)
    q_radius = np.ascontiguousarray(q_radius, np.float32); q_desc = np.ascontiguousarray(q_desc, np.uint8)
    nq = len(q_radius)
    mn = _opt(q_min_level, np.int32); mx = _opt(q_max_level, np.int32); pl = _opt(q_pred_level, np.int32)
    qv = _opt(q_valid, np.uint8); qa = _opt(q_angle, np.float32); isg = _opt(inv_level_sigma2, np.float32)
    tk = None if taken is None else np.ascontiguousarray(taken, np.uint8).copy()
    m = np.zeros(nq, np.int32); bd = np.zeros(nq, np.int32)
    n = lib().orc_search_by_projection(_p(kps4), _p(desc), len(kps4), _p(b), _p(q_uv), _p(q_radius), _p(mn), _p(mx), _p(pl),
                                       _p(q_desc), _p(qv), _p(qa), nq, _p(isg), float(chi2_gate), _p(tk), int(mode_best2),
                                       float(ratio), int(th), int(check_ori), _p(m), _p(bd))
    return n, m, bd, tk


def search_by_sim3(kps1, desc1, kps2, desc2, bounds, q12_uv, q12_radius, q12_pred, q12_valid, q21_uv, q21_radius, q21_pred, q21_valid,
                   q12_desc=None, q21_desc=None, bounds2=None):
    c = np.ascontiguousarray
    k1, k2, d1, d2, b = c(kps1, np.float32), c(kps2, np.float32), c(desc1, np.uint8), c(desc2, np.uint8), c(bounds, np.float32)
    b2 = b if bounds2 is None else c(bounds2, np.float32)
    od = lambda d: None if d is None else c(d, np.uint8)
    a = [c(q12_uv, np.float32), c(q12_radius, np.float32), c(q12_pred, np.int32), c(q12_valid, np.uint8), od(q12_desc),
         c(q21_uv, np.float32), c(q21_radius, np.float32), c(q21_pred, np.int32), c(q21_valid, np.uint8), od(q21_desc)]
    m = np.zeros(len(k1), np.int32)
    L = lib()
    L.orc_search_by_sim3.restype = C.c_int
    L.orc_search_by_sim3.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 13
    n = L.orc_search_by_sim3(_p(k1), _p(d1), len(k1), _p(k2), _p(d2), len(k2), _p(b), _p(b2), *[_p(x) for x in a], _p(m))
    return n, m


def search_by_bow(desc1, valid1, angle1, desc2, valid2, angle2, fv1, fv2, ratio=0.7, th=50, strict=False, check_ori=True):
    """fv = (node ids ascending uint32, offsets uint32 [nnodes+1], indices uint32)"""
    d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
    v1 = _opt(valid1, np.uint8); v2 = _opt(valid2, np.uint8); a1 = _opt(angle1, np.float32); a2 = _opt(angle2, np.float32)
    f1 = [np.ascontiguousarray(x, np.uint32) for x in fv1]; f2 = [np.ascontiguousarray(x, np.uint32) for x in fv2]
    m = np.zeros(len(d1), np.int32)
    n = lib().orc_search_by_bow(_p(d1), len(d1), _p(v1), _p(a1), _p(d2), len(d2), _p(v2), _p(a2), _p(f1[0]), _p(f1[1]), _p(f1[2]),
                                len(f1[0]), _p(f2[0]), _p(f2[1]), _p(f2[2]), len(f2[0]), float(ratio), int(th), int(strict),
                                int(check_ori), _p(m))
    return n, m


def search_for_triangulation(kps1, desc1, unmapped1, kps2, desc2, unmapped2, fv1, fv2, F12, epipole, scale_factors, level_sigma2,
                             check_ori=False):
    k1 = np.ascontiguousarray(kps1, np.float32); k2 = np.ascontiguousarray(kps2, np.float32)
    d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
    u1 = _opt(unmapped1, np.uint8); u2 = _opt(unmapped2, np.uint8)
    f1 = [np.ascontiguousarray(x, np.uint32) for x in fv1]; f2 = [np.ascontiguousarray(x, np.uint32) for x in fv2]
    F = np.ascontiguousarray(F12, np.float64).reshape(9); sf = np.ascontiguousarray(scale_factors, np.float32)
    ls = np.ascontiguousarray(level_sigma2, np.float32)
    m = np.zeros(len(k1), np.int32)
    n = lib().orc_search_for_triangulation(_p(k1), _p(d1), _p(u1), len(k1), _p(k2), _p(d2), _p(u2), len(k2), _p(f1[0]), _p(f1[1]),
                                           _p(f1[2]), len(f1[0]), _p(f2[0]), _p(f2[1]), _p(f2[2]), len(f2[0]), _p(F),
                                           float(epipole[0]), float(epipole[1]), _p(sf), _p(ls), int(check_ori), _p(m))
    return n, m


# --------------------------------- BA ----------------------------------------
def ba_solve(K4, poses7, cam_fixed, pts3, obs_cam, obs_pt, obs_uv, obs_w, obs_robust, max_iters,
             huber_delta=np.sqrt(5.991), fix_points=False, stop=None):
    K4 = np.ascontiguousarray(K4, np.float64); poses = np.ascontiguousarray(poses7, np.float64).copy()
    cf = np.ascontiguousarray(cam_fixed, np.uint8); pts = np.ascontiguousarray(pts3, np.float64).copy()
    oc = np.ascontiguousarray(obs_cam, np.int32); op = np.ascontiguousarray(obs_pt, np.int32)
    uv = np.ascontiguousarray(obs_uv, np.float64); w = np.ascontiguousarray(obs_w, np.float64)
    rb = np.ascontiguousarray(obs_robust, np.uint8)
    o = BaOpts(int(max_iters), float(huber_delta), int(fix_points), _p(stop) if stop is not None else None)
    s = BaSummary()
    lib().orc_ba_solve(_p(K4), _p(poses), _p(cf), len(cf), _p(pts), len(pts), _p(oc), _p(op), _p(uv), _p(w), _p(rb),
                       len(oc), C.byref(o), C.byref(s))
    return poses, pts, s.as_dict()


def pose_optimization(K4, pose7, Xw, uv, inv_sigma2):
    K4 = np.ascontiguousarray(K4, np.float64); pose = np.ascontiguousarray(pose7, np.float64).copy()
    Xw = np.ascontiguousarray(Xw, np.float64); uv = np.ascontiguousarray(uv, np.float64)
    isg = np.ascontiguousarray(inv_sigma2, np.float32)
    out = np.zeros(len(Xw), np.uint8)
    s = BaSummary()
    n = lib().orc_pose_optimization(_p(K4), _p(pose), _p(Xw), _p(uv), _p(isg), len(Xw), _p(out), C.byref(s))
    return n, pose, out, s.as_dict()


def local_ba(K4, poses7, cam_fixed, cam_local, pts3, obs_cam, obs_pt, obs_uv, obs_inv_sigma2, stop=None,
             duplicate_blocks=True):
    K4 = np.ascontiguousarray(K4, np.float64); poses = np.ascontiguousarray(poses7, np.float64).copy()
    cf = np.ascontiguousarray(cam_fixed, np.uint8); cl = np.ascontiguousarray(cam_local, np.uint8)
    pts = np.ascontiguousarray(pts3, np.float64).copy()
    oc = np.ascontiguousarray(obs_cam, np.int32); op = np.ascontiguousarray(obs_pt, np.int32)
    uv = np.ascontiguousarray(obs_uv, np.float64); isg = np.ascontiguousarray(obs_inv_sigma2, np.float32)
    er = np.zeros(len(oc), np.uint8)
    s1 = BaSummary(); s2 = BaSummary()
    rc = lib().orc_local_ba(_p(K4), _p(poses), _p(cf), _p(cl), len(cf), _p(pts), len(pts), _p(oc), _p(op), _p(uv),
                            _p(isg), len(oc), _p(stop) if stop is not None else None, int(duplicate_blocks), _p(er),
                            C.byref(s1), C.byref(s2))
    return rc, poses, pts, er, s1.as_dict(), s2.as_dict()


def matrix4d_to_pose7(T):
    a = np.ascontiguousarray(T, np.float64).reshape(4, 4); o = np.zeros(7)
    lib().orc_matrix4d_to_pose7(_p(a), _p(o))
    return o


def pose7_to_matrix4d(p7):
    a = np.ascontiguousarray(p7, np.float64); o = np.zeros((4, 4))
    lib().orc_pose7_to_matrix4d(_p(a), _p(o))
    return o


def set_ba_threads(n):
    """Worker threads of the BA oracle's evaluation / Schur elimination (the reference's LocalBA uses 4,
    src/CeresOptimizer.cc:516).  Results are bit-identical for any count.  Returns the previous value."""
    return lib().orc_set_ba_threads(int(n))


def ba_eval_obs(K4, pose7, X, uv, w, robust=False, huber_delta=np.sqrt(5.991)):
    K4 = np.ascontiguousarray(K4, np.float64); pose7 = np.ascontiguousarray(pose7, np.float64)
    X = np.ascontiguousarray(X, np.float64); uv = np.ascontiguousarray(uv, np.float64)
    r = np.zeros(2); Jc = np.zeros((2, 6)); Jp = np.zeros((2, 3)); rho = C.c_double()
    lib().orc_ba_eval_obs(_p(K4), _p(pose7), _p(X), _p(uv), float(w), int(robust), float(huber_delta), _p(r), _p(Jc),
                          _p(Jp), C.byref(rho))
    return r, Jc, Jp, rho.value


def quat_plus(q, d):
    q = np.ascontiguousarray(q, np.float64); d = np.ascontiguousarray(d, np.float64); o = np.zeros(4)
    lib().orc_quat_plus(_p(q), _p(d), _p(o))
    return o


def quat_rotate(q, v):
    q = np.ascontiguousarray(q, np.float64); v = np.ascontiguousarray(v, np.float64); o = np.zeros(3)
    lib().orc_quat_rotate(_p(q), _p(v), _p(o))
    return o


# --------------------------------- Sim3 ---------------------------------------
def _f64(a):
    return np.ascontiguousarray(a, np.float64)


def sim3_exp(a):
    a = _f64(a); o = np.zeros(7); lib().orc_sim3_exp(_p(a), _p(o)); return o


def sim3_log(S):
    S = _f64(S); o = np.zeros(7); lib().orc_sim3_log(_p(S), _p(o)); return o


def sim3_inverse(S):
    S = _f64(S); o = np.zeros(7); lib().orc_sim3_inverse(_p(S), _p(o)); return o


def sim3_plus(x, d):
    x = _f64(x); d = _f64(d); o = np.zeros(7); lib().orc_sim3_plus(_p(x), _p(d), _p(o)); return o


def sim3_act(S, p):
    S = _f64(S); p = _f64(p); o = np.zeros(3); lib().orc_sim3_act(_p(S), _p(p), _p(o)); return o


def sim3_eval_term(K4, lie7, P, uv, w, inverse):
    K4 = _f64(K4); lie7 = _f64(lie7); P = _f64(P); uv = _f64(uv)
    r = np.zeros(2); J = np.zeros((2, 7))
    lib().orc_sim3_eval_term(_p(K4), _p(lie7), _p(P), _p(uv), float(w), int(inverse), _p(r), _p(J))
    return r, J


def optimize_sim3(K1, K2, s12, P3D2c, obs1, inv_sigma2_1, P3D1c, obs2, inv_sigma2_2, th2=10.0, fix_scale=False):
    K1 = _f64(K1); K2 = _f64(K2); S = _f64(s12).copy()
    P2 = _f64(P3D2c).reshape(-1, 3); o1 = _f64(obs1).reshape(-1, 2); w1 = np.ascontiguousarray(inv_sigma2_1, np.float32)
    P1 = _f64(P3D1c).reshape(-1, 3); o2 = _f64(obs2).reshape(-1, 2); w2 = np.ascontiguousarray(inv_sigma2_2, np.float32)
    n = len(P2)
    out = np.zeros(max(n, 1), np.uint8)
    s = BaSummary()
    ninl = lib().orc_optimize_sim3(_p(K1), _p(K2), _p(S), _p(P2), _p(o1), _p(w1), _p(P1), _p(o2), _p(w2), n, float(th2),
                                   int(bool(fix_scale)), _p(out), C.byref(s))
    return ninl, S, out[:n], s.as_dict()


# --------------------------------- frame grid / frustum / undistort ----------------
def assign_features_to_grid(kps4, bounds):
    k = np.ascontiguousarray(kps4, np.float32).reshape(-1, 4); b = np.ascontiguousarray(bounds, np.float32)
    off = np.zeros(64 * 48 + 1, np.uint32); idx = np.zeros(max(len(k), 1), np.uint32)
    L = lib(); L.orc_assign_features_to_grid.restype = C.c_int
    L.orc_assign_features_to_grid.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    n = L.orc_assign_features_to_grid(_p(k), len(k), _p(b), _p(off), _p(idx))
    return off, idx[:n]


def undistort_keypoints(xy, K4, dist5):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2); out = np.zeros_like(xy)
    K4 = np.ascontiguousarray(K4, np.float32); d = np.ascontiguousarray(dist5, np.float32)
    L = lib(); L.orc_undistort_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_undistort_keypoints(_p(xy), len(xy), _p(K4), _p(d), _p(out))
    return out


def is_in_frustum(Rcw, tcw, K4, bounds, P, Pn, min_dist, max_dist, cos_limit, log_scale, nlevels):
    R = _f64(Rcw).reshape(9); t = _f64(tcw); K4 = np.ascontiguousarray(K4, np.float32); b = np.ascontiguousarray(bounds, np.float32)
    P = _f64(P).reshape(-1, 3); Pn = _f64(Pn).reshape(-1, 3)
    mn = np.ascontiguousarray(min_dist, np.float32); mx = np.ascontiguousarray(max_dist, np.float32)
    n = len(P)
    iv = np.zeros(n, np.uint8); uv = np.zeros((n, 2), np.float32); lv = np.zeros(n, np.int32); vc = np.zeros(n, np.float32)
    L = lib()
    L.orc_is_in_frustum.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_float, C.c_float, C.c_int] + [C.c_void_p] * 4
    L.orc_is_in_frustum(_p(R), _p(t), _p(K4), _p(b), _p(P), _p(Pn), _p(mn), _p(mx), n, float(cos_limit), float(log_scale), int(nlevels),
                        _p(iv), _p(uv), _p(lv), _p(vc))
    return iv, uv, lv, vc


# --------------------------------- BoW ---------------------------------------------
def bow_transform(voc, desc, levelsup=4):
    """voc = dict(node_desc, child_off, children, word_id, weight, L).  Returns (bow_word, bow_value, fv_node, fv_off, fv_idx)."""
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); n = len(d)
    bw = np.zeros(max(n, 1), np.uint32); bv = np.zeros(max(n, 1), np.float64); nw = C.c_int(0)
    fn = np.zeros(max(n, 1), np.uint32); fo = np.zeros(n + 2, np.uint32); fi = np.zeros(max(n, 1), np.uint32); nf = C.c_int(0)
    L = lib()
    L.orc_bow_transform.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 7
    L.orc_bow_transform(_p(voc["node_desc"]), _p(voc["child_off"]), _p(voc["children"]), _p(voc["word_id"]), _p(voc["weight"]),
                        len(voc["word_id"]), int(voc["L"]), int(levelsup), _p(d), n, _p(bw), _p(bv), C.byref(nw), _p(fn), _p(fo), _p(fi),
                        C.byref(nf))
    return bw[:nw.value], bv[:nw.value], fn[:nf.value], fo[:nf.value + 1], fi[:fo[nf.value]]


def bow_descend(voc, desc, levelsup=4):
    """Per-feature descent only: (word id, idf weight, node id at level L - levelsup) of every descriptor."""
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); n = len(d)
    wid = np.zeros(n, np.int32); w = np.zeros(n, np.float64); nid = np.zeros(n, np.uint32)
    L = lib()
    L.orc_bow_descend.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int] + [C.c_void_p] * 4
    for i in range(n):
        L.orc_bow_descend(_p(voc["node_desc"]), _p(voc["child_off"]), _p(voc["children"]), _p(voc["word_id"]), _p(voc["weight"]),
                          int(voc["L"]), int(levelsup), d[i].ctypes.data, wid[i:].ctypes.data, w[i:].ctypes.data, nid[i:].ctypes.data)
    return wid, w, nid


def _merge_call(fn, wid, w, nid, *extra):
    wid = np.ascontiguousarray(wid, np.int32); w = np.ascontiguousarray(w, np.float64); nid = np.ascontiguousarray(nid, np.uint32)
    n = len(wid)
    bw = np.zeros(max(n, 1), np.uint32); bv = np.zeros(max(n, 1), np.float64)
    fn_ = np.zeros(max(n, 1), np.uint32); fo = np.zeros(n + 2, np.uint32); fi = np.zeros(max(n, 1), np.uint32); nf = C.c_int(0)
    return n, wid, w, nid, bw, bv, fn_, fo, fi, nf


def bow_merge(wid, w, nid):
    """The oracle's merge of per-feature triples (restatement of BowVector / FeatureVector); same outputs as bow_transform."""
    n, wid, w, nid, bw, bv, fn_, fo, fi, nf = _merge_call(None, wid, w, nid)
    nw = C.c_int(0)
    L = lib()
    L.orc_bow_merge.argtypes = [C.c_int] + [C.c_void_p] * 10
    L.orc_bow_merge(n, _p(wid), _p(w), _p(nid), _p(bw), _p(bv), C.byref(nw), _p(fn_), _p(fo), _p(fi), C.byref(nf))
    return bw[:nw.value], bv[:nw.value], fn_[:nf.value], fo[:nf.value + 1], fi[:fo[nf.value]]


# --------------------------------- oracle/_ref: the REFERENCE's own DBoW2 classes ------------------------------------
_REF_SO = os.path.join(_HERE, "_ref", "libdbow2_ref.so")
_REFERENCE_ROOT = os.environ.get("ORB_REFERENCE_ROOT", "/root/reference")


def build_ref(force=False):
    """Compile the reference's BowVector.cpp / FeatureVector.cpp (where they lie) + oracle/ref_dbow2_shim.cpp into
    oracle/_ref/libdbow2_ref.so.  Returns the path, or None when the reference tree is absent (GPU box: the prebuilt file travels)."""
    src = os.path.join(_REFERENCE_ROOT, "lib", "DBoW2", "DBoW2", "BowVector.cpp")
    if not os.path.exists(src):
        return _REF_SO if os.path.exists(_REF_SO) else None
    if force or not os.path.exists(_REF_SO) or os.path.getmtime(os.path.join(_HERE, "ref_dbow2_shim.cpp")) > os.path.getmtime(_REF_SO):
        subprocess.check_call(["make", "-C", _HERE, "ref", "REF=" + _REFERENCE_ROOT] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return _REF_SO


_ref = None


def ref_lib():
    """The built oracle/_ref library, or None if it was never built (no reference tree and no prebuilt file)."""
    global _ref
    if _ref is None and os.path.exists(_REF_SO):
        _ref = C.CDLL(_REF_SO)
        _ref.ref_bow_merge.restype = C.c_int
        _ref.ref_bow_merge.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 6
        _ref.ref_bow_merge_if_not_exist.restype = C.c_int
        _ref.ref_bow_merge_if_not_exist.argtypes = [C.c_int] + [C.c_void_p] * 4
    return _ref


def ref_bow_merge(wid, w, nid, norm=0):
    """The same merge done by the reference's BowVector::addWeight / normalize and FeatureVector::addFeature."""
    n, wid, w, nid, bw, bv, fn_, fo, fi, nf = _merge_call(None, wid, w, nid)
    k = ref_lib().ref_bow_merge(n, _p(wid), _p(w), _p(nid), int(norm), _p(bw), _p(bv), _p(fn_), _p(fo), _p(fi), C.byref(nf))
    return bw[:k], bv[:k], fn_[:nf.value], fo[:nf.value + 1], fi[:fo[nf.value]]


def bow_score_l1(w1, v1, w2, v2):
    L = lib(); L.orc_bow_score_l1.restype = C.c_double
    L.orc_bow_score_l1.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    w1 = np.ascontiguousarray(w1, np.uint32); w2 = np.ascontiguousarray(w2, np.uint32); v1 = _f64(v1); v2 = _f64(v2)
    return L.orc_bow_score_l1(_p(w1), _p(v1), len(w1), _p(w2), _p(v2), len(w2))


# --------------------------------- triangulation -----------------------------------
def null_vector4(A):
    A = _f64(A).reshape(16); x = np.zeros(4)
    L = lib(); L.orc_null_vector4.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_null_vector4(_p(A), _p(x))
    return x


def triangulate_matches(T1, T2, K1, K2, kp1, kp2, level_sigma2, scale_factors, ratio_factor):
    T1 = _f64(T1).reshape(12); T2 = _f64(T2).reshape(12)
    K1 = np.ascontiguousarray(K1, np.float32); K2 = np.ascontiguousarray(K2, np.float32)
    kp1 = np.ascontiguousarray(kp1, np.float32).reshape(-1, 3); kp2 = np.ascontiguousarray(kp2, np.float32).reshape(-1, 3)
    ls = np.ascontiguousarray(level_sigma2, np.float32); sf = np.ascontiguousarray(scale_factors, np.float32)
    n = len(kp1); X = np.zeros((n, 3)); ok = np.zeros(n, np.uint8)
    L = lib(); L.orc_triangulate_matches.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    L.orc_triangulate_matches(_p(T1), _p(T2), _p(K1), _p(K2), _p(kp1), _p(kp2), n, _p(ls), _p(sf), float(ratio_factor), _p(X), _p(ok))
    return X, ok


# --------------------------------- essential graph ---------------------------------
def sim3_adj(S):
    S = _f64(S); A = np.zeros((7, 7)); L = lib(); L.orc_sim3_adj.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_sim3_adj(_p(S), _p(A)); return A


def sim3_mul(a, b):
    a = _f64(a); b = _f64(b); o = np.zeros(7); L = lib(); L.orc_sim3_mul.argtypes = [C.c_void_p] * 3
    L.orc_sim3_mul(_p(a), _p(b), _p(o)); return o


def eg_eval_edge(lie_j, lie_i, Sji):
    lj = _f64(lie_j); li = _f64(lie_i); S = _f64(Sji); r = np.zeros(7); J = np.zeros((7, 7))
    L = lib(); L.orc_eg_eval_edge.argtypes = [C.c_void_p] * 5
    L.orc_eg_eval_edge(_p(lj), _p(li), _p(S), _p(r), _p(J)); return r, J


def optimize_essential_graph(lie7, kf_fixed, edge_j, edge_i, edge_Sji, max_iters=100):
    x = _f64(lie7).reshape(-1, 7).copy(); fx = np.ascontiguousarray(kf_fixed, np.uint8)
    ej = np.ascontiguousarray(edge_j, np.int32); ei = np.ascontiguousarray(edge_i, np.int32); S = _f64(edge_Sji).reshape(-1, 7)
    s = BaSummary(); L = lib()
    L.orc_optimize_essential_graph.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.orc_optimize_essential_graph(_p(x), _p(fx), len(x), _p(ej), _p(ei), _p(S), len(ej), int(max_iters), C.byref(s))
    return x, s.as_dict()


def essential_graph_correct(lie_orig, lie_opt, pt_ref, pts):
    a = _f64(lie_orig).reshape(-1, 7); b = _f64(lie_opt).reshape(-1, 7); n = len(a)
    T = np.zeros((n, 12)); pr = np.ascontiguousarray(pt_ref, np.int32); P = _f64(pts).reshape(-1, 3).copy()
    L = lib(); L.orc_essential_graph_correct.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_essential_graph_correct(_p(a), _p(b), n, _p(T), _p(pr), _p(P), len(P))
    return T, P
