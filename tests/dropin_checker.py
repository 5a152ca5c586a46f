"""Checker of the reference-signature drop-in classes (csrc/compat/orbslam_dropin.h).  TEST INFRASTRUCTURE.

tests/cpp/test_dropin.cpp runs every entry point through the HIP library on a mock map and writes the complete map state
before and after the call (tests/cpp/scene_io.h).  This module loads the "before" state into a small Python data model and
REPLAYS the entry point on it: a sequential restatement of what the reference's function does to the map - which points are
considered, the projection geometry in the reference's float / double mix, the greedy bookkeeping, the mutations - written in
Python from the reference lines cited at each function, with the frame grid of tests/npmatch.py and, for the numeric solves,
the CPU oracle's flat functions (oracle/pyoracle.py).  The result must equal the "after" state field by field.

It shares no text with the drop-in header (C++ templates over the C ABI) nor with the reference (C++ over OpenCV / Eigen /
Ceres): float32 steps are explicit np.float32 operations, doubles are Python floats in the reference's left-to-right order.
"""
import ctypes
import math

import numpy as np

from tests.npmatch import Grid, rot_bin, three_maxima

F32 = np.float32
TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30                   # src/ORBmatcher.cc:35-37
_POP = np.array([bin(i).count("1") for i in range(256)], np.int32)
_libm = ctypes.CDLL("libm.so.6")
_libm.logf.restype = ctypes.c_float
_libm.logf.argtypes = [ctypes.c_float]
NEVER = (1 << 64) - 1                                          # ~0ul of the bookkeeping fields, as dumped (int64 -1)


# ------------------------------------------------------------------------------------------------ dump reader
_DT = {ord("b"): np.uint8, ord("i"): np.int32, ord("q"): np.int64, ord("f"): np.float32, ord("d"): np.float64}


def load_records(path):
    buf = open(path, "rb").read()
    off, out = 0, {}
    while off < len(buf):
        nl = int(np.frombuffer(buf, np.uint32, 1, off)[0]); off += 4
        name = buf[off:off + nl].decode(); off += nl
        dt = _DT[buf[off]]; off += 1
        n = int(np.frombuffer(buf, np.uint64, 1, off)[0]); off += 8
        out[name] = np.frombuffer(buf, dt, n, off).copy(); off += n * np.dtype(dt).itemsize
    return out


# ------------------------------------------------------------------------------------------------ data model
class Owner:
    """The members Frame and KeyFrame share (include/Frame.h, include/KeyFrame.h)."""

    def __init__(self, R, p, scene):
        self.scene = scene
        self.kpu = R[p + ".kpu"].reshape(-1, 4).copy()         # undistort_keypoints_: x, y, octave, angle
        self.kp = R[p + ".kp"].reshape(-1, 4).copy()           # keypoints_
        self.N = len(self.kpu)
        self.desc = R[p + ".desc"].reshape(-1, 32).copy()
        self.mp = [int(v) for v in R[p + ".mp"]]               # map_points_ as indices, -1 = nullptr
        self.owner = [int(v) for v in R[p + ".owner"]]
        node, off, idx = R[p + ".fv_node"], R[p + ".fv_off"], R[p + ".fv_idx"]
        self.fv = {int(node[m]): [int(v) for v in idx[off[m]:off[m + 1]]] for m in range(len(node))}
        self.bow = {int(w): float(v) for w, v in zip(R[p + ".bow_word"], R[p + ".bow_value"])}
        self.T = R[p + ".T"].reshape(4, 4).copy()
        self._grid = None

    def grid(self):
        if self._grid is None:
            self._grid = Grid(self.kpu, self.bounds)
        return self._grid

    def octave(self, i):
        return int(self.kpu[i, 2])


class KeyFrame(Owner):
    def __init__(self, R, p, scene):
        Owner.__init__(self, R, p, scene)
        self.K = R[p + ".K"].copy(); self.bounds = R[p + ".bounds"].copy()
        m = R[p + ".meta"]
        self.id, self.bad = int(m[0]), bool(m[1])
        self.ba_local, self.ba_fixed, self.ba_global = int(m[2]) % (1 << 64), int(m[3]) % (1 << 64), int(m[4]) % (1 << 64)
        self.n_set_pose, self.parent = int(m[5]), int(m[6])
        self.gba_T = R[p + ".gba_T"].reshape(4, 4).copy()
        self.conn = [int(v) for v in R[p + ".conn"]]; self.conn_w = [int(v) for v in R[p + ".conn_w"]]
        self.children = set(int(v) for v in R[p + ".children"]); self.loop_edges = sorted(int(v) for v in R[p + ".loop_edges"])
        self.weights = {int(k): int(v) for k, v in zip(R[p + ".w_kf"], R[p + ".w_val"])}

    def set_pose(self, T):
        self.T = np.array(T, np.float64); self.n_set_pose += 1

    def in_image(self, u, v):                                  # KeyFrame::IsInImage (src/KeyFrame.cc:624)
        b = self.bounds
        return u >= b[0] and u < b[1] and v >= b[2] and v < b[3]


class Frame(Owner):
    def __init__(self, R, p, scene):
        Owner.__init__(self, R, p, scene)
        self.K = scene.frame_K; self.bounds = scene.frame_bounds
        self.outl = [bool(v) for v in R[p + ".outl"]]
        self.n_set_pose = int(R[p + ".n_set_pose"][0])


class Scene:
    def __init__(self, R, p):
        self.scale = R[p + ".scale"].copy(); self.sigma2 = R[p + ".sigma2"].copy(); self.inv_sigma2 = R[p + ".inv_sigma2"].copy()
        self.frame_K = R[p + ".frame_K"].copy(); self.frame_bounds = R[p + ".frame_bounds"].copy()
        self.n_levels = 8
        self.log_scale = F32(_libm.logf(F32(1.2)))             # log_scale_factor_ = log(scale_factor_) on floats
        self.kfs = [KeyFrame(R, "%s.kf%d" % (p, k), self) for k in range(int(R[p + ".n_kf"][0]))]
        self.frames = [Frame(R, "%s.fr%d" % (p, k), self) for k in range(int(R[p + ".n_frames"][0]))]
        q = p + ".mp"
        self.pos = R[q + ".pos"].reshape(-1, 3).copy(); self.normal = R[q + ".normal"].reshape(-1, 3).copy()
        self.gba_pos = R[q + ".gba_pos"].reshape(-1, 3).copy(); self.mp_desc = R[q + ".desc"].reshape(-1, 32).copy()
        m = R[q + ".meta"].reshape(-1, 12)
        n = len(m)
        self.mp_id = [int(v) for v in m[:, 0]]; self.mp_bad = [bool(v) for v in m[:, 1]]; self.replaced = [int(v) for v in m[:, 2]]
        self.nobs = [int(v) for v in m[:, 3]]; self.mp_ba_local = [int(v) % (1 << 64) for v in m[:, 4]]; self.mp_ba_global = [int(v) % (1 << 64) for v in m[:, 5]]
        self.n_update_normal = [int(v) for v in m[:, 6]]; self.in_view = [bool(v) for v in m[:, 7]]; self.track_level = [int(v) for v in m[:, 8]]
        self.corrected_by = [int(v) % (1 << 64) for v in m[:, 9]]; self.corrected_ref = [int(v) for v in m[:, 10]]; self.ref_kf = [int(v) for v in m[:, 11]]
        fl = R[q + ".fl"].reshape(-1, 5)
        self.min_dist = fl[:, 0].copy(); self.max_dist = fl[:, 1].copy()
        self.track_u = fl[:, 2].copy(); self.track_v = fl[:, 3].copy(); self.track_cos = fl[:, 4].copy()
        off, okf, oidx = R[q + ".obs_off"], R[q + ".obs_kf"], R[q + ".obs_idx"]
        self.obs = [{int(okf[e]): int(oidx[e]) for e in range(off[i], off[i + 1])} for i in range(n)]
        self.map_kfs = [int(v) for v in R[p + ".map_kfs"]]; self.map_mps = [int(v) for v in R[p + ".map_mps"]]

    # ---- MapPoint members (src/MapPoint.cc) on index p
    def min_inv(self, p):
        return F32(0.8) * self.min_dist[p]                     # GetMinDistanceInvariance (:379-382)

    def max_inv(self, p):
        return F32(1.2) * self.max_dist[p]                     # GetMaxDistanceInvariance (:384-388)

    def predict_scale(self, p, dist):                          # MapPoint::PredictScale (:390-420): float ratio, float log, ceil, clamp
        ratio = F32(self.max_dist[p] / F32(dist))
        ns = int(math.ceil(float(F32(F32(_libm.logf(ratio)) / self.log_scale))))
        return 0 if ns < 0 else min(ns, self.n_levels - 1)

    def add_observation(self, p, kf, idx):                     # AddObservation (:66-77, monocular)
        if kf in self.obs[p]:
            return
        self.obs[p][kf] = idx; self.nobs[p] += 1

    def erase_observation(self, p, kf):                        # EraseObservation (:79-105) as far as the mock carries it
        if kf in self.obs[p]:
            del self.obs[p][kf]; self.nobs[p] -= 1

    def replace(self, p, q):                                   # MapPoint::Replace (:185-222)
        if self.mp_id[q] == self.mp_id[p]:
            return
        old = dict(self.obs[p])
        self.obs[p] = {}; self.mp_bad[p] = True; self.replaced[p] = q
        for kf in sorted(old):
            idx = old[kf]
            if kf not in self.obs[q]:
                self.kfs[kf].mp[idx] = q; self.add_observation(q, kf, idx)
            else:
                self.kfs[kf].mp[idx] = -1
        self.compute_distinctive_descriptors(q)                # (:230)

    def compute_distinctive_descriptors(self, p):              # MapPoint::ComputeDistinctiveDescriptors (:256-315)
        if self.mp_bad[p] or not self.obs[p]:
            return
        ds = [self.kfs[kf].desc[idx] for kf, idx in sorted(self.obs[p].items()) if not self.kfs[kf].bad]
        if not ds:
            return
        N = len(ds)
        D = [[int(np.unpackbits(ds[i] ^ ds[j]).sum()) for j in range(N)] for i in range(N)]
        best_median, best = 1 << 30, 0
        for i in range(N):
            median = sorted(D[i])[int(0.5 * (N - 1))]
            if median < best_median:
                best_median, best = median, i
        self.mp_desc[p] = ds[best]


def _rt(T):
    return [[float(T[r, c]) for c in range(3)] for r in range(3)], [float(T[r, 3]) for r in range(3)]


def _mul(R, p):                                                # Eigen Matrix3d * Vector3d: plain left-to-right sums in double
    return [R[r][0] * p[0] + R[r][1] * p[1] + R[r][2] * p[2] for r in range(3)]


def _mul_t(R, p):
    return [R[0][c] * p[0] + R[1][c] * p[1] + R[2][c] * p[2] for c in range(3)]


def _norm(p):
    return math.sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2])


def _dists(descs, d):
    return _POP[np.bitwise_xor(descs, d[None, :])].sum(1)


def _prune_rotation(hist):
    """ComputeThreeMaxima (src/ORBmatcher.cc:1386-1418): the bins that are NOT kept."""
    i1, i2, i3 = three_maxima([len(h) for h in hist])
    return [b for b in range(HISTO_LENGTH) if b not in (i1, i2, i3)]


# ------------------------------------------------------------------------------------------------ ORBmatcher
def is_in_frustum(S, F, p, cos_limit):
    """Frame::isInFrustum (src/Frame.cc:191-241): float camera coordinates, double PO, float dist / viewCos."""
    S.in_view[p] = False
    R, t = _rt(F.T)
    P = [float(v) for v in S.pos[p]]
    Pc = _mul(R, P)
    PcX, PcY, PcZ = F32(Pc[0] + t[0]), F32(Pc[1] + t[1]), F32(Pc[2] + t[2])
    if PcZ < F32(0):
        return False
    invz = F32(1.0) / PcZ
    fx, fy, cx, cy = [F32(v) for v in F.K]
    u = F32(F32(fx * PcX) * invz) + cx
    v = F32(F32(fy * PcY) * invz) + cy
    b = F.bounds
    if u < b[0] or u > b[1] or v < b[2] or v > b[3]:
        return False
    Ow = [-x for x in _mul_t(R, t)]
    PO = [P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]]
    dist = F32(_norm(PO))
    if dist < S.min_inv(p) or dist > S.max_inv(p):
        return False
    Pn = [float(x) for x in S.normal[p]]
    view_cos = F32((PO[0] * Pn[0] + PO[1] * Pn[1] + PO[2] * Pn[2]) / float(dist))
    if view_cos < F32(cos_limit):
        return False
    S.in_view[p] = True; S.track_u[p] = u; S.track_v[p] = v; S.track_level[p] = S.predict_scale(p, dist); S.track_cos[p] = view_cos
    return True


def search_by_projection_points(S, F, points, th, nnratio):
    """ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th) (src/ORBmatcher.cc:42-119)."""
    n = 0
    g = F.grid()
    for p in points:
        if not S.in_view[p] or S.mp_bad[p]:
            continue
        lvl = S.track_level[p]
        r = F32(2.5) if S.track_cos[p] > 0.998 else F32(4.0)   # RadiusByViewingCos (:121-126): float compared with a double literal
        if th != 1.0:
            r = F32(r * F32(th))
        cand = g.features_in_area(S.track_u[p], S.track_v[p], F32(r * S.scale[lvl]), lvl - 1, lvl)
        best, best_lvl, second, second_lvl, best_idx = 256, -1, 256, -1, -1
        d = _dists(F.desc[cand], S.mp_desc[p]) if cand else []
        for k, idx in enumerate(cand):
            if F.mp[idx] >= 0 and S.nobs[F.mp[idx]] > 0:
                continue
            dist = int(d[k])
            if dist < best:
                second, second_lvl = best, best_lvl
                best, best_lvl, best_idx = dist, F.octave(idx), idx
            elif dist < second:
                second, second_lvl = dist, F.octave(idx)
        if best <= TH_HIGH:
            if best_lvl == second_lvl and best > F32(nnratio) * F32(second):
                continue
            F.mp[best_idx] = p
            n += 1
    return n


def _project_frame(F, P):
    """x3Dc = Rcw x3Dw + tcw in double; xc, yc, 1/zc narrowed to float; u, v in float (src/ORBmatcher.cc:1190-1201, :1300-1310)."""
    R, t = _rt(F.T)
    c = _mul(R, P)
    xc, yc = F32(c[0] + t[0]), F32(c[1] + t[1])
    zc = c[2] + t[2]
    invz = F32(1.0 / zc) if zc != 0.0 else F32(np.inf)
    fx, fy, cx, cy = [F32(v) for v in F.K]
    u = F32(F32(fx * xc) * invz) + cx
    v = F32(F32(fy * yc) * invz) + cy
    return u, v, invz


def search_by_projection_last_frame(S, cur, last, th, check_ori):
    """ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th) (src/ORBmatcher.cc:1161-1271, monocular)."""
    n = 0
    hist = [[] for _ in range(HISTO_LENGTH)]
    g = cur.grid()
    b = cur.bounds
    for i in range(last.N):
        p = last.mp[i]
        if p < 0 or last.outl[i]:
            continue
        u, v, invz = _project_frame(cur, [float(x) for x in S.pos[p]])
        if invz < 0:
            continue
        if u < b[0] or u > b[1] or v < b[2] or v > b[3]:
            continue
        octv = int(last.kp[i, 2])
        cand = g.features_in_area(u, v, F32(F32(th) * S.scale[octv]), octv - 1, octv + 1)
        if not cand:
            continue
        d = _dists(cur.desc[cand], S.mp_desc[p])
        best, best_idx = 256, -1
        for k, i2 in enumerate(cand):
            if cur.mp[i2] >= 0 and S.nobs[cur.mp[i2]] > 0:
                continue
            if int(d[k]) < best:
                best, best_idx = int(d[k]), i2
        if best <= TH_HIGH:
            cur.mp[best_idx] = p
            n += 1
            if check_ori:
                hist[rot_bin(last.kpu[i, 3], cur.kpu[best_idx, 3])].append(best_idx)
    if check_ori:
        for bn in _prune_rotation(hist):
            for idx in hist[bn]:
                cur.mp[idx] = -1
                n -= 1
    return n


def search_by_projection_keyframe(S, cur, kf, found, th, orb_dist, check_ori):
    """ORBmatcher::SearchByProjection(Frame&, KeyFrame*, set<MapPoint*>&, th, ORBdist) (src/ORBmatcher.cc:1273-1384)."""
    n = 0
    R, t = _rt(cur.T)
    Ow = [-x for x in _mul_t(R, t)]
    hist = [[] for _ in range(HISTO_LENGTH)]
    g = cur.grid()
    b = cur.bounds
    for i, p in enumerate(kf.mp):
        if p < 0 or S.mp_bad[p] or p in found:
            continue
        P = [float(x) for x in S.pos[p]]
        u, v, _ = _project_frame(cur, P)
        if u < b[0] or u > b[1] or v < b[2] or v > b[3]:
            continue
        dist3d = F32(_norm([P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]]))
        if dist3d < S.min_inv(p) or dist3d > S.max_inv(p):
            continue
        lvl = S.predict_scale(p, dist3d)
        cand = g.features_in_area(u, v, F32(F32(th) * S.scale[lvl]), lvl - 1, lvl + 1)
        if not cand:
            continue
        d = _dists(cur.desc[cand], S.mp_desc[p])
        best, best_idx = 256, -1
        for k, i2 in enumerate(cand):
            if cur.mp[i2] >= 0:
                continue
            if int(d[k]) < best:
                best, best_idx = int(d[k]), i2
        if best <= orb_dist:
            cur.mp[best_idx] = p
            n += 1
            if check_ori:
                hist[rot_bin(kf.kpu[i, 3], cur.kpu[best_idx, 3])].append(best_idx)
    if check_ori:
        for bn in _prune_rotation(hist):
            for idx in hist[bn]:
                cur.mp[idx] = -1
                n -= 1
    return n


def _decompose_sim3(Scw):
    """"Decompose Scw" (src/ORBmatcher.cc:269-274, :854-859): float scale from the first row, Rcw, tcw, Ow in double."""
    sR = [[float(Scw[r, c]) for c in range(3)] for r in range(3)]
    scw = F32(math.sqrt(sR[0][0] * sR[0][0] + sR[0][1] * sR[0][1] + sR[0][2] * sR[0][2]))
    s = float(scw)
    R = [[sR[r][c] / s for c in range(3)] for r in range(3)]
    t = [float(Scw[r, 3]) / s for r in range(3)]
    Ow = [-x for x in _mul_t(R, t)]
    return R, t, Ow


def _project_gates(S, kf, R, t, Ow, p, viewing=True):
    """The gates the Sim3 projection and both Fuse overloads share (src/ORBmatcher.cc:286-322, :746-777, :873-906)."""
    P = [float(x) for x in S.pos[p]]
    c = _mul(R, P)
    c = [c[0] + t[0], c[1] + t[1], c[2] + t[2]]
    if c[2] < 0.0:
        return None
    invz = F32(1.0 / c[2])
    x, y = F32(c[0] * float(invz)), F32(c[1] * float(invz))
    fx, fy, cx, cy = [F32(v) for v in kf.K]
    u, v = F32(fx * x) + cx, F32(fy * y) + cy
    if not kf.in_image(u, v):
        return None
    PO = [P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]]
    dist = F32(_norm(PO))
    if dist < S.min_inv(p) or dist > S.max_inv(p):
        return None
    if viewing:
        Pn = [float(x) for x in S.normal[p]]
        if PO[0] * Pn[0] + PO[1] * Pn[1] + PO[2] * Pn[2] < 0.5 * float(dist):
            return None
    return u, v, S.predict_scale(p, dist)


def search_by_projection_sim3(S, kf, Scw, points, matched, th):
    """ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (src/ORBmatcher.cc:258-361)."""
    R, t, Ow = _decompose_sim3(Scw)
    found = set(m for m in matched if m >= 0)
    n = 0
    g = kf.grid()
    for p in points:
        if S.mp_bad[p] or p in found:
            continue
        pr = _project_gates(S, kf, R, t, Ow, p)
        if pr is None:
            continue
        u, v, lvl = pr
        cand = g.features_in_area(u, v, F32(F32(th) * S.scale[lvl]))
        if not cand:
            continue
        d = _dists(kf.desc[cand], S.mp_desc[p])
        best, best_idx = 256, -1
        for k, idx in enumerate(cand):
            if matched[idx] >= 0:
                continue
            if kf.octave(idx) < lvl - 1 or kf.octave(idx) > lvl:
                continue
            if int(d[k]) < best:
                best, best_idx = int(d[k]), idx
        if best <= TH_LOW:
            matched[best_idx] = p
            n += 1
    return n


def _walk_nodes(fa, fb):
    """The merge walk over two DBoW2 feature vectors (std::map order, lower_bound jumps): the common nodes, ascending."""
    return [(fa[k], fb[k]) for k in sorted(fa) if k in fb]


def search_by_bow_frame(S, kf, F, nnratio, check_ori):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (src/ORBmatcher.cc:151-256)."""
    out = [-1] * F.N
    n = 0
    hist = [[] for _ in range(HISTO_LENGTH)]
    for la, lb in _walk_nodes(kf.fv, F.fv):
        for ia in la:
            p = kf.mp[ia]
            if p < 0 or S.mp_bad[p]:
                continue
            d = _dists(F.desc[lb], kf.desc[ia])
            b1, b2, bi = 256, 256, -1
            for k, ib in enumerate(lb):
                if out[ib] >= 0:
                    continue
                dist = int(d[k])
                if dist < b1:
                    b2 = b1; b1 = dist; bi = ib
                elif dist < b2:
                    b2 = dist
            if b1 <= TH_LOW and F32(b1) < F32(nnratio) * F32(b2):
                out[bi] = p
                if check_ori:
                    hist[rot_bin(kf.kpu[ia, 3], F.kp[bi, 3])].append(bi)       # (:221: the frame's RAW keypoint angle)
                n += 1
    if check_ori:
        for bn in _prune_rotation(hist):
            for idx in hist[bn]:
                out[idx] = -1
                n -= 1
    return n, out


def search_by_bow_keyframes(S, k1, k2, nnratio, check_ori):
    """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (src/ORBmatcher.cc:470-580)."""
    out = [-1] * len(k1.mp)
    used2 = [False] * len(k2.mp)
    n = 0
    hist = [[] for _ in range(HISTO_LENGTH)]
    for la, lb in _walk_nodes(k1.fv, k2.fv):
        for ia in la:
            p1 = k1.mp[ia]
            if p1 < 0 or S.mp_bad[p1]:
                continue
            d = _dists(k2.desc[lb], k1.desc[ia])
            b1, b2, bi = 256, 256, -1
            for k, ib in enumerate(lb):
                p2 = k2.mp[ib]
                if used2[ib] or p2 < 0 or S.mp_bad[p2]:
                    continue
                dist = int(d[k])
                if dist < b1:
                    b2 = b1; b1 = dist; bi = ib
                elif dist < b2:
                    b2 = dist
            if b1 < TH_LOW and F32(b1) < F32(nnratio) * F32(b2):
                out[ia] = k2.mp[bi]; used2[bi] = True
                if check_ori:
                    hist[rot_bin(k1.kpu[ia, 3], k2.kpu[bi, 3])].append(ia)
                n += 1
    if check_ori:
        for bn in _prune_rotation(hist):
            for idx in hist[bn]:
                out[idx] = -1
                n -= 1
    return n, out


def search_for_initialization(S, F1, F2, prev, window, nnratio, check_ori):
    """ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:363-468); prev = vbPrevMatched as an n x 2 float array (updated)."""
    n = 0
    m12 = [-1] * F1.N
    m21 = [-1] * F2.N
    claimed = [1 << 31] * F2.N
    hist = [[] for _ in range(HISTO_LENGTH)]
    g = F2.grid()
    for i1 in range(F1.N):
        lvl = F1.octave(i1)
        if lvl > 0:
            continue
        cand = g.features_in_area(prev[i1, 0], prev[i1, 1], F32(window), lvl, lvl)
        if not cand:
            continue
        d = _dists(F2.desc[cand], F1.desc[i1])
        b1, b2, bi = 1 << 31, 1 << 31, -1
        for k, i2 in enumerate(cand):
            dist = int(d[k])
            if claimed[i2] <= dist:
                continue
            if dist < b1:
                b2 = b1; b1 = dist; bi = i2
            elif dist < b2:
                b2 = dist
        if b1 <= TH_LOW and b1 < F32(F32(b2) * F32(nnratio)):
            if m21[bi] >= 0:
                m12[m21[bi]] = -1
                n -= 1
            m12[i1] = bi; m21[bi] = i1; claimed[bi] = b1
            n += 1
            if check_ori:
                hist[rot_bin(F1.kpu[i1, 3], F2.kpu[bi, 3])].append(i1)
    if check_ori:
        for bn in _prune_rotation(hist):
            for i1 in hist[bn]:
                if m12[i1] >= 0:
                    m12[i1] = -1
                    n -= 1
    for i1 in range(F1.N):
        if m12[i1] >= 0:
            prev[i1] = F2.kpu[m12[i1], :2]
    return n, m12


def _epipolar_ok(kp1, kp2, F12, sigma2):
    """ORBmatcher::CheckDistEpipolarLine (src/ORBmatcher.cc:128-149): the line in float from double products."""
    x1, y1 = float(kp1[0]), float(kp1[1])
    a = F32(x1 * F12[0][0] + y1 * F12[1][0] + F12[2][0])
    b = F32(x1 * F12[0][1] + y1 * F12[1][1] + F12[2][1])
    c = F32(x1 * F12[0][2] + y1 * F12[1][2] + F12[2][2])
    num = F32(F32(a * kp2[0]) + F32(b * kp2[1])) + c
    den = F32(a * a) + F32(b * b)
    if den == 0:
        return False
    dsqr = F32(F32(num * num) / den)
    return float(dsqr) < 3.84 * float(sigma2[int(kp2[2])])


def search_for_triangulation(S, k1, k2, F12, check_ori):
    """ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:582-722, monocular): pairs (idx1, idx2) in idx1 order."""
    R2, t2 = _rt(k2.T)
    R1, t1 = _rt(k1.T)
    Cw = [-x for x in _mul_t(R1, t1)]                          # GetCameraCenter
    C2 = _mul(R2, Cw)
    C2 = [C2[0] + t2[0], C2[1] + t2[1], C2[2] + t2[2]]
    invz = F32(1.0) / F32(C2[2])
    fx, fy, cx, cy = [F32(v) for v in k2.K]
    ex = F32(F32(float(fx) * C2[0]) * invz) + cx
    ey = F32(F32(float(fy) * C2[1]) * invz) + cy
    Fm = [[float(F12[r, c]) for c in range(3)] for r in range(3)]
    m12 = [-1] * k1.N
    n = 0
    hist = [[] for _ in range(HISTO_LENGTH)]
    for la, lb in _walk_nodes(k1.fv, k2.fv):
        for ia in la:
            if k1.mp[ia] >= 0:
                continue
            d = _dists(k2.desc[lb], k1.desc[ia])
            best, bi = TH_LOW, -1
            for k, ib in enumerate(lb):
                if k2.mp[ib] >= 0:                             # (vbMatched2 is never set in this fork)
                    continue
                dist = int(d[k])
                if dist > TH_LOW or dist > best:
                    continue
                kp2 = k2.kpu[ib]
                dx, dy = F32(ex - kp2[0]), F32(ey - kp2[1])
                if F32(dx * dx) + F32(dy * dy) < F32(100) * S.scale[int(kp2[2])]:
                    continue
                if _epipolar_ok(k1.kpu[ia], kp2, Fm, S.sigma2):
                    bi, best = ib, dist
            if bi >= 0:
                m12[ia] = bi
                n += 1
                if check_ori:
                    hist[rot_bin(k1.kpu[ia, 3], k2.kpu[bi, 3])].append(ia)
    if check_ori:
        for bn in _prune_rotation(hist):
            for ia in hist[bn]:
                m12[ia] = -1
                n -= 1
    return n, [(i, m12[i]) for i in range(k1.N) if m12[i] >= 0]


def search_by_sim3(S, k1, i1, k2, i2, m12, s12, R12, t12, th):
    """ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:956-1159); m12 = vpMatches12 as indices (updated in place)."""
    R1, t1 = _rt(k1.T)
    R2, t2 = _rt(k2.T)
    s = float(F32(s12))
    sR12 = [[s * float(R12[r, c]) for c in range(3)] for r in range(3)]
    sR21 = [[(1.0 / s) * float(R12[c, r]) for c in range(3)] for r in range(3)]
    t12 = [float(x) for x in t12]
    t21 = [-x for x in _mul(sR21, t12)]
    N1, N2 = len(k1.mp), len(k2.mp)
    done1, done2 = [False] * N1, [False] * N2
    for i in range(N1):
        if m12[i] >= 0:
            done1[i] = True
            j = S.obs[m12[i]].get(i2, -1)
            if 0 <= j < N2:
                done2[j] = True

    def one_way(src, done, Ra, ta, sRba, tba, dst):
        out = [-1] * len(src.mp)
        g = dst.grid()
        fx, fy, cx, cy = [F32(v) for v in k1.K]                # (both directions use keyframe 1's intrinsics, :958-961)
        for i, p in enumerate(src.mp):
            if p < 0 or done[i] or S.mp_bad[p]:
                continue
            ca = _mul(Ra, [float(x) for x in S.pos[p]])
            ca = [ca[0] + ta[0], ca[1] + ta[1], ca[2] + ta[2]]
            cb = _mul(sRba, ca)
            cb = [cb[0] + tba[0], cb[1] + tba[1], cb[2] + tba[2]]
            if cb[2] < 0.0:
                continue
            invz = F32(1.0 / cb[2])
            x, y = F32(cb[0] * float(invz)), F32(cb[1] * float(invz))
            u, v = F32(fx * x) + cx, F32(fy * y) + cy
            if not dst.in_image(u, v):
                continue
            dist3d = F32(_norm(cb))
            if dist3d < S.min_inv(p) or dist3d > S.max_inv(p):
                continue
            lvl = S.predict_scale(p, dist3d)
            cand = g.features_in_area(u, v, F32(F32(th) * S.scale[lvl]))
            if not cand:
                continue
            d = _dists(dst.desc[cand], S.mp_desc[p])
            best, bi = 1 << 31, -1
            for k, idx in enumerate(cand):
                if dst.octave(idx) < lvl - 1 or dst.octave(idx) > lvl:
                    continue
                if int(d[k]) < best:
                    best, bi = int(d[k]), idx
            if best <= TH_HIGH:
                out[i] = bi
        return out

    f12 = one_way(k1, done1, R1, t1, sR21, t21, k2)
    f21 = one_way(k2, done2, R2, t2, sR12, t12, k1)
    found = 0
    for a in range(N1):
        b = f12[a]
        if b >= 0 and f21[b] == a:
            m12[a] = k2.mp[b]
            found += 1
    return found


def fuse_keyframe(S, ki, kf, points, th=3.0):
    """ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th) (src/ORBmatcher.cc:724-842): sequential, with the map mutations."""
    R, t = _rt(kf.T)
    Ow = [-x for x in _mul_t(R, t)]
    fused = 0
    g = kf.grid()
    for p in points:
        if p < 0 or S.mp_bad[p] or ki in S.obs[p]:
            continue
        pr = _project_gates(S, kf, R, t, Ow, p)
        if pr is None:
            continue
        u, v, lvl = pr
        cand = g.features_in_area(u, v, F32(F32(th) * S.scale[lvl]))
        if not cand:
            continue
        d = _dists(kf.desc[cand], S.mp_desc[p])
        best, bi = 256, -1
        for k, idx in enumerate(cand):
            o = kf.octave(idx)
            if o < lvl - 1 or o > lvl:
                continue
            ex, ey = F32(u - kf.kpu[idx, 0]), F32(v - kf.kpu[idx, 1])
            e2 = F32(ex * ex) + F32(ey * ey)
            if float(F32(e2 * S.inv_sigma2[o])) > 5.99:
                continue
            if int(d[k]) < best:
                best, bi = int(d[k]), idx
        if best <= TH_LOW:
            q = kf.mp[bi]
            if q >= 0:
                if not S.mp_bad[q]:
                    if S.nobs[q] > S.nobs[p]:
                        S.replace(p, q)
                    else:
                        S.replace(q, p)
            else:
                S.add_observation(p, ki, bi); kf.mp[bi] = p
            fused += 1
    return fused


def fuse_sim3(S, ki, kf, Scw, points, th):
    """ORBmatcher::Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint) (src/ORBmatcher.cc:844-954)."""
    R, t, Ow = _decompose_sim3(Scw)
    already = set(p for p in kf.mp if p >= 0 and not S.mp_bad[p])          # KeyFrame::GetMapPoints
    replace = [-1] * len(points)
    fused = 0
    g = kf.grid()
    for k, p in enumerate(points):
        if S.mp_bad[p] or p in already:
            continue
        pr = _project_gates(S, kf, R, t, Ow, p)
        if pr is None:
            continue
        u, v, lvl = pr
        cand = g.features_in_area(u, v, F32(F32(th) * S.scale[lvl]))
        if not cand:
            continue
        d = _dists(kf.desc[cand], S.mp_desc[p])
        best, bi = 1 << 31, -1
        for c, idx in enumerate(cand):
            if kf.octave(idx) < lvl - 1 or kf.octave(idx) > lvl:
                continue
            if int(d[c]) < best:
                best, bi = int(d[c]), idx
        if best <= TH_LOW:
            q = kf.mp[bi]
            if q >= 0:
                if not S.mp_bad[q]:
                    replace[k] = q
            else:
                S.add_observation(p, ki, bi); kf.mp[bi] = p
            fused += 1
    return fused, replace


# ------------------------------------------------------------------------------------------------ CeresOptimizer
def _pose7(oracle, T):
    return oracle.matrix4d_to_pose7(np.ascontiguousarray(T, np.float64))


def pose_optimization(S, oracle, F):
    """CeresOptimizer::PoseOptimization (src/CeresOptimizer.cc:275-342)."""
    slots = [i for i in range(F.N) if F.mp[i] >= 0]
    for i in slots:
        F.outl[i] = False
    if len(slots) < 3:
        return 0
    K4 = np.array(F.K, np.float64)
    Xw = np.array([S.pos[F.mp[i]] for i in slots], np.float64)
    uv = np.array([[F.kpu[i, 0], F.kpu[i, 1]] for i in slots], np.float64)
    w = np.array([S.inv_sigma2[F.octave(i)] for i in slots], np.float32)
    n_in, pose, out, _ = oracle.pose_optimization(K4, _pose7(oracle, F.T), Xw, uv, w)
    for k, i in enumerate(slots):
        F.outl[i] = bool(out[k])
    F.T = oracle.pose7_to_matrix4d(pose); F.n_set_pose += 1
    return int(n_in)


def bundle_adjustment(S, oracle, kf_list, mp_list, n_iterations, n_loop_kf, robust):
    """CeresOptimizer::BundleAdjustment (src/CeresOptimizer.cc:59-225) as GlobalBundleAdjustemnt calls it."""
    cams = [k for k in kf_list if not S.kfs[k].bad]
    if not kf_list:
        return
    cam_of = {k: c for c, k in enumerate(cams)}
    max_id = max([S.kfs[k].id for k in cams] + [0])
    poses = np.array([_pose7(oracle, S.kfs[k].T) for k in cams], np.float64)
    K4 = np.array([S.kfs[k].K for k in cams], np.float64)
    fixed = np.array([S.kfs[k].id == 0 for k in cams], np.uint8)
    pts, pt_of, oc, op, uv, w = [], {}, [], [], [], []
    for p in mp_list:
        if S.mp_bad[p]:
            continue
        edges = []
        for k in sorted(S.obs[p]):
            kf = S.kfs[k]
            if kf.bad or kf.id > max_id or k not in cam_of:
                continue
            i = S.obs[p][k]
            edges.append((cam_of[k], float(kf.kpu[i, 0]), float(kf.kpu[i, 1]), float(S.inv_sigma2[kf.octave(i)])))
        if not edges:
            continue
        pt_of[p] = len(pts); pts.append(S.pos[p].copy())
        for c, x, y, ww in edges:
            oc.append(c); op.append(pt_of[p]); uv.append([x, y]); w.append(ww)
    rob = np.full(len(oc), 1 if robust else 0, np.uint8)
    poses, pts3, _ = oracle.ba_solve(K4, poses, fixed, np.array(pts, np.float64).reshape(-1, 3), np.array(oc, np.int32), np.array(op, np.int32),
                                     np.array(uv, np.float64).reshape(-1, 2), np.array(w, np.float64), rob, n_iterations)
    for c, k in enumerate(cams):
        T = oracle.pose7_to_matrix4d(poses[c])
        if n_loop_kf == 0:
            S.kfs[k].set_pose(T)
        else:
            S.kfs[k].gba_T = T; S.kfs[k].ba_global = n_loop_kf
    for p in mp_list:
        if p not in pt_of or S.mp_bad[p]:
            continue
        if n_loop_kf == 0:
            S.pos[p] = pts3[pt_of[p]]; S.n_update_normal[p] += 1
        else:
            S.gba_pos[p] = pts3[pt_of[p]]; S.mp_ba_global[p] = n_loop_kf


def local_bundle_adjustment(S, oracle, ki, abort):
    """CeresOptimizer::LocalBundleAdjustment (src/CeresOptimizer.cc:344-599)."""
    kf = S.kfs[ki]
    local = [ki]
    kf.ba_local = kf.id
    for nb in kf.conn:                                          # GetVectorCovisibleKeyFrames
        S.kfs[nb].ba_local = kf.id
        if not S.kfs[nb].bad and nb not in local:
            local.append(nb)
    pts = set()
    for k in local:
        for p in S.kfs[k].mp:
            if p >= 0 and not S.mp_bad[p] and S.mp_ba_local[p] != kf.id:
                pts.add(p); S.mp_ba_local[p] = kf.id
    pts = sorted(pts)                                           # std::map<MapPoint*, ...>: address order = index order
    fixed = set()
    for p in pts:
        for k in sorted(S.obs[p]):
            o = S.kfs[k]
            if o.ba_local != kf.id and o.ba_fixed != kf.id:
                o.ba_fixed = kf.id
                if not o.bad:
                    fixed.add(k)
    cams = local + sorted(fixed)
    cam_of = {k: c for c, k in enumerate(cams)}
    poses = np.array([_pose7(oracle, S.kfs[k].T) for k in cams], np.float64)
    K4 = np.array([S.kfs[k].K for k in cams], np.float64)
    is_local = np.array([c < len(local) for c in range(len(cams))], np.uint8)
    is_fixed = np.array([(c >= len(local)) or S.kfs[k].id == 0 for c, k in enumerate(cams)], np.uint8)
    oc, op, uv, w, edge = [], [], [], [], []
    for j, p in enumerate(pts):
        for k in sorted(S.obs[p]):
            o = S.kfs[k]
            if o.bad or k not in cam_of:
                continue
            i = S.obs[p][k]
            oc.append(cam_of[k]); op.append(j); uv.append([float(o.kpu[i, 0]), float(o.kpu[i, 1])]); w.append(S.inv_sigma2[o.octave(i)]); edge.append((k, p))
    if abort:
        return
    X = np.array([S.pos[p] for p in pts], np.float64).reshape(-1, 3)
    rc, poses, pts3, erase, _, _ = oracle.local_ba(K4, poses, is_fixed, is_local, X, np.array(oc, np.int32), np.array(op, np.int32),
                                                    np.array(uv, np.float64).reshape(-1, 2), np.array(w, np.float32))
    for e, (k, p) in enumerate(edge):
        if erase[e]:
            i = S.obs[p].get(k, -1)                             # KeyFrame::EraseMapPointMatch(MapPoint*), MapPoint::EraseObservation
            if i >= 0:
                S.kfs[k].mp[i] = -1
            S.erase_observation(p, k)
    for c, k in enumerate(local):
        S.kfs[k].set_pose(oracle.pose7_to_matrix4d(poses[c]))
    for j, p in enumerate(pts):
        S.pos[p] = pts3[j]; S.n_update_normal[p] += 1


def _se3_as_sim3(oracle, T):
    p7 = _pose7(oracle, T)                                      # Sophus::Sim3d(RxSO3d(1.0, R), t): q = Quaterniond(R), scale 1
    return np.array([p7[3], p7[4], p7[5], p7[6], p7[0], p7[1], p7[2]], np.float64)


def optimize_sim3(S, oracle, k1, i1, k2, i2, m12, s12, th2):
    """CeresOptimizer::OptimizeSim3 (src/CeresOptimizer.cc:601-735): returns (inliers, S12)."""
    R1, t1 = _rt(k1.T)
    R2, t2 = _rt(k2.T)
    P2c, o1, w1, P1c, o2, w2 = [], [], [], [], [], []
    for i, q in enumerate(m12):
        if q < 0:
            continue
        p = k1.mp[i]
        j = S.obs[q].get(i2, -1)
        if p < 0 or S.mp_bad[p] or S.mp_bad[q] or j < 0:
            continue
        c2 = _mul(R2, [float(x) for x in S.pos[q]])
        c1 = _mul(R1, [float(x) for x in S.pos[p]])
        P2c.append([c2[0] + t2[0], c2[1] + t2[1], c2[2] + t2[2]]); o1.append([float(k1.kpu[i, 0]), float(k1.kpu[i, 1])]); w1.append(S.inv_sigma2[k1.octave(i)])
        P1c.append([c1[0] + t1[0], c1[1] + t1[1], c1[2] + t1[2]]); o2.append([float(k2.kpu[j, 0]), float(k2.kpu[j, 1])]); w2.append(S.inv_sigma2[k2.octave(j)])
    a = lambda v, d: np.array(v, np.float64).reshape(-1, d)
    n, S12, _, _ = oracle.optimize_sim3(np.array(k1.K, np.float64), np.array(k2.K, np.float64), np.array(s12, np.float64), a(P2c, 3), a(o1, 2),
                                         np.array(w1, np.float32), a(P1c, 3), a(o2, 2), np.array(w2, np.float32), float(th2))
    return int(n), S12, len(w1)


def optimize_essential_graph(S, oracle, loop_kf, cur_kf, non_corrected, corrected, loop_connections):
    """CeresOptimizer::OptimizeEssentialGraph (src/CeresOptimizer.cc:737-957).  non_corrected / corrected: {kf: qt7},
    loop_connections: {kf: sorted list of kf}.  Vertices = the non-bad keyframes of the map, in map order."""
    MINW = 100
    kfs = [k for k in S.map_kfs if not S.kfs[k].bad]
    vtx = {k: v for v, k in enumerate(kfs)}
    lie = np.array([oracle.sim3_log(corrected[k] if k in corrected else _se3_as_sim3(oracle, S.kfs[k].T)) for k in kfs], np.float64)
    fixed = np.array([k == loop_kf for k in kfs], np.uint8)
    ej, ei, es = [], [], []

    def block(k):
        return oracle.sim3_exp(lie[vtx[k]])

    def world(k):
        return np.array(non_corrected[k], np.float64) if k in non_corrected else block(k)

    def edge(j, i, Sjw, Swi):
        ej.append(vtx[j]); ei.append(vtx[i]); es.append(oracle.sim3_mul(Sjw, Swi))

    inserted = set()
    for i in sorted(loop_connections):
        if i not in vtx:
            continue
        Swi = oracle.sim3_inverse(block(i))
        for j in sorted(loop_connections[i]):
            if (S.kfs[i].id != S.kfs[cur_kf].id or S.kfs[j].id != S.kfs[loop_kf].id) and S.kfs[i].weights.get(j, 0) < MINW:
                continue
            if j not in vtx:
                continue
            edge(j, i, block(j), Swi)
            inserted.add((min(S.kfs[i].id, S.kfs[j].id), max(S.kfs[i].id, S.kfs[j].id)))
    for i in S.map_kfs:
        if i not in vtx:
            continue
        kf = S.kfs[i]
        Swi = oracle.sim3_inverse(world(i))
        par = kf.parent
        if par >= 0 and par in vtx:
            edge(par, i, world(par), Swi)
        for l in kf.loop_edges:
            if S.kfs[l].id < kf.id and l in vtx:
                edge(l, i, world(l), Swi)
        nw = 0
        while nw < len(kf.conn_w) and kf.conn_w[nw] >= MINW:   # KeyFrame::GetCovisiblesByWeight (src/KeyFrame.cc:218-235): nothing when ALL weights pass
            nw += 1
        covis = kf.conn[:nw] if (kf.conn and nw < len(kf.conn_w)) else []
        for nb in covis:
            if nb == par or nb in kf.children or nb in kf.loop_edges:
                continue
            o = S.kfs[nb]
            if o.bad or not o.id < kf.id:
                continue
            if (min(kf.id, o.id), max(kf.id, o.id)) in inserted:
                continue
            if nb in vtx:
                edge(nb, i, world(nb), Swi)
    lie_opt, _ = oracle.optimize_essential_graph(lie, fixed, np.array(ej, np.int32), np.array(ei, np.int32), np.array(es, np.float64).reshape(-1, 7), 100)
    by_id = {S.kfs[k].id: k for k in kfs}
    pts, ref = [], []
    for p in S.map_mps:
        if S.mp_bad[p]:
            continue
        r = by_id.get(S.corrected_ref[p], -1) if S.corrected_by[p] == S.kfs[cur_kf].id else S.ref_kf[p]
        if r not in vtx:
            continue
        pts.append(p); ref.append(vtx[r])
    Tiw, newp = oracle.essential_graph_correct(lie, lie_opt, np.array(ref, np.int32), np.array([S.pos[p] for p in pts], np.float64).reshape(-1, 3))
    for v, k in enumerate(kfs):
        T = np.eye(4); T[:3, :] = np.asarray(Tiw[v]).reshape(3, 4)
        S.kfs[k].set_pose(T)
    for j, p in enumerate(pts):
        S.pos[p] = newp[j]; S.n_update_normal[p] += 1
    return len(ej)


# ------------------------------------------------------------------------------------------------ state comparison
def compare(A, B, pose_tol=0.0, point_tol=0.0, mp_rows=None):
    """Differences between two scenes (list of strings, empty = equal).  Poses / points within the given tolerances (0 =
    bit-identical), everything else exact.  mp_rows: number of leading map points to compare (default all of A)."""
    diffs = []

    def eq(name, a, b):
        if a != b:
            diffs.append("%s: %r != %r" % (name, a if not isinstance(a, list) or len(a) < 12 else "[%d]" % len(a), b if not isinstance(b, list) or len(b) < 12 else "[%d]" % len(b)))

    def close(name, a, b, tol):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
        if a.shape != b.shape:
            diffs.append("%s: shape %s != %s" % (name, a.shape, b.shape)); return
        if tol == 0.0:
            if not np.array_equal(a, b):
                diffs.append("%s: differs (max %.3e)" % (name, np.abs(a - b).max()))
        elif a.size and np.abs(a - b).max() > tol:
            diffs.append("%s: max difference %.3e > %.1e" % (name, np.abs(a - b).max(), tol))

    def owner(name, a, b):
        eq(name + ".mp", a.mp, b.mp)
        eq(name + ".fv", a.fv, b.fv)
        eq(name + ".bow_words", sorted(a.bow), sorted(b.bow))
        close(name + ".bow_values", [a.bow[k] for k in sorted(a.bow)], [b.bow.get(k, np.nan) for k in sorted(a.bow)], 0.0)
        close(name + ".T", a.T, b.T, pose_tol)
        eq(name + ".n_set_pose", a.n_set_pose, b.n_set_pose)

    for k, (a, b) in enumerate(zip(A.kfs, B.kfs)):
        n = "kf%d" % k
        owner(n, a, b)
        for f in ("bad", "ba_local", "ba_fixed", "ba_global"):
            eq(n + "." + f, getattr(a, f), getattr(b, f))
        close(n + ".gba_T", a.gba_T, b.gba_T, pose_tol)
    for k, (a, b) in enumerate(zip(A.frames, B.frames)):
        n = "fr%d" % k
        owner(n, a, b)
        eq(n + ".outl", a.outl, b.outl)
    m = len(A.mp_id) if mp_rows is None else mp_rows
    for f in ("mp_bad", "replaced", "nobs", "mp_ba_local", "mp_ba_global", "n_update_normal", "in_view", "obs"):
        eq("mp." + f, getattr(A, f)[:m], getattr(B, f)[:m])
    iv = np.array(A.in_view[:m], bool)
    eq("mp.track_level", [l for l, s in zip(A.track_level[:m], iv) if s], [l for l, s in zip(B.track_level[:m], iv) if s])
    for f in ("track_u", "track_v", "track_cos"):
        close("mp." + f, getattr(A, f)[:m][iv], getattr(B, f)[:m][iv], 0.0)
    scale = np.maximum(1.0, np.linalg.norm(B.pos[:m], axis=1))[:, None]
    close("mp.pos", A.pos[:m] / scale, B.pos[:m] / scale, point_tol)
    gs = np.maximum(1.0, np.linalg.norm(B.gba_pos[:m], axis=1))[:, None]
    close("mp.gba_pos", A.gba_pos[:m] / gs, B.gba_pos[:m] / gs, point_tol)
    close("mp.desc", A.mp_desc[:m], B.mp_desc[:m], 0.0)          # (MapPoint::Replace recomputes the survivor's descriptor)
    eq("map.kfs", A.map_kfs, B.map_kfs)
    return diffs


# ------------------------------------------------------------------------------------------------ case replay
def _sc(R, name):
    return R[name][0]


def check_case(path, oracle):
    """Replay the case file written by tests/cpp/test_dropin.cpp; returns (list of failures, summary string)."""
    import os
    R = load_records(path)
    name = os.path.basename(path)[:-4]
    S = Scene(R, "before")
    A = Scene(R, "after")
    fails = []
    pose_tol = point_tol = 0.0
    mp_rows = None
    info = ""

    def expect(what, got, want):
        if isinstance(got, np.ndarray) or isinstance(want, np.ndarray):
            ok = np.array_equal(np.asarray(got), np.asarray(want))
        else:
            ok = got == want
        if not ok:
            fails.append("%s: drop-in %r, checker %r" % (what, got if np.size(got) < 12 else "[%d]" % np.size(got), want if np.size(want) < 12 else "[%d]" % np.size(want)))

    if name.startswith("search_local_points"):
        F = S.frames[0]; th = int(_sc(R, "arg.th")); pts = [int(p) for p in R["arg.local_map_points"]]
        n_view = sum(is_in_frustum(S, F, p, 0.5) for p in pts if not S.mp_bad[p])
        ret = search_by_projection_points(S, F, pts, float(th), 0.8) if n_view > 0 else 0
        expect("n_to_match", int(_sc(R, "out.n_to_match")), n_view); expect("ret", int(_sc(R, "ret")), ret)
        info = "%d in view, %d matches" % (n_view, ret)
        if ret < 100: fails.append("weak case: %d matches" % ret)
    elif name.startswith("track_motion_model"):
        ret = search_by_projection_last_frame(S, S.frames[1], S.frames[0], float(_sc(R, "arg.th")), True)
        expect("ret", int(_sc(R, "ret")), ret); info = "%d matches" % ret
        if ret < 100: fails.append("weak case: %d matches" % ret)
    elif name == "relocalization_projection":
        ret = search_by_projection_keyframe(S, S.frames[0], S.kfs[int(_sc(R, "arg.kf"))], set(int(p) for p in R["arg.found"]), 10.0, 100, True)
        expect("ret", int(_sc(R, "ret")), ret); info = "%d matches" % ret
        if ret < 50: fails.append("weak case: %d matches" % ret)
    elif name == "loop_projection":
        matched = [int(v) for v in R["arg.matched"]]
        ret = search_by_projection_sim3(S, S.kfs[int(_sc(R, "arg.kf"))], R["arg.Scw"].reshape(4, 4), [int(p) for p in R["arg.points"]], matched, 10)
        expect("ret", int(_sc(R, "ret")), ret); expect("matched", R["out.matched"].tolist(), matched); info = "%d matches" % ret
        if ret < 30: fails.append("weak case: %d matches" % ret)
    elif name == "bow_kf_frame":
        ret, out = search_by_bow_frame(S, S.kfs[5], S.frames[0], 0.7, True)
        expect("ret", int(_sc(R, "ret")), ret); expect("matches", R["out.matches"].tolist(), out); info = "%d matches" % ret
        if ret < 50: fails.append("weak case: %d matches" % ret)
    elif name == "bow_kf_kf":
        ret, out = search_by_bow_keyframes(S, S.kfs[5], S.kfs[1], 0.75, True)
        expect("ret", int(_sc(R, "ret")), ret); expect("matches", R["out.matches"].tolist(), out); info = "%d matches" % ret
        if ret < 30: fails.append("weak case: %d matches" % ret)
    elif name == "initialization":
        prev = S.frames[0].kpu[:, :2].copy()
        ret, m12 = search_for_initialization(S, S.frames[0], S.frames[1], prev, 100, 0.9, True)
        expect("ret", int(_sc(R, "ret")), ret); expect("matches", R["out.matches"].tolist(), m12)
        expect("prev_matched", R["out.prev_matched"].reshape(-1, 2), prev); info = "%d matches" % ret
        if ret < 20: fails.append("weak case: %d matches" % ret)
    elif name == "triangulation_search":
        ret, pairs = search_for_triangulation(S, S.kfs[int(_sc(R, "arg.kf1"))], S.kfs[int(_sc(R, "arg.kf2"))], R["arg.F12"].reshape(3, 3), False)
        expect("ret", int(_sc(R, "ret")), ret); expect("pairs", R["out.pairs"].reshape(-1, 2).tolist(), [list(p) for p in pairs]); info = "%d pairs" % ret
        if ret < 20: fails.append("weak case: %d pairs" % ret)
    elif name == "sim3_search_and_optimize":
        i1, i2 = int(_sc(R, "arg.kf1")), int(_sc(R, "arg.kf2"))
        m12 = [int(v) for v in R["arg.matches"]]
        nf = search_by_sim3(S, S.kfs[i1], i1, S.kfs[i2], i2, m12, float(_sc(R, "arg.s")), R["arg.R"].reshape(3, 3), R["arg.t"], 7.5)
        expect("nfound", int(_sc(R, "out.nfound")), nf); expect("matches", R["out.matches"].tolist(), m12)
        # Sophus::Sim3d(RxSO3d(s, R), t) as the call site builds it, recomputed with the oracle's codec
        T = np.eye(4); T[:3, :3] = R["arg.R"].reshape(3, 3); T[:3, 3] = R["arg.t"]
        g = _se3_as_sim3(oracle, T); g[:4] *= math.sqrt(float(_sc(R, "arg.s"))) / np.linalg.norm(g[:4])
        if np.abs(g - R["arg.gScm"]).max() > 1e-12: fails.append("gScm construction differs: %.2e" % np.abs(g - R["arg.gScm"]).max())
        n_in, S12, ncorr = optimize_sim3(S, oracle, S.kfs[i1], i1, S.kfs[i2], i2, m12, R["arg.gScm"], 10.0)
        expect("n_inliers", int(_sc(R, "ret")), n_in)
        if np.abs(S12 - R["out.gScm"]).max() > 1e-6: fails.append("S12 differs by %.2e" % np.abs(S12 - R["out.gScm"]).max())
        info = "%d found, %d correspondences, %d inliers, S12 diff %.1e" % (nf, ncorr, n_in, np.abs(S12 - R["out.gScm"]).max())
        if nf < 30 or ncorr < 30: fails.append("weak case")
    elif name == "fuse_kf":
        ki = int(_sc(R, "arg.kf"))
        ret = fuse_keyframe(S, ki, S.kfs[ki], [int(p) for p in R["arg.points"]])
        expect("ret", int(_sc(R, "ret")), ret); info = "%d fused, %d replaced" % (ret, sum(1 for a, b in zip(S.mp_bad, Scene(R, "before").mp_bad) if a and not b))
        if ret < 30: fails.append("weak case: %d fused" % ret)
    elif name == "fuse_many":                                   # LocalMapping::SearchInNeighbors, first loop (src/LocalMapping.cc:437-442)
        pts = [int(p) for p in R["arg.points"]]
        rets = [fuse_keyframe(S, int(ki), S.kfs[int(ki)], pts) for ki in R["arg.kfs"]]
        expect("ret", [int(v) for v in R["ret"]], rets)
        before = Scene(R, "before")
        changed = sum(1 for a, b in zip(S.mp_desc, before.mp_desc) if not np.array_equal(a, b))
        info = "%s fused per keyframe, %d replaced, %d descriptors recomputed" % (rets, sum(1 for a, b in zip(S.mp_bad, before.mp_bad) if a and not b), changed)
        if sum(rets) < 60 or changed < 3: fails.append("weak case: %s fused, %d descriptors changed" % (rets, changed))
    elif name == "fuse_sim3":
        ki = int(_sc(R, "arg.kf"))
        ret, rep = fuse_sim3(S, ki, S.kfs[ki], R["arg.Scw"].reshape(4, 4), [int(p) for p in R["arg.points"]], 4.0)
        expect("ret", int(_sc(R, "ret")), ret); expect("replace", R["out.replace"].tolist(), rep); info = "%d fused, %d to replace" % (ret, sum(r >= 0 for r in rep))
        if ret < 20: fails.append("weak case: %d fused" % ret)
    elif name == "search_and_fuse":                             # LoopClosing::SearchAndFuse (src/LoopClosing.cc:599-630): Fuse, then the Replace loop, keyframe after keyframe
        pts = [int(p) for p in R["arg.points"]]
        Scws = R["arg.Scws"].reshape(-1, 4, 4)
        rets = []; nrep = 0
        for ki, Scw in zip(R["arg.kfs"], Scws):
            ret, rep = fuse_sim3(S, int(ki), S.kfs[int(ki)], Scw, pts, 4.0)
            rets.append(ret)
            for k, q in enumerate(rep):
                if q >= 0:
                    S.replace(q, pts[k]); nrep += 1
        expect("ret", [int(v) for v in R["ret"]], rets)
        before = Scene(R, "before")
        changed = sum(1 for a, b in zip(S.mp_desc, before.mp_desc) if not np.array_equal(a, b))
        info = "%s fused per keyframe, %d replaced, %d descriptors recomputed" % (rets, nrep, changed)
        if sum(rets) < 60 or nrep < 10 or changed < 3: fails.append("weak case: %s fused, %d replaced, %d descriptors changed" % (rets, nrep, changed))
    elif name.startswith("pose_optimization"):
        F = S.frames[int(_sc(R, "arg.frame"))]
        ret = pose_optimization(S, oracle, F)
        expect("ret", int(_sc(R, "ret")), ret); pose_tol = 1e-7; info = "%d inliers, %d outliers" % (ret, sum(F.outl))
        if name == "pose_optimization" and sum(F.outl) < 10: fails.append("weak case: %d outliers" % sum(F.outl))
    elif name.startswith("global_ba"):
        bundle_adjustment(S, oracle, S.map_kfs, S.map_mps, int(_sc(R, "arg.n_iterations")), int(_sc(R, "arg.n_loop_kf")), bool(_sc(R, "arg.robust")))
        pose_tol, point_tol = 1e-6, 1e-5; info = "%d SetPose calls" % sum(k.n_set_pose for k in S.kfs)
    elif name.startswith("local_ba"):
        before = Scene(R, "before")
        local_bundle_adjustment(S, oracle, int(_sc(R, "arg.kf")), bool(_sc(R, "arg.abort")))
        pose_tol, point_tol = 1e-6, 1e-5
        erased = sum(a != b for ka, kb in zip(before.kfs, S.kfs) for a, b in zip(ka.mp, kb.mp)); moved = sum(k.n_set_pose for k in S.kfs)
        info = "%d observations erased, %d SetPose calls" % (erased, moved)
        if bool(_sc(R, "arg.abort")):
            if erased or moved: fails.append("an aborted run must not touch the map")
        elif erased < 10 or moved != 4: fails.append("weak case: erased %d moved %d" % (erased, moved))
    elif name == "essential_graph":
        ck = [int(k) for k in R["arg.corrected_kf"]]; nk = [int(k) for k in R["arg.non_corrected_kf"]]
        cor = dict(zip(ck, R["arg.corrected_sim3"].reshape(-1, 7))); non = dict(zip(nk, R["arg.non_corrected_sim3"].reshape(-1, 7)))
        lc = {}
        for i, j in R["arg.loop_connections"].reshape(-1, 2):
            lc.setdefault(int(i), []).append(int(j))
        cur = int(_sc(R, "arg.cur_kf"))
        # the corrected Sim3 of the group as src/LoopClosing.cc:447-469 forms them, recomputed with the oracle's Sim3 product
        Twc = np.linalg.inv(S.kfs[cur].T)
        for k in ck:
            want = cor[cur] if k == cur else oracle.sim3_mul(_se3_as_sim3(oracle, S.kfs[k].T @ Twc), cor[cur])
            if np.abs(want - cor[k]).max() > 1e-9: fails.append("corrected Sim3 of keyframe %d differs by %.1e" % (k, np.abs(want - cor[k]).max()))
        n_edges = optimize_essential_graph(S, oracle, int(_sc(R, "arg.loop_kf")), cur, non, cor, lc)
        pose_tol, point_tol = 1e-6, 1e-6
        moved = max(np.abs(a.T - b.T).max() for a, b in zip(Scene(R, "before").kfs, A.kfs))
        info = "%d edges, largest pose change %.3f" % (n_edges, moved)
        if moved < 0.05: fails.append("weak case: nothing moved")
    elif name == "compute_bow":
        voc = dict(node_desc=R["arg.voc_node_desc"].reshape(-1, 32), child_off=R["arg.voc_child_off"].astype(np.uint32), children=R["arg.voc_children"].astype(np.uint32),
                   word_id=R["arg.voc_word_id"], weight=R["arg.voc_weight"], L=int(_sc(R, "arg.voc_L")))
        for o, kf_rule in ((S.frames[0], False), (S.kfs[2], True), (S.kfs[3], True)):
            if not (len(o.bow) == 0 or (kf_rule and len(o.fv) == 0)):
                continue
            bw, bv, fn, fo, fi = oracle.bow_transform(voc, o.desc, 4)
            o.bow = {int(w): float(v) for w, v in zip(bw, bv)}
            o.fv = {int(fn[m]): [int(v) for v in fi[fo[m]:fo[m + 1]]] for m in range(len(fn))}
        info = "%d words / %d nodes (frame), %d / %d (keyframe)" % (len(S.frames[0].bow), len(S.frames[0].fv), len(S.kfs[2].bow), len(S.kfs[2].fv))
        if len(S.frames[0].fv) < 4 or len(S.frames[0].bow) < 100: fails.append("weak case")
    elif name == "features_in_area":
        q = R["arg.queries"].reshape(-1, 3); lv = R["arg.levels"].reshape(-1, 2); off = R["out.off"]; idx = R["out.idx"]
        tot = 0
        for k in range(len(q)):
            o = S.frames[1] if k & 1 else S.kfs[4]
            want = o.grid().features_in_area(q[k, 0], q[k, 1], q[k, 2], int(lv[k, 0]), int(lv[k, 1]))
            expect("query %d" % k, idx[off[k]:off[k + 1]].tolist(), want); tot += len(want)
        info = "%d candidates" % tot
        if tot < 200: fails.append("weak case")
    elif name in ("create_new_map_points", "create_new_map_points_batched"):
        ki = int(_sc(R, "arg.kf")); kf = S.kfs[ki]
        nnew = 0
        n0 = len(S.mp_id)
        R1, t1 = _rt(kf.T); Ow1 = [-x for x in _mul_t(R1, t1)]
        newpos = []
        for nb in kf.conn[:20]:                                  # GetBestCovisibilityKeyFrames(20)
            o = S.kfs[nb]
            R2, t2 = _rt(o.T); Ow2 = [-x for x in _mul_t(R2, t2)]
            baseline = _norm([Ow2[0] - Ow1[0], Ow2[1] - Ow1[1], Ow2[2] - Ow1[2]])
            depths = sorted(float(o.T[2, :3] @ S.pos[p] + o.T[2, 3]) for p in o.mp if p >= 0)
            if baseline / depths[(len(depths) - 1) // 2] < 0.01:
                continue
            T12 = kf.T @ np.linalg.inv(o.T)                     # LocalMapping::ComputeF12 (src/LocalMapping.cc:482-503)
            tx = np.array([[0, -T12[2, 3], T12[1, 3]], [T12[2, 3], 0, -T12[0, 3]], [-T12[1, 3], T12[0, 3], 0]])
            Ki = np.array([[1.0 / kf.K[0], 0, -kf.K[2] / kf.K[0]], [0, 1.0 / kf.K[1], -kf.K[3] / kf.K[1]], [0, 0, 1]], np.float64)
            F12 = Ki.T @ tx @ T12[:3, :3] @ Ki
            _, pairs = search_for_triangulation(S, kf, o, F12, False)
            if not pairs:
                continue
            kp1 = np.array([[kf.kpu[a, 0], kf.kpu[a, 1], kf.kpu[a, 2]] for a, _ in pairs], np.float32)
            kp2 = np.array([[o.kpu[b, 0], o.kpu[b, 1], o.kpu[b, 2]] for _, b in pairs], np.float32)
            X, ok = oracle.triangulate_matches(kf.T[:3, :], o.T[:3, :], np.array(kf.K, np.float32), np.array(o.K, np.float32), kp1, kp2, S.sigma2, S.scale,
                                               float(F32(1.5) * F32(1.2)))
            for k, (a, b) in enumerate(pairs):
                if not ok[k]:
                    continue
                p = n0 + nnew                                    # the new MapPoint: observations in both keyframes, appended to the map
                S.mp_id.append(-1); S.mp_bad.append(False); S.replaced.append(-1); S.nobs.append(2); S.mp_ba_local.append(NEVER); S.mp_ba_global.append(0)
                S.n_update_normal.append(1); S.in_view.append(False); S.track_level.append(0); S.obs.append({ki: a, nb: b})
                kf.mp[a] = p; o.mp[b] = p; S.map_mps.append(p); newpos.append(X[k]); nnew += 1
        expect("ret", int(_sc(R, "ret")), nnew)
        expect("map.mps", A.map_mps, S.map_mps)
        got = A.pos[n0:]
        if len(got) == len(newpos) and nnew:
            err = np.abs(got - np.array(newpos)).max() / max(1.0, np.abs(got).max())
            if err > 1e-9: fails.append("triangulated points differ by %.2e" % err)
        mp_rows = n0
        for f in ("mp_bad", "nobs", "n_update_normal", "obs"):
            if getattr(A, f)[n0:] != getattr(S, f)[n0:]: fails.append("new points: %s differs" % f)
        info = "%d new points" % nnew
        if nnew < 20: fails.append("weak case: %d new points" % nnew)
    else:
        fails.append("no checker for case %s" % name)
    fails += compare(A, S, pose_tol, point_tol, mp_rows)
    return fails, info
