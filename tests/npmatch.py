"""Independent pure-Python / numpy restatements of the reference's guided matcher entry points, written from the reference
lines cited at each function -- NOT from oracle/match_oracle.cpp -- so that oracle <-> numpy agreement on the same inputs is
evidence about the reference's semantics and not two copies of one reading ("de-twin the matcher evidence").

Test infrastructure only.  Style: plain loops over Python lists, float32 arithmetic made explicit with np.float32 where the
reference computes in float; data are the flattened arrays the C ABI takes (kps4 = n x {x, y, octave, angle}).
"""
import math

import numpy as np

F32 = np.float32
TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30          # src/ORBmatcher.cc:35-37
GRID_COLS, GRID_ROWS = 64, 48                        # include/Frame.h FRAME_GRID_COLS / FRAME_GRID_ROWS


def _c_round(v):
    """C round(): half away from zero (Frame::PosInGrid, the rotation bins)."""
    return int(math.floor(v + 0.5)) if v >= 0 else -int(math.floor(-v + 0.5))


def descriptor_distance(a, b):
    """ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1422-1437): popcount of the xor of 8 x 32 bits."""
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


class Grid:
    """Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cc:158-173, :309-320) and GetFeaturesInArea (:243-307)."""

    def __init__(self, kps4, bounds):
        self.k = np.asarray(kps4, np.float32)
        self.min_x, self.max_x, self.min_y, self.max_y = [F32(v) for v in bounds]
        self.winv = F32(GRID_COLS) / F32(self.max_x - self.min_x)          # src/Frame.cc:138-141
        self.hinv = F32(GRID_ROWS) / F32(self.max_y - self.min_y)
        self.cells = [[[] for _ in range(GRID_ROWS)] for _ in range(GRID_COLS)]
        for i in range(len(self.k)):
            px = _c_round(float(F32(F32(self.k[i, 0] - self.min_x) * self.winv)))
            py = _c_round(float(F32(F32(self.k[i, 1] - self.min_y) * self.hinv)))
            if px < 0 or px >= GRID_COLS or py < 0 or py >= GRID_ROWS:
                continue
            self.cells[px][py].append(i)

    def features_in_area(self, x, y, r, min_level=-1, max_level=-1):
        x, y, r = F32(x), F32(y), F32(r)
        out = []
        min_cx = max(0, int(math.floor(float(F32(F32(F32(x - self.min_x) - r) * self.winv)))))
        if min_cx >= GRID_COLS:
            return out
        max_cx = min(GRID_COLS - 1, int(math.ceil(float(F32(F32(F32(x - self.min_x) + r) * self.winv)))))
        if max_cx < 0:
            return out
        min_cy = max(0, int(math.floor(float(F32(F32(F32(y - self.min_y) - r) * self.hinv)))))
        if min_cy >= GRID_ROWS:
            return out
        max_cy = min(GRID_ROWS - 1, int(math.ceil(float(F32(F32(F32(y - self.min_y) + r) * self.hinv)))))
        if max_cy < 0:
            return out
        check = (min_level > 0) or (max_level >= 0)
        for ix in range(min_cx, max_cx + 1):
            for iy in range(min_cy, max_cy + 1):
                for j in self.cells[ix][iy]:
                    if check:
                        octv = int(self.k[j, 2])
                        if octv < min_level:
                            continue
                        if max_level >= 0 and octv > max_level:
                            continue
                    if abs(F32(self.k[j, 0] - x)) < r and abs(F32(self.k[j, 1] - y)) < r:
                        out.append(j)
        return out


def three_maxima(sizes):
    """ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:1386-1418) on the bin populations."""
    max1 = max2 = max3 = 0
    i1 = i2 = i3 = -1
    for i, s in enumerate(sizes):
        if s > max1:
            max3, max2, max1 = max2, max1, s
            i3, i2, i1 = i2, i1, i
        elif s > max2:
            max3, max2 = max2, s
            i3, i2 = i2, i
        elif s > max3:
            max3, i3 = s, i
    if max2 < F32(0.1) * F32(max1):
        i2 = i3 = -1
    elif max3 < F32(0.1) * F32(max1):
        i3 = -1
    return i1, i2, i3


def rot_bin(a1, a2):
    """rot = a1 - a2; if (rot < 0) rot += 360; bin = round(rot * (1/30)); 30 -> 0  (e.g. src/ORBmatcher.cc:431-437)."""
    rot = F32(F32(a1) - F32(a2))
    if rot < 0.0:
        rot = F32(rot + F32(360.0))
    b = _c_round(float(F32(rot * (F32(1.0) / F32(HISTO_LENGTH)))))
    return 0 if b == HISTO_LENGTH else b


def _keep_bins(hist):
    i1, i2, i3 = three_maxima([len(h) for h in hist])
    return [i for i in range(HISTO_LENGTH) if i not in (i1, i2, i3)]


# ------------------------------------------------------------------------------------------------ M3
def search_for_initialization(kps1, d1, kps2, d2, bounds2, prev_matched, window, nnratio, check_ori):
    """ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:363-468)."""
    g2 = Grid(kps2, bounds2)
    n1, n2 = len(kps1), len(kps2)
    m12 = [-1] * n1
    m21 = [-1] * n2
    mdist = [2 ** 31 - 1] * n2
    hist = [[] for _ in range(HISTO_LENGTH)]
    prev = np.array(prev_matched, np.float32).copy()
    nm = 0
    for i1 in range(n1):
        level1 = int(kps1[i1, 2])
        if level1 > 0:
            continue
        cand = g2.features_in_area(prev[i1, 0], prev[i1, 1], window, level1, level1)
        if not cand:
            continue
        best, best2, bidx = 2 ** 31 - 1, 2 ** 31 - 1, -1
        for i2 in cand:
            dist = descriptor_distance(d1[i1], d2[i2])
            if mdist[i2] <= dist:
                continue
            if dist < best:
                best2, best, bidx = best, dist, i2
            elif dist < best2:
                best2 = dist
        if best <= TH_LOW and best < F32(best2) * F32(nnratio):
            if m21[bidx] >= 0:
                m12[m21[bidx]] = -1
                nm -= 1
            m12[i1] = bidx
            m21[bidx] = i1
            mdist[bidx] = best
            nm += 1
            if check_ori:
                hist[rot_bin(kps1[i1, 3], kps2[bidx, 3])].append(i1)
    if check_ori:
        for b in _keep_bins(hist):
            for i1 in hist[b]:
                if m12[i1] >= 0:
                    m12[i1] = -1
                    nm -= 1
    for i1 in range(n1):
        if m12[i1] >= 0:
            prev[i1] = kps2[m12[i1], :2]
    return np.array(m12, np.int32), nm, prev


# ------------------------------------------------------------------------------------------------ M4
def search_by_projection_mappoints(K, D, bounds, q_uv, q_radius, q_level, q_desc, q_valid, taken, nnratio):
    """SearchByProjection(Frame&, vector<MapPoint*>&, th) (src/ORBmatcher.cc:42-119) from GetFeaturesInArea on.
    q_level = nPredictedLevel; `taken` = F.map_points_[idx] holds a point with observations (in/out)."""
    g = Grid(K, bounds)
    taken = list(taken)
    match = [-1] * len(q_uv); bdist = [0] * len(q_uv); n = 0
    for q in range(len(q_uv)):
        if not q_valid[q]:
            continue
        cand = g.features_in_area(q_uv[q, 0], q_uv[q, 1], q_radius[q], int(q_level[q]) - 1, int(q_level[q]))
        if not cand:
            continue
        best, blvl, best2, blvl2, bidx = 256, -1, 256, -1, -1
        for idx in cand:
            if taken[idx]:
                continue
            dist = descriptor_distance(q_desc[q], D[idx])
            if dist < best:
                best2, best = best, dist
                blvl2, blvl = blvl, int(K[idx, 2])
                bidx = idx
            elif dist < best2:
                blvl2 = int(K[idx, 2])
                best2 = dist
        bdist[q] = best
        if best <= TH_HIGH:
            if blvl == blvl2 and best > F32(nnratio) * F32(best2):
                continue
            match[q] = bidx
            if not (int(q_valid[q]) & 2):           # Observations() > 0 of the assigned point closes the feature (:83-84)
                taken[bidx] = 1
            n += 1
    return n, np.array(match, np.int32), np.array(taken, np.uint8)


# ------------------------------------------------------------------------------------------------ M5 / M7
def search_by_projection_frame(K, D, bounds, q_uv, q_radius, q_level, q_desc, q_valid, q_angle, taken, th, check_ori):
    """SearchByProjection(Frame& cur, const Frame& last, th) (src/ORBmatcher.cc:1161-1271, th = TH_HIGH, q_level = the last
    frame's octave) and SearchByProjection(Frame&, KeyFrame*, set&, th, ORBdist) (:1273-1384, th = ORBdist, q_level = predicted
    level): levels [l-1, l+1], best only, rotation histogram over the CURRENT frame's indices."""
    g = Grid(K, bounds)
    taken = list(taken)
    match = [-1] * len(q_uv); n = 0
    hist = [[] for _ in range(HISTO_LENGTH)]
    for q in range(len(q_uv)):
        if not q_valid[q]:
            continue
        cand = g.features_in_area(q_uv[q, 0], q_uv[q, 1], q_radius[q], int(q_level[q]) - 1, int(q_level[q]) + 1)
        if not cand:
            continue
        best, bidx = 256, -1
        for i2 in cand:
            if taken[i2]:
                continue
            dist = descriptor_distance(q_desc[q], D[i2])
            if dist < best:
                best, bidx = dist, i2
        if best <= th:
            match[q] = bidx
            if not (int(q_valid[q]) & 2):           # (:1220-1221)
                taken[bidx] = 1
            n += 1
            if check_ori:
                hist[rot_bin(q_angle[q], K[bidx, 3])].append((q, bidx))
    if check_ori:
        for b in _keep_bins(hist):
            for q, i2 in hist[b]:
                taken[i2] = 0                       # CurrentFrame.map_points_[i2] = nullptr
                match[q] = -2 - i2                  # (removed: the value keeps which slot was reset)
                n -= 1
    return n, np.array(match, np.int32), np.array(taken, np.uint8)


# ------------------------------------------------------------------------------------------------ M12
def search_by_projection_sim3(K, D, bounds, q_uv, q_radius, q_pred, q_desc, q_valid, matched):
    """SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (src/ORBmatcher.cc:258-361) from GetFeaturesInArea on:
    KeyFrame::GetFeaturesInArea has no level filter, the level gate [pred-1, pred] is inside the loop, TH_LOW."""
    g = Grid(K, bounds)
    matched = list(matched)
    match = [-1] * len(q_uv); n = 0
    for q in range(len(q_uv)):
        if not q_valid[q]:
            continue
        cand = g.features_in_area(q_uv[q, 0], q_uv[q, 1], q_radius[q])
        if not cand:
            continue
        best, bidx = 256, -1
        for idx in cand:
            if matched[idx]:
                continue
            lvl = int(K[idx, 2])
            if lvl < q_pred[q] - 1 or lvl > q_pred[q]:
                continue
            dist = descriptor_distance(q_desc[q], D[idx])
            if dist < best:
                best, bidx = dist, idx
        if best <= TH_LOW:
            match[q] = bidx
            matched[bidx] = 1
            n += 1
    return n, np.array(match, np.int32), np.array(matched, np.uint8)


# ------------------------------------------------------------------------------------------------ M10 (candidate selection)
def fuse_candidates(K, D, bounds, q_uv, q_radius, q_pred, q_desc, q_valid, inv_level_sigma2):
    """Fuse(KeyFrame*, vector<MapPoint*>&, th) (src/ORBmatcher.cc:724-842), the per-point search :775-811: level gate, then the
    chi-square gate e2 * invSigma2 > 5.99, best <= TH_LOW; no `taken` state (the map mutation is the caller's)."""
    g = Grid(K, bounds)
    match = [-1] * len(q_uv); n = 0
    for q in range(len(q_uv)):
        if not q_valid[q]:
            continue
        u, v = F32(q_uv[q, 0]), F32(q_uv[q, 1])
        cand = g.features_in_area(u, v, q_radius[q])
        best, bidx = 256, -1
        for idx in cand:
            lvl = int(K[idx, 2])
            if lvl < q_pred[q] - 1 or lvl > q_pred[q]:
                continue
            ex = F32(u - K[idx, 0]); ey = F32(v - K[idx, 1])
            e2 = F32(F32(ex * ex) + F32(ey * ey))
            if F32(e2 * F32(inv_level_sigma2[lvl])) > 5.99:
                continue
            dist = descriptor_distance(q_desc[q], D[idx])
            if dist < best:
                best, bidx = dist, idx
        if best <= TH_LOW:
            match[q] = bidx
            n += 1
    return n, np.array(match, np.int32)


# ------------------------------------------------------------------------------------------------ M6 / M9
def _fv_lists(fv):
    nodes, off, idx = fv
    return [(int(nodes[m]), [int(i) for i in idx[off[m]:off[m + 1]]]) for m in range(len(nodes))]


def _lower_bound(lst, pos, key):
    while pos < len(lst) and lst[pos][0] < key:
        pos += 1
    return pos


def search_by_bow(d1, valid1, a1, d2, valid2, a2, fv1, fv2, nnratio, strict, check_ori):
    """strict = False: SearchByBoW(KeyFrame*, Frame&, ...) (src/ORBmatcher.cc:151-256): set 1 = keyframe (its map points must
    exist = valid1), set 2 = frame, `vpMapPointMatches[realIdxF]` marks taken frame features, accept best <= TH_LOW; the
    histogram stores FRAME indices and the result is indexed by frame feature.
    strict = True: SearchByBoW(KeyFrame*, KeyFrame*, ...) (:470-580): both sets need map points (valid1, valid2), vbMatched2,
    accept best < TH_LOW, histogram over idx1.
    Returns match12 (index into set 2 per element of set 1) in both cases, as the C ABI does."""
    L1, L2 = _fv_lists(fv1), _fv_lists(fv2)
    n1, n2 = len(d1), len(d2)
    m12 = [-1] * n1
    taken2 = [False] * n2
    hist = [[] for _ in range(HISTO_LENGTH)]
    nm = 0
    p1 = p2 = 0
    while p1 < len(L1) and p2 < len(L2):
        if L1[p1][0] == L2[p2][0]:
            for idx1 in L1[p1][1]:
                if valid1 is not None and not valid1[idx1]:
                    continue
                best1, bidx, best2 = 256, -1, 256
                for idx2 in L2[p2][1]:
                    if taken2[idx2]:
                        continue
                    if strict and valid2 is not None and not valid2[idx2]:
                        continue
                    dist = descriptor_distance(d1[idx1], d2[idx2])
                    if dist < best1:
                        best2, best1, bidx = best1, dist, idx2
                    elif dist < best2:
                        best2 = dist
                ok = best1 < TH_LOW if strict else best1 <= TH_LOW
                if ok and F32(best1) < F32(nnratio) * F32(best2):
                    m12[idx1] = bidx
                    taken2[bidx] = True
                    if check_ori:
                        hist[rot_bin(a1[idx1], a2[bidx])].append(idx1)
                    nm += 1
            p1 += 1
            p2 += 1
        elif L1[p1][0] < L2[p2][0]:
            p1 = _lower_bound(L1, p1, L2[p2][0])
        else:
            p2 = _lower_bound(L2, p2, L1[p1][0])
    if check_ori:
        for b in _keep_bins(hist):
            for idx1 in hist[b]:
                m12[idx1] = -1
                nm -= 1
    return nm, np.array(m12, np.int32)


# ------------------------------------------------------------------------------------------------ M8
def check_dist_epipolar_line(kp1, kp2, F12, level_sigma2):
    """ORBmatcher::CheckDistEpipolarLine (src/ORBmatcher.cc:128-149): the products are double (Eigen) and every sum is
    narrowed to float on assignment."""
    x1, y1 = float(kp1[0]), float(kp1[1])
    a = F32(x1 * F12[0, 0] + y1 * F12[1, 0] + F12[2, 0])
    b = F32(x1 * F12[0, 1] + y1 * F12[1, 1] + F12[2, 1])
    c = F32(x1 * F12[0, 2] + y1 * F12[1, 2] + F12[2, 2])
    num = F32(F32(F32(a * F32(kp2[0])) + F32(b * F32(kp2[1]))) + c)
    den = F32(F32(a * a) + F32(b * b))
    if den == 0:
        return False
    dsqr = F32(F32(num * num) / den)
    return float(dsqr) < 3.84 * float(level_sigma2[int(kp2[2])])


def search_for_triangulation(k1, d1, unmapped1, k2, d2, unmapped2, fv1, fv2, F12, epipole, scale_factors, level_sigma2, check_ori):
    """ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:582-722), monocular (bStereo false everywhere).  vbMatched2 is
    never set in this fork (":602,:643"), so queries are independent; a later candidate with an EQUAL distance replaces
    the earlier one (`dist > bestDist -> continue`, ":654")."""
    L1, L2 = _fv_lists(fv1), _fv_lists(fv2)
    F12 = np.asarray(F12, np.float64).reshape(3, 3)
    ex, ey = F32(epipole[0]), F32(epipole[1])
    m12 = [-1] * len(k1)
    hist = [[] for _ in range(HISTO_LENGTH)]
    nm = 0
    p1 = p2 = 0
    while p1 < len(L1) and p2 < len(L2):
        if L1[p1][0] == L2[p2][0]:
            for idx1 in L1[p1][1]:
                if not unmapped1[idx1]:
                    continue
                best, bidx = TH_LOW, -1
                for idx2 in L2[p2][1]:
                    if not unmapped2[idx2]:
                        continue
                    dist = descriptor_distance(d1[idx1], d2[idx2])
                    if dist > TH_LOW or dist > best:
                        continue
                    dx = F32(ex - k2[idx2, 0]); dy = F32(ey - k2[idx2, 1])
                    if F32(F32(dx * dx) + F32(dy * dy)) < F32(100) * F32(scale_factors[int(k2[idx2, 2])]):
                        continue
                    if check_dist_epipolar_line(k1[idx1], k2[idx2], F12, level_sigma2):
                        bidx, best = idx2, dist
                if bidx >= 0:
                    m12[idx1] = bidx
                    nm += 1
                    if check_ori:
                        hist[rot_bin(k1[idx1, 3], k2[bidx, 3])].append(idx1)
            p1 += 1
            p2 += 1
        elif L1[p1][0] < L2[p2][0]:
            p1 = _lower_bound(L1, p1, L2[p2][0])
        else:
            p2 = _lower_bound(L2, p2, L1[p1][0])
    if check_ori:
        for b in _keep_bins(hist):
            for idx1 in hist[b]:
                m12[idx1] = -1
                nm -= 1
    return nm, np.array(m12, np.int32)


# ------------------------------------------------------------------------------------------------ M11
def search_by_sim3(K1, D1, K2, D2, bounds, q12_uv, q12_radius, q12_pred, q12_valid, q21_uv, q21_radius, q21_pred, q21_valid,
                   q12_desc=None, q21_desc=None, bounds2=None):
    """ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:956-1159) from the two GetFeaturesInArea calls on: direction 1->2 searches
    keyframe 2 for the map point of every keyframe-1 feature (descriptor = that feature's, i.e. the map point's
    representative descriptor passed by the caller), direction 2->1 likewise; best <= TH_HIGH each way, no `taken` state;
    a pair is kept iff vnMatch2[vnMatch1[i1]] == i1 (:1145-1157).  Returns (nFound, match12)."""
    def one_way(Kt, Dt, bt, q_uv, q_rad, q_pred, q_valid, q_desc):
        g = Grid(Kt, bt)                                  # the TARGET keyframe's own grid (:1022, :1102)
        out = [-1] * len(q_uv)
        for q in range(len(q_uv)):
            if not q_valid[q]:
                continue
            best, bidx = 2 ** 31 - 1, -1
            for idx in g.features_in_area(q_uv[q, 0], q_uv[q, 1], q_rad[q]):
                lvl = int(Kt[idx, 2])
                if lvl < q_pred[q] - 1 or lvl > q_pred[q]:
                    continue
                dist = descriptor_distance(q_desc[q], Dt[idx])
                if dist < best:
                    best, bidx = dist, idx
            if best <= TH_HIGH:
                out[q] = bidx
        return out
    m1 = one_way(K2, D2, bounds if bounds2 is None else bounds2, q12_uv, q12_radius, q12_pred, q12_valid, D1 if q12_desc is None else q12_desc)
    m2 = one_way(K1, D1, bounds, q21_uv, q21_radius, q21_pred, q21_valid, D2 if q21_desc is None else q21_desc)
    m12 = [-1] * len(K1)
    found = 0
    for i1 in range(len(K1)):
        idx2 = m1[i1]
        if idx2 >= 0 and m2[idx2] == i1:
            m12[i1] = idx2
            found += 1
    return found, np.array(m12, np.int32)
