"""Seeded synthetic inputs for the ORB front-end and the bundle-adjustment back-end.

These are the workloads BASELINE.md / SURVEY.md 8(d) define (there are no datasets in
the image and no network): u8 frames in three families, frame sequences that are pure
translations of one canvas, KITTI-intrinsics pose problems and BA graphs.  numpy only.
"""
import numpy as np

KITTI_K4 = np.array([718.856, 718.856, 607.1928, 185.2157], np.float64)  # reference configs/KITTI00-02.yaml:8-11


# --------------------------------------------------------------------------- frames
def _blocks(rng, w, h):
    img = np.full((h, w), float(rng.integers(90, 160)), np.float32)
    n = int(rng.integers(200, 600)) * max(1, (w * h) // (640 * 480))
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        hw, hh = rng.uniform(4, 40), rng.uniform(4, 40)
        g = float(rng.integers(0, 256))
        x0, x1 = int(max(0, cx - 60)), int(min(w, cx + 60))
        y0, y1 = int(max(0, cy - 60)), int(min(h, cy + 60))
        if x1 <= x0 or y1 <= y0:
            continue
        if rng.random() < 0.5:
            th = 0.0
        else:
            th = rng.uniform(0, np.pi)
        c, s = np.cos(th), np.sin(th)
        X = xx[y0:y1, x0:x1] - cx
        Y = yy[y0:y1, x0:x1] - cy
        m = (np.abs(c * X + s * Y) <= hw) & (np.abs(-s * X + c * Y) <= hh)
        img[y0:y1, x0:x1][m] = g
    img += rng.normal(0, 2.0, img.shape).astype(np.float32)
    return img


def _checker(rng, w, h):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.full((h, w), 128.0)
    for _ in range(3):
        H = np.eye(3)
        H[0, 0], H[1, 1] = rng.uniform(0.6, 1.4), rng.uniform(0.6, 1.4)
        H[0, 1], H[1, 0] = rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3)
        H[2, 0], H[2, 1] = rng.uniform(-6e-4, 6e-4), rng.uniform(-6e-4, 6e-4)
        H[0, 2], H[1, 2] = rng.uniform(-50, 50), rng.uniform(-50, 50)
        den = H[2, 0] * xx + H[2, 1] * yy + 1.0
        u = (H[0, 0] * xx + H[0, 1] * yy + H[0, 2]) / den
        v = (H[1, 0] * xx + H[1, 1] * yy + H[1, 2]) / den
        sq = rng.uniform(14, 40)
        par = (np.floor(u / sq) + np.floor(v / sq)).astype(np.int64) & 1
        lo, hi = rng.integers(20, 100), rng.integers(150, 240)
        layer = np.where(par == 1, float(hi), float(lo))
        x0, x1 = sorted(rng.integers(0, w, 2)); y0, y1 = sorted(rng.integers(0, h, 2))
        if x1 - x0 < w // 4:
            x0, x1 = 0, w
        if y1 - y0 < h // 4:
            y0, y1 = 0, h
        img[y0:y1, x0:x1] = layer[y0:y1, x0:x1]
    img += rng.normal(0, 2.0, img.shape)
    return img.astype(np.float32)


def _flat(rng, w, h):
    return (float(rng.integers(60, 200)) + rng.normal(0, 3.0, (h, w))).astype(np.float32)


FAMILIES = ("blocks", "checker", "flat")


def make_canvas(seed, w, h, family="blocks", margin=0):
    """One u8 canvas of (h+2*margin) x (w+2*margin)."""
    rng = np.random.default_rng(seed)
    W, Hh = w + 2 * margin, h + 2 * margin
    f = {"blocks": _blocks, "checker": _checker, "flat": _flat}[family](rng, W, Hh)
    return np.clip(np.rint(f), 0, 255).astype(np.uint8)


def make_frame(seed, w, h, family="blocks"):
    return make_canvas(seed, w, h, family, 0)


def make_sequence(seed, w, h, n, family="blocks", max_shift=8):
    """n frames, frame t+1 = frame t translated by (dx,dy) in [-max_shift,max_shift]^2
    (crops of one canvas, so true correspondences exist).  Returns (frames[n,h,w] u8, offsets[n,2])."""
    rng = np.random.default_rng(seed + 7919)
    m = max_shift * 4
    canvas = make_canvas(seed, w, h, family, m)
    ox, oy = m, m
    frames = np.empty((n, h, w), np.uint8)
    offs = np.zeros((n, 2), np.int32)
    for t in range(n):
        frames[t] = canvas[oy:oy + h, ox:ox + w]
        offs[t] = (ox - m, oy - m)
        dx, dy = rng.integers(-max_shift, max_shift + 1, 2)
        ox = int(np.clip(ox + dx, 0, 2 * m)); oy = int(np.clip(oy + dy, 0, 2 * m))
    return frames, offs


# --------------------------------------------------------------------------- geometry
def quat_from_rotvec(rv):
    """[x,y,z,w] unit quaternion of a rotation vector."""
    rv = np.asarray(rv, np.float64)
    a = np.linalg.norm(rv)
    if a < 1e-300:
        return np.array([0, 0, 0, 1.0])
    ax = rv / a
    return np.concatenate([ax * np.sin(a / 2), [np.cos(a / 2)]])


def quat_mul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz])


def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def project(K4, pose7, X):
    R = quat_to_R(pose7[3:7]); p = X @ R.T + pose7[:3]
    return np.stack([K4[0] * p[:, 0] / p[:, 2] + K4[2], K4[1] * p[:, 1] / p[:, 2] + K4[3]], 1), p[:, 2]


_SCALE = np.float32(1.2) ** np.arange(8, dtype=np.float32)
_QUOTA_P = np.array([434, 362, 302, 251, 209, 175, 145, 122], np.float64) / 2000.0


def _octave_inv_sigma2(rng, n):
    octv = rng.choice(8, size=n, p=_QUOTA_P / _QUOTA_P.sum())
    sc = _SCALE[octv]
    return octv, sc.astype(np.float64), (np.float32(1.0) / (sc * sc)).astype(np.float32)


def make_pose_problem(seed, n=2000, outlier_frac=0.05, w=1241, h=376):
    """C3: one KITTI camera, n observations, sigma = 1 px * scale[octave], gross outliers."""
    rng = np.random.default_rng(seed)
    K4 = KITTI_K4.copy()
    q_gt = quat_from_rotvec(rng.normal(0, 0.05, 3)); t_gt = rng.normal(0, 0.3, 3)
    pose_gt = np.concatenate([t_gt, q_gt])
    uv = np.stack([rng.uniform(20, w - 20, n), rng.uniform(20, h - 20, n)], 1)
    z = rng.uniform(4, 60, n)
    pc = np.stack([(uv[:, 0] - K4[2]) / K4[0] * z, (uv[:, 1] - K4[3]) / K4[1] * z, z], 1)
    R = quat_to_R(q_gt)
    Xw = (pc - t_gt) @ R           # X = R^T (pc - t)
    octv, sc, inv_sigma2 = _octave_inv_sigma2(rng, n)
    obs = uv + rng.normal(0, 1.0, (n, 2)) * sc[:, None]
    nout = int(round(outlier_frac * n))
    if nout:
        idx = rng.choice(n, nout, replace=False)
        obs[idx] += rng.choice([-1, 1], (nout, 2)) * rng.uniform(30, 50, (nout, 2))
    q0 = quat_mul(quat_from_rotvec(rng.normal(0, np.deg2rad(0.5), 3)), q_gt)
    pose0 = np.concatenate([t_gt + rng.normal(0, 0.05, 3), q0])
    return dict(K4=K4, pose0=pose0, pose_gt=pose_gt, Xw=Xw, uv=obs, inv_sigma2=inv_sigma2, octave=octv)


def make_sim3_problem(seed, n=120, outlier_frac=0.1, scale=1.15, noise=1.0, perturb=(0.02, 0.1, 0.03), w=1241, h=376):
    """Loop-closure pair for OptimizeSim3: n matched map points seen by keyframe 1 and keyframe 2 whose maps differ by
    the similarity S12 (P1c = S12 * P2c, scale drift `scale`).  Returns the flattened correspondences of
    include/orbslam_hip.h::ba_optimize_sim3, the ground truth and a perturbed start, s12 in [qx,qy,qz,qw(|q|^2=s),t]."""
    rng = np.random.default_rng(seed)
    K1 = KITTI_K4.copy(); K2 = KITTI_K4.copy()
    uv1 = np.stack([rng.uniform(100, w - 100, n), rng.uniform(60, h - 60, n)], 1)
    z = rng.uniform(5, 40, n)
    P1c = np.stack([(uv1[:, 0] - K1[2]) / K1[0] * z, (uv1[:, 1] - K1[3]) / K1[1] * z, z], 1)
    q = quat_from_rotvec(rng.normal(0, 0.04, 3)); t = rng.normal(0, 0.4, 3)
    R = quat_to_R(q)
    P2c = ((P1c - t) @ R) / scale                       # P2c = S12^-1 P1c
    uv2 = np.stack([K2[0] * P2c[:, 0] / P2c[:, 2] + K2[2], K2[1] * P2c[:, 1] / P2c[:, 2] + K2[3]], 1)
    _, sc1, w1 = _octave_inv_sigma2(rng, n)
    _, sc2, w2 = _octave_inv_sigma2(rng, n)
    obs1 = uv1 + rng.normal(0, noise, (n, 2)) * sc1[:, None]
    obs2 = uv2 + rng.normal(0, noise, (n, 2)) * sc2[:, None]
    nout = int(round(outlier_frac * n))
    if nout:
        idx = rng.choice(n, nout, replace=False)
        obs1[idx] += rng.choice([-1, 1], (nout, 2)) * rng.uniform(15, 40, (nout, 2))
    s12_gt = np.concatenate([np.sqrt(scale) * q, t])
    dr, dt, ds = perturb
    q0 = quat_mul(quat_from_rotvec(rng.normal(0, dr, 3)), q)
    s0 = scale * float(np.exp(rng.normal(0, ds)))
    s12_0 = np.concatenate([np.sqrt(s0) * q0, t + rng.normal(0, dt, 3)])
    return dict(K1=K1, K2=K2, s12_0=s12_0, s12_gt=s12_gt, P3D2c=P2c, obs1=obs1, inv_sigma2_1=w1, P3D1c=P1c, obs2=obs2,
                inv_sigma2_2=w2)


def make_ba_graph(seed, ncam=100, npts=10000, nobs=50000, outlier_frac=0.05, noise=1.0, perturb=True,
                  n_fixed=1, w=1241, h=376, max_depth=60.0, min_len=2):
    """C4/C5-style graph: forward-moving KITTI cameras, points in the frusta, each point seen by a
    run of consecutive cameras (mean track length nobs/npts).  Camera 0..n_fixed-1 are fixed (gauge).
    The defaults are SURVEY 8(d)'s workload (depth 4-60 m, tracks of >= 2 views 0.8 m apart: many points have a parallax
    below 1 degree, so the problem is ill-conditioned by construction); max_depth / min_len make a well-conditioned
    variant of the same shape for the parity tests that need one."""
    rng = np.random.default_rng(seed)
    K4 = KITTI_K4.copy()
    step = 0.8
    poses_gt = np.zeros((ncam, 7))
    yaw = np.cumsum(rng.normal(0, 0.01, ncam))
    for c in range(ncam):
        q = quat_from_rotvec([0, yaw[c], 0])
        R = quat_to_R(q)
        C = np.array([np.sum(np.sin(yaw[:c + 1])) * step * 0 + 0.02 * c * np.sin(yaw[c]), 0.0, step * c])
        poses_gt[c, :3] = -R @ C        # Tcw: p_c = R X + t, camera centre C
        poses_gt[c, 3:] = q
    mean_len = nobs / npts
    pts = np.zeros((npts, 3))
    oc, op = [], []
    lens = np.clip(rng.poisson(max(mean_len - min_len, 0.0), npts) + min_len, min_len, ncam)
    # adjust total to nobs exactly
    diff = int(lens.sum() - nobs)
    order = rng.permutation(npts)
    i = 0
    while diff != 0 and i < 50 * npts:
        p = order[i % npts]
        if diff > 0 and lens[p] > min_len:
            lens[p] -= 1; diff -= 1
        elif diff < 0 and lens[p] < ncam:
            lens[p] += 1; diff += 1
        i += 1
    for p in range(npts):
        L = int(lens[p])
        c0 = int(rng.integers(0, ncam - L + 1))
        cm = c0 + L // 2
        # place the point in the frustum of the middle camera, far enough to be seen by the run
        u = rng.uniform(100, w - 100); v = rng.uniform(40, h - 40)
        z = rng.uniform(min(4 + step * L, max_depth - 1.0), max_depth)
        pc = np.array([(u - K4[2]) / K4[0] * z, (v - K4[3]) / K4[1] * z, z])
        R = quat_to_R(poses_gt[cm, 3:])
        pts[p] = R.T @ (pc - poses_gt[cm, :3])
        for c in range(c0, c0 + L):
            oc.append(c); op.append(p)
    oc = np.array(oc, np.int32); op = np.array(op, np.int32)
    n = len(oc)
    uv = np.zeros((n, 2))
    for c in range(ncam):
        m = oc == c
        if m.any():
            uv[m], _ = project(K4, poses_gt[c], pts[op[m]])
    octv, sc, inv_sigma2 = _octave_inv_sigma2(rng, n)
    obs = uv + rng.normal(0, noise, (n, 2)) * sc[:, None]
    nout = int(round(outlier_frac * n))
    if nout:
        idx = rng.choice(n, nout, replace=False)
        obs[idx] += rng.choice([-1, 1], (nout, 2)) * rng.uniform(30, 50, (nout, 2))
    poses0 = poses_gt.copy(); pts0 = pts.copy()
    if perturb:
        for c in range(n_fixed, ncam):
            poses0[c, 3:] = quat_mul(quat_from_rotvec(rng.normal(0, np.deg2rad(0.5) / np.sqrt(3), 3)), poses_gt[c, 3:])
            poses0[c, :3] += rng.normal(0, 0.05 / np.sqrt(3), 3)
        pts0 += pts * rng.normal(0, 0.01, (npts, 1))
    cam_fixed = np.zeros(ncam, np.uint8); cam_fixed[:n_fixed] = 1
    return dict(K4=np.tile(K4, (ncam, 1)), poses0=poses0, poses_gt=poses_gt, cam_fixed=cam_fixed, pts0=pts0,
                pts_gt=pts, obs_cam=oc, obs_pt=op, obs_uv=obs, obs_inv_sigma2=inv_sigma2, octave=octv)



def shuffle_keyframes(g, seed):
    """The same graph with its keyframes listed in a random order (a merged or re-indexed map: ids that do not follow the covisibility
    chain).  The reduced camera system of make_ba_graph's band then fills its whole triangle unless the solver reorders it
    (ORBHIP_BA_ORDER=rcm).  Returns a new dict; `kf_perm[new] = old`."""
    rng = np.random.default_rng(seed)
    ncam = len(g["cam_fixed"])
    perm = rng.permutation(ncam)                      # new position -> old keyframe
    inv = np.empty(ncam, np.int64); inv[perm] = np.arange(ncam)
    h = dict(g)
    for k in ("K4", "poses0", "poses_gt", "cam_fixed"):
        if k in g: h[k] = np.ascontiguousarray(g[k][perm])
    h["obs_cam"] = inv[g["obs_cam"]].astype(g["obs_cam"].dtype)
    h["kf_perm"] = perm
    return h


def _quat_from_R(R):
    """[x,y,z,w] of a rotation matrix (largest-component branch)."""
    m = R; tr = m[0, 0] + m[1, 1] + m[2, 2]
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2; q = [(m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s]
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2; q = [0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s, (m[2, 1] - m[1, 2]) / s]
    elif m[1, 1] > m[2, 2]:
        s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2; q = [(m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s, (m[0, 2] - m[2, 0]) / s]
    else:
        s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2; q = [(m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s, (m[1, 0] - m[0, 1]) / s]
    q = np.array(q); return q / np.linalg.norm(q)


BA_STRUCTURES = ("covis", "dense", "loop")


def make_ba_graph_covis(seed, ncam=100, npts=10000, nobs=50000, structure="covis", window=(15, 30), cur_share=15,
                        outlier_frac=0.05, noise=1.0, perturb=True, n_fixed=1, w=1241, h=376):
    """BA graphs whose REDUCED CAMERA SYSTEM has the structure a map of the reference has, beside make_ba_graph's odometry band.

    The reference's local window is the current keyframe plus ALL its covisible keyframes (src/CeresOptimizer.cc:353-363: every
    local keyframe shares >= 15 map points with the current one) and a landmark is matched across a neighbourhood of keyframes
    with gaps (a keyframe that missed the match does not end the track).  Keyframes are numbered by id, the current keyframe is
    the LAST one (the drop-in flattens keyframes by id, csrc/compat/orbslam_dropin.h).

      structure "covis": a landmark is seen by k >= 2 keyframes drawn at random (gaps) inside a window of window[0]..window[1]
                         consecutive keyframes; on top the current keyframe (ncam - 1) shares `cur_share` landmarks with EVERY
                         other keyframe - a band of window[1] keyframes plus a dense last block row (an arrowhead).
                "dense": a landmark is seen by k random keyframes out of all of them: every keyframe pair shares landmarks, the
                         reduced system is full.
                 "loop": "covis" windows that wrap around (keyframe ncam - 1 is a neighbour of keyframe 0): a chain whose ends
                         are tied by a loop closure - the LAST block rows reach back to column 0 (src/LoopClosing.cc:656 runs
                         GlobalBundleAdjustemnt on exactly such a map); no current-keyframe landmarks.

    Geometry: KITTI intrinsics; the keyframes stand on an arc (a full circle for "loop") of radius 12 m and look at its centre,
    the landmarks fill a flat ellipsoid around the centre, so every landmark is in front of and inside the image of every
    keyframe and WHO sees it is the structure's choice alone.  Noise, octaves, gross outliers and the perturbed start as in
    make_ba_graph.  nobs is met exactly; the mean track length is nobs / npts.  Returns make_ba_graph's dict plus `current`."""
    assert structure in BA_STRUCTURES
    rng = np.random.default_rng(seed)
    K4 = KITTI_K4.copy()
    rho = 12.0
    span = 2 * np.pi * (1.0 - 1.0 / ncam) if structure == "loop" else np.deg2rad(min(1.0 * (ncam - 1), 150.0))
    ang = -0.5 * span + span * np.arange(ncam) / max(ncam - 1, 1) + rng.normal(0, 0.1 * span / max(ncam, 2), ncam)
    poses_gt = np.zeros((ncam, 7))
    for c in range(ncam):
        C = np.array([rho * np.sin(ang[c]), rng.normal(0, 0.05), -rho * np.cos(ang[c])]) * (1.0 + rng.normal(0, 0.01))
        f = -C / np.linalg.norm(C)
        f = quat_to_R(quat_from_rotvec(rng.normal(0, 0.01, 3))) @ f          # not exactly at the centre
        d = np.array([0.0, 1.0, 0.0]); r = np.cross(d, f); r /= np.linalg.norm(r); d = np.cross(f, r)
        R = np.stack([r, d, f])                                              # p_c = R X + t
        poses_gt[c, :3] = -R @ C; poses_gt[c, 3:] = _quat_from_R(R)
    wrap = structure == "loop"
    cur = ncam - 1
    # ---- who sees what
    n_cur_pts = 0 if structure != "covis" else min(cur_share * (ncam - 1), npts // 3)
    per_kf = n_cur_pts // max(ncam - 1, 1) if structure == "covis" else 0
    n_cur_pts = per_kf * (ncam - 1)
    n_win = npts - n_cur_pts
    budget = nobs - 2 * n_cur_pts                                            # a current-keyframe landmark: the current keyframe + one other
    wmax = ncam if structure == "dense" else min(window[1], ncam)
    lens = np.clip(rng.poisson(max(budget / n_win - 2.0, 0.0), n_win) + 2, 2, min(wmax, ncam))
    diff = int(lens.sum() - budget)
    order = rng.permutation(n_win); i = 0
    while diff != 0 and i < 50 * n_win:
        p = order[i % n_win]
        if diff > 0 and lens[p] > 2:
            lens[p] -= 1; diff -= 1
        elif diff < 0 and lens[p] < min(wmax, ncam):
            lens[p] += 1; diff += 1
        i += 1
    assert diff == 0, "nobs cannot be met with these sizes"
    oc, op = [], []
    for p in range(n_win):
        k = int(lens[p])
        if structure == "dense":
            cams = np.sort(rng.choice(ncam, k, replace=False))
        else:
            W = int(min(max(rng.integers(window[0], window[1] + 1), k), ncam))
            s = int(rng.integers(0, ncam if wrap else ncam - W + 1))
            cams = np.sort((s + rng.choice(W, k, replace=False)) % ncam)
        oc.extend(cams.tolist()); op.extend([p] * k)
    for j in range(ncam - 1 if n_cur_pts else 0):                            # the current keyframe's shared landmarks, keyframe by keyframe
        for q in range(per_kf):
            p = n_win + j * per_kf + q
            oc.extend([j, cur]); op.extend([p, p])
    oc = np.array(oc, np.int32); op = np.array(op, np.int32)
    assert len(oc) == nobs
    # ---- landmarks: inside every keyframe's image (checked), in front of every keyframe
    pts = np.zeros((npts, 3)); todo = np.arange(npts)
    Rs = np.stack([quat_to_R(poses_gt[c, 3:]) for c in range(ncam)]); ts = poses_gt[:, :3]
    for _ in range(50):
        if len(todo) == 0:
            break
        u = rng.normal(0, 1, (len(todo), 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
        X = u * (rng.uniform(0, 1, (len(todo), 1)) ** (1 / 3)) * np.array([4.5, 1.4, 4.5])
        ok = np.ones(len(todo), bool)
        for n0 in range(0, len(todo), 4096):                                  # (chunks: ncam x points x 3 doubles)
            pc = np.einsum("cij,nj->cni", Rs, X[n0:n0 + 4096]) + ts[:, None, :]
            uu = K4[0] * pc[..., 0] / pc[..., 2] + K4[2]; vv = K4[1] * pc[..., 1] / pc[..., 2] + K4[3]
            ok[n0:n0 + 4096] = ((pc[..., 2] > 4.0) & (uu > 60) & (uu < w - 60) & (vv > 30) & (vv < h - 30)).all(0)
        pts[todo[ok]] = X[ok]; todo = todo[~ok]
    assert len(todo) == 0
    pc = np.einsum("nij,nj->ni", Rs[oc], pts[op]) + ts[oc]
    uv = np.stack([K4[0] * pc[:, 0] / pc[:, 2] + K4[2], K4[1] * pc[:, 1] / pc[:, 2] + K4[3]], 1)
    octv, sc, inv_sigma2 = _octave_inv_sigma2(rng, nobs)
    obs = uv + rng.normal(0, noise, (nobs, 2)) * sc[:, None]
    nout = int(round(outlier_frac * nobs))
    if nout:
        idx = rng.choice(nobs, nout, replace=False)
        obs[idx] += rng.choice([-1, 1], (nout, 2)) * rng.uniform(30, 50, (nout, 2))
    poses0 = poses_gt.copy(); pts0 = pts.copy()
    if perturb:
        for c in range(n_fixed, ncam):
            poses0[c, 3:] = quat_mul(quat_from_rotvec(rng.normal(0, np.deg2rad(0.5) / np.sqrt(3), 3)), poses_gt[c, 3:])
            poses0[c, :3] += rng.normal(0, 0.05 / np.sqrt(3), 3)
        pts0 += rng.normal(0, 0.05, (npts, 3))
    cam_fixed = np.zeros(ncam, np.uint8); cam_fixed[:n_fixed] = 1
    return dict(K4=np.tile(K4, (ncam, 1)), poses0=poses0, poses_gt=poses_gt, cam_fixed=cam_fixed, pts0=pts0, pts_gt=pts,
                obs_cam=oc, obs_pt=op, obs_uv=obs, obs_inv_sigma2=inv_sigma2, octave=octv, current=cur, structure=structure)


def make_vocabulary(seed, k=10, L=4, flip=40, ragged=0.0, stop_frac=0.02):
    """Synthetic ORB vocabulary tree in the flattened form of include/orbslam_hip.h::orbv_create (stands in for the
    un-shipped ORBvoc.txt, k = 10, L = 6): children descriptors are their parent's with up to `flip` random bits toggled
    (a clustering-like hierarchy), leaves carry consecutive word ids and idf-like weights (a fraction stop_frac are 0 =
    stopped words); `ragged` randomly drops children so that nodes have fewer than k.  Nodes are numbered level by level."""
    rng = np.random.default_rng(seed)
    descs = [rng.integers(0, 256, (1, 32), dtype=np.uint8)]
    counts = []                                      # children count of every node, level by level
    for lev in range(L):
        par = descs[-1]
        m = len(par)
        nk = np.full(m, k, np.int64) if ragged == 0.0 else np.maximum(2, k - rng.binomial(k - 2, ragged, m))
        counts.append(nk)
        rep = np.repeat(np.arange(m), nk)
        out = np.empty((len(rep), 32), np.uint8)
        for c0 in range(0, len(rep), 1 << 16):       # chunked: the bit masks are 256 bytes per child
            r = rep[c0:c0 + (1 << 16)]
            mask = np.zeros((len(r), 256), np.uint8)
            mask[np.arange(len(r))[:, None], rng.integers(0, 256, (len(r), flip))] = 1
            out[c0:c0 + len(r)] = par[r] ^ np.packbits(mask, axis=1)
        descs.append(out)
        flip = max(4, int(flip * 0.7))
    counts.append(np.zeros(len(descs[-1]), np.int64))
    node_desc = np.concatenate(descs)
    nk_all = np.concatenate(counts)
    n = len(node_desc)
    child_off = np.zeros(n + 1, np.uint32); child_off[1:] = np.cumsum(nk_all)
    children = np.arange(1, n, dtype=np.uint32)      # level-by-level numbering: the children of all nodes are consecutive
    word_id = np.full(n, -1, np.int32); weight = np.zeros(n, np.float64)
    leaves = np.nonzero(nk_all == 0)[0]
    word_id[leaves] = np.arange(len(leaves))
    wts = rng.uniform(0.5, 9.0, len(leaves)); wts[rng.random(len(leaves)) < stop_frac] = 0.0
    weight[leaves] = wts
    return dict(node_desc=node_desc, child_off=child_off, children=children, word_id=word_id, weight=weight, L=L, k=k)


def _sim3_qt(s, q, t):
    return np.concatenate([np.sqrt(s) * np.asarray(q, np.float64), np.asarray(t, np.float64)])


def make_essential_graph(seed, n=200, drift=0.002, n_corrected=6, extra_every=5):
    """Loop-closure pose graph for OptimizeEssentialGraph: n keyframes on a closed loop; the estimated Scw drift in scale and
    position along the trajectory; the last n_corrected keyframes (the current keyframe and its neighbours) start from their
    loop-corrected Sim(3) as LoopClosing::CorrectLoop leaves them; vertex 0 (the loop keyframe) is constant.  Edges in the
    reference's insertion order: the loop connection first, then per keyframe its spanning-tree parent and a few older
    covisibility neighbours, each with Sji = Sjw * Swi from the NON-corrected estimates (src/CeresOptimizer.cc:821-905).
    Returns tangents (via the caller's sim3_log), qt7 arrays and the edge lists."""
    rng = np.random.default_rng(seed)
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    true = []; est = []
    scale = 1.0; pos_drift = np.zeros(3)
    for k in range(n):
        q = quat_from_rotvec([0, -ang[k], 0]); R = quat_to_R(q)
        Cw = np.array([30 * np.cos(ang[k]), 0.2 * np.sin(3 * ang[k]), 30 * np.sin(ang[k])])
        t = -R @ Cw
        true.append((1.0, q, t))
        scale *= 1.0 + drift * (1 + 0.3 * rng.normal()); pos_drift = pos_drift + rng.normal(0, 0.01, 3)
        qe = quat_mul(quat_from_rotvec(rng.normal(0, 0.002, 3)), q)
        est.append((1.0 / scale, qe, (t + pos_drift) / scale))     # a map whose scale shrinks along the way
    est[0] = true[0]
    S_true = np.stack([_sim3_qt(*x) for x in true]); S_est = np.stack([_sim3_qt(*x) for x in est])
    init = S_est.copy()
    init[n - n_corrected:] = S_true[n - n_corrected:]              # corrected Sim3 of the current keyframe's neighbourhood
    fixed = np.zeros(n, np.uint8); fixed[0] = 1
    edges = [(0, n - 1, "corr")]                                    # (j, i): the loop connection, from the corrected values
    for i in range(1, n):
        edges.append((i - 1, i, "est"))
        if i % extra_every == 0 and i >= 3:
            edges.append((i - 3, i, "est"))
    return dict(S_true=S_true, S_est=S_est, S_init=init, fixed=fixed, edges=edges)


def make_degenerate_ba(seed):
    """Small BA graphs with one structural degeneracy each (kind = seed % 6): a camera with a single observation,
    duplicated (camera, point) observations, zero-weight observations, every point seen once, gross errors, points seen
    only by fixed cameras; random loss flags 0 / 1 / 2 (2 = Huber block folded with its loss-free twin) and 2 / 6 / 15
    iterations.  Returns a dict; 'oracle_obs' expands the folded twins into the literal duplicated list the oracle solves."""
    rng = np.random.default_rng(4000 + seed)
    ncam = int(rng.integers(2, 12)); npts = int(rng.integers(4, 120)); nobs = int(npts * rng.uniform(1.5, 4))
    g = make_ba_graph(300 + seed, ncam=ncam, npts=npts, nobs=max(nobs, 2 * npts), n_fixed=1, outlier_frac=0.1)
    oc, op, uv = g["obs_cam"].copy(), g["obs_pt"].copy(), g["obs_uv"].copy()
    w = g["obs_inv_sigma2"].astype(np.float64)
    n = len(oc)
    kind = seed % 6
    fixed = g["cam_fixed"].copy()
    if kind == 0:      # a camera with a single observation
        keep = np.ones(n, bool); idx = np.nonzero(oc == ncam - 1)[0]; keep[idx[1:]] = False
        oc, op, uv, w = oc[keep], op[keep], uv[keep], w[keep]
    elif kind == 1:    # duplicated (camera, point) observations
        d = rng.choice(n, n // 5, replace=False); oc = np.concatenate([oc, oc[d]]); op = np.concatenate([op, op[d]]); uv = np.concatenate([uv, uv[d] + 0.3]); w = np.concatenate([w, w[d]])
    elif kind == 2:    # zero-weight observations
        w[rng.random(len(w)) < 0.3] = 0.0
    elif kind == 3:    # every point seen once
        _, first = np.unique(op, return_index=True); oc, op, uv, w = oc[first], op[first], uv[first], w[first]
    elif kind == 4:    # gross errors
        uv[rng.random(len(uv)) < 0.3] += 400
    elif kind == 5:    # points seen only by fixed cameras
        fixed[: max(1, ncam // 2)] = 1
    rb = rng.integers(0, 3, len(oc)).astype(np.uint8)
    iters = int(rng.choice([2, 6, 15]))
    twin = rb == 2
    oracle_obs = (np.concatenate([oc, oc[twin]]), np.concatenate([op, op[twin]]), np.concatenate([uv, uv[twin]]), np.concatenate([w, w[twin]]),
                  np.concatenate([np.where(rb >= 1, 1, 0), np.zeros(int(twin.sum()))]).astype(np.uint8))
    return dict(kind=kind, K4=g["K4"], poses0=g["poses0"], cam_fixed=fixed, pts0=g["pts0"], obs_cam=oc, obs_pt=op, obs_uv=uv, obs_w=w,
                obs_robust=rb, iterations=iters, oracle_obs=oracle_obs)

