"""Independent numpy / pure-Python re-statements used to cross-check the C++ oracle
(SURVEY.md 8(c) "cross-checks available in this container").  Deliberately written in a
different style (vectorised integer numpy, literal definitions) from oracle/*.cpp."""
import numpy as np


def np_resize_linear_u8(src, dw, dh):
    """cv::resize INTER_LINEAR 8UC1 fixed-point."""
    sh, sw = src.shape

    def coef(dn, sn, clamp):
        scale = 1.0 / (np.float64(dn) / np.float64(sn))
        d = np.arange(dn, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if clamp:
            lo = s < 0
            f[lo] = 0; s[lo] = 0
            hi = s >= sn - 1
            f[hi] = 0; s[hi] = sn - 1
        c1 = np.rint((f * np.float32(2048)).astype(np.float64)).astype(np.int64)
        c0 = np.rint(((np.float32(1) - f) * np.float32(2048)).astype(np.float64)).astype(np.int64)
        return s, c0, c1

    sx, a0, a1 = coef(dw, sw, True)
    sy, b0, b1 = coef(dh, sh, False)
    S = src.astype(np.int64)
    sx1 = np.minimum(sx + 1, sw - 1)
    H = S[:, sx] * a0[None, :] + S[:, sx1] * a1[None, :]
    y0 = np.clip(sy, 0, sh - 1); y1 = np.clip(sy + 1, 0, sh - 1)
    out = (((b0[:, None] * (H[y0] >> 4)) >> 16) + ((b1[:, None] * (H[y1] >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def np_gaussian_blur7(src):
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    p = np.pad(src.astype(np.int64), 3, mode="reflect")       # numpy 'reflect' == BORDER_REFLECT_101
    h, w = src.shape
    rows = sum(k[t] * p[:, t:t + w] for t in range(7))         # (h+6, w)
    cols = sum(k[t] * rows[t:t + h, :] for t in range(7))
    return np.clip((cols + (1 << 15)) >> 16, 0, 255).astype(np.uint8)


RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def py_fast_score_map(img, t):
    """Literal FAST-9/16: corner test at threshold t and score = largest threshold at which the pixel is
    still a corner, by brute force over all 16 arcs."""
    h, w = img.shape
    I = img.astype(np.int64)
    sc = np.zeros((h, w), np.int64)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            p = I[y, x]
            ring = [I[y + dy, x + dx] for dx, dy in RING]
            best = -1
            for s in range(16):
                arc = [ring[(s + k) % 16] for k in range(9)]
                best = max(best, min(a - p for a in arc), min(p - a for a in arc))
            # corner at t  <=>  best > t ; score = best - 1
            if best > t:
                sc[y, x] = best - 1
    return sc


def py_fast_nms(img, t):
    sc = py_fast_score_map(img, t)
    h, w = img.shape
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = sc[y, x]
            if s <= 0:
                continue
            nb = sc[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if (s > nb).all():
                out.append((x, y, int(s)))
    return np.array(out, np.int32).reshape(-1, 3)


def np_ic_angle_moments(img, x, y, umax):
    I = img.astype(np.int64)
    m10 = m01 = 0
    for v in range(-15, 16):
        d = umax[abs(v)]
        for u in range(-d, d + 1):
            m10 += u * I[y + v, x + u]
            m01 += v * I[y + v, x + u]
    return m01, m10
