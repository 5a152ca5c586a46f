"""The 32 x 32 diagonal factor + inverse of the Cholesky chain (diag_factor_invert_nw, csrc/ba_cholesky.inc) with 1, 2 and 4 cooperating
waves: builds ba_solver.hip with -DORBHIP_CHOL_PROF into a scratch library, runs the three forms on random SPD blocks, compares the
inverses bit for bit (the forms must agree), against numpy, and prints the time per factor (s_memrealtime, 100 MHz)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
out = os.path.join(ROOT, "gpurun_out", "cholprof"); os.makedirs(out, exist_ok=True)
so = os.path.join(out, "liborbslam_hip_factor.so")
csrc = os.path.join(ROOT, "ceres_mono_orb_slam2_amd", "csrc")
srcs = [os.path.join(csrc, f) for f in ("ba_solver.hip", "capi_common.hip")]
so_st = os.path.join(out, "liborbslam_hip_factor_stamps.so")                       # the same with per-wave time stamps inside the factor (they cost ~0.3 us)
if not os.path.exists(so) or os.environ.get("REBUILD"):
    for target, extra in ((so, []), (so_st, ["-DORBHIP_DF_STAMP"])):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                               "-DORBHIP_CHOL_PROF"] + extra + ["-I", os.path.join(ROOT, "include"), "-shared", "-o", target] + srcs)
if "--build-only" in sys.argv:
    sys.exit(0)
L = C.CDLL(so)
L.ba_debug_factor_nw.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
L.ba_debug_df_stamps.argtypes = [C.c_void_p, C.c_int]
rng = np.random.default_rng(5)
bad = 0
best = {1: 1e9, 2: 1e9, 4: 1e9}
worst = 0.0
for trial in range(16):
    M = rng.standard_normal((32, 48))
    A = M @ M.T + (0.5 if trial % 3 else 1e-6) * np.eye(32)
    if trial >= 12:                                                             # badly scaled rows / columns (what Jacobi scaling leaves is milder)
        sc = 10.0 ** rng.uniform(-6, 6, 32); A = A * sc[:, None] * sc[None, :]
    if trial == 11: A[20, 20] = -1.0                                            # a non-positive pivot: every form must report it
    A = np.ascontiguousarray(A)
    n = 200
    X = {}; flag = {}; us = {}
    for nw in (1, 2, 4):
        Xn = np.zeros((32, 32)); t = np.zeros(2, np.uint64)
        assert L.ba_debug_factor_nw(A.ctypes.data, Xn.ctypes.data, n, nw, t.ctypes.data) == 0
        X[nw] = Xn; flag[nw] = int(t[1]); us[nw] = t[0] / 100.0 / n
        best[nw] = min(best[nw], us[nw])
    same = (X[1].tobytes() == X[2].tobytes() == X[4].tobytes()) if trial != 11 else (flag[1] == flag[2] == flag[4] == 1)
    if trial != 11:
        Lc = np.linalg.cholesky(A); ref = np.linalg.inv(Lc)
        err = float(np.abs(X[1] - ref).max() / np.abs(ref).max())
        resid = float(np.abs(X[1] @ A @ X[1].T - np.eye(32)).max())             # X A X^T = I
        rref = float(np.abs(ref @ A @ ref.T - np.eye(32)).max())
        upper = float(np.abs(np.triu(X[1], 1)).max())
        worst = max(worst, resid / max(rref, 1e-16))
    else:
        err = resid = rref = upper = float("nan")
    print("trial %2d: forms identical %s, bad flags %d / %d / %d, rel. error vs numpy %.1e, |X A X^T - I| %.1e (numpy's own inverse: %.1e), above the diagonal %.0e, "
          "%.2f / %.2f / %.2f us per factor" % (trial, same, flag[1], flag[2], flag[4], err, resid, rref, upper, us[1], us[2], us[4]))
    bad += not same
    if trial != 11 and (flag[1] or flag[2] or flag[4] or upper != 0.0): bad += 1
st = np.zeros(16, np.uint64)
L = C.CDLL(so_st)
L.ba_debug_factor_nw.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
L.ba_debug_df_stamps.argtypes = [C.c_void_p, C.c_int]
for nw in (1, 2, 4):                                                            # where a wave's time goes (one more run per form: the stamps cost ~0.3 us)
    L.ba_debug_df_stamps(None, 1)
    Xn = np.zeros((32, 32)); t = np.zeros(2, np.uint64)
    L.ba_debug_factor_nw(A.ctypes.data, Xn.ctypes.data, 200, nw, t.ctypes.data)
    L.ba_debug_df_stamps(st.ctypes.data, 0)
    s4 = st.reshape(4, 4).astype(np.float64)
    print("NW=%d (%.2f us with stamps): per wave [own block starts, ends, function ends] us after entry: " % (nw, t[0] / 100.0 / 200) +
          "  ".join("w%d %.2f %.2f %.2f" % (q, s4[q, 0] / max(s4[q, 3], 1) / 100, s4[q, 1] / max(s4[q, 3], 1) / 100, s4[q, 2] / max(s4[q, 3], 1) / 100) for q in range(nw)))
print("one wave %.2f us, two waves %.2f us, four waves %.2f us per factor; %d mismatches; worst residual %.1f x numpy's" % (best[1], best[2], best[4], bad, worst))
sys.exit(1 if bad else 0)
