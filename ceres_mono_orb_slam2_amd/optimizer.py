"""Host mirror of CeresOptimizer's flattened entry points (reference include/CeresOptimizer.h:351-376,
src/CeresOptimizer.cc:49-599) over the HIP C ABI.  All arithmetic fp64; poses are 7-vectors
[tx,ty,tz,qx,qy,qz,qw] (src/MatEigenConverter.cc:66-85)."""
import ctypes as C

import numpy as np

from . import _lib

HUBER_DELTA = float(np.sqrt(5.991))      # ceres::HuberLoss(sqrt(5.991)), src/CeresOptimizer.cc:81,296,421
CHI2_THRESHOLD = 5.991                   # :257, :553


def _f64(a):
    return np.ascontiguousarray(a, np.float64)


def pose_optimization(K4, pose7, Xw, uv, inv_sigma2):
    """CeresOptimizer::PoseOptimization (src/CeresOptimizer.cc:275-342).
    Returns (n_inliers, pose7, outlier_flags, summary)."""
    L = _lib.load()
    K4 = _f64(K4); pose = _f64(pose7).copy(); Xw = _f64(Xw).reshape(-1, 3); uv = _f64(uv).reshape(-1, 2)
    isg = np.ascontiguousarray(inv_sigma2, np.float32)
    n = len(Xw)
    out = np.zeros(max(n, 1), np.uint8)
    ninl = C.c_int(0); s = _lib.BaSummary()
    _lib.check(L.ba_pose_optimization(_lib.ptr(K4), _lib.ptr(pose), _lib.ptr(Xw), _lib.ptr(uv), _lib.ptr(isg), n,
                                      _lib.ptr(out), C.byref(ninl), C.byref(s)), "ba_pose_optimization")
    return ninl.value, pose, out[:n], s.as_dict()


def pose_optimization_batch(d_K4, d_pose7, d_Xw, d_uv, d_inv_sigma2, d_offsets, stream=None):
    """Batched device-resident PoseOptimization (torch CUDA tensors); d_pose7 is updated in place.
    Returns (outlier[total] uint8, n_inliers[np] int32, summaries[np, 6] float64-packed bytes)."""
    import torch
    L = _lib.load()
    npb = d_pose7.shape[0]
    total = d_Xw.shape[0]
    outl = torch.empty((max(total, 1),), dtype=torch.uint8, device=d_pose7.device)
    ninl = torch.empty((npb,), dtype=torch.int32, device=d_pose7.device)
    summ = torch.empty((npb, C.sizeof(_lib.BaSummary)), dtype=torch.uint8, device=d_pose7.device)
    st = torch.cuda.current_stream(d_pose7.device).cuda_stream if stream is None else stream
    _lib.check(L.ba_pose_optimization_batch_device(_lib.ptr(d_K4), _lib.ptr(d_pose7), _lib.ptr(d_Xw), _lib.ptr(d_uv),
                                                   _lib.ptr(d_inv_sigma2), _lib.ptr(d_offsets), npb, _lib.ptr(outl),
                                                   _lib.ptr(ninl), _lib.ptr(summ), C.c_void_p(st)),
               "ba_pose_optimization_batch_device")
    return outl[:total], ninl, summ


def bundle_adjustment(K4, poses7, cam_fixed, pts3, obs_cam, obs_pt, obs_uv, obs_weight, obs_robust, n_iterations=200,
                      huber_delta=HUBER_DELTA, fix_points=False, stop_flag=None):
    """CeresOptimizer::BundleAdjustment (src/CeresOptimizer.cc:59-225) on flattened arrays.
    stop_flag: optional 1-element uint8 numpy array polled between iterations.
    Returns (poses7, pts3, summary)."""
    L = _lib.load()
    K4 = _f64(K4).reshape(-1, 4); poses = _f64(poses7).copy(); pts = _f64(pts3).reshape(-1, 3).copy()
    cf = np.ascontiguousarray(cam_fixed, np.uint8)
    oc = np.ascontiguousarray(obs_cam, np.int32); op = np.ascontiguousarray(obs_pt, np.int32)
    uv = _f64(obs_uv).reshape(-1, 2); w = _f64(obs_weight); rb = np.ascontiguousarray(obs_robust, np.uint8)
    o = _lib.BaOptions(int(n_iterations), float(huber_delta), int(fix_points),
                       _lib.ptr(stop_flag) if stop_flag is not None else None)
    s = _lib.BaSummary()
    _lib.check(L.ba_solve(_lib.ptr(K4), _lib.ptr(poses), _lib.ptr(cf), len(cf), _lib.ptr(pts), len(pts), _lib.ptr(oc),
                          _lib.ptr(op), _lib.ptr(uv), _lib.ptr(w), _lib.ptr(rb), len(oc), C.byref(o), C.byref(s)),
               "ba_solve")
    return poses, pts, s.as_dict()


def global_bundle_adjustment(K4, poses7, cam_fixed, pts3, obs_cam, obs_pt, obs_uv, obs_inv_sigma2, n_iterations=200,
                             stop_flag=None, is_robust=True):
    """CeresOptimizer::GlobalBundleAdjustemnt (src/CeresOptimizer.cc:49-57): every observation weighted by
    invSigma2 (F7), Huber iff is_robust.  Returns (poses7 with NORMALISED quaternions, pts3, summary)."""
    n = len(obs_cam)
    poses, pts, s = bundle_adjustment(K4, poses7, cam_fixed, pts3, obs_cam, obs_pt, obs_uv,
                                      np.asarray(obs_inv_sigma2, np.float32).astype(np.float64),
                                      np.full(n, 1 if is_robust else 0, np.uint8), n_iterations, stop_flag=stop_flag)
    poses[:, 3:] /= np.linalg.norm(poses[:, 3:], axis=1, keepdims=True)      # Matrix_7_1_ToMatrix4d (:198)
    return poses, pts, s


def check_outlier(K4, pose7, Xw, uv, inv_sigma2, thres=CHI2_THRESHOLD):
    """CeresOptimizer::CheckOutlier (src/CeresOptimizer.cc:227-241). Returns (is_outlier, depth)."""
    L = _lib.load()
    d = C.c_double()
    r = L.ba_check_outlier(_lib.ptr(_f64(K4)), _lib.ptr(_f64(pose7)), _lib.ptr(_f64(Xw)), _lib.ptr(_f64(uv)),
                           float(inv_sigma2), float(thres), C.byref(d))
    return bool(r), d.value


def local_bundle_adjustment(K4, poses7, cam_fixed, cam_local, pts3, obs_cam, obs_pt, obs_uv, obs_inv_sigma2,
                            stop_flag=None, duplicate_blocks=True):
    """Optimisation core of CeresOptimizer::LocalBundleAdjustment (src/CeresOptimizer.cc:408-598).
    Returns (aborted, poses7, pts3, obs_erase, summary_pass1, summary_pass2)."""
    L = _lib.load()
    K4 = _f64(K4).reshape(-1, 4); poses = _f64(poses7).copy(); pts = _f64(pts3).reshape(-1, 3).copy()
    cf = np.ascontiguousarray(cam_fixed, np.uint8); cl = np.ascontiguousarray(cam_local, np.uint8)
    oc = np.ascontiguousarray(obs_cam, np.int32); op = np.ascontiguousarray(obs_pt, np.int32)
    uv = _f64(obs_uv).reshape(-1, 2); isg = np.ascontiguousarray(obs_inv_sigma2, np.float32)
    er = np.zeros(max(len(oc), 1), np.uint8)
    ab = C.c_int(0); s1 = _lib.BaSummary(); s2 = _lib.BaSummary()
    _lib.check(L.ba_local_bundle_adjustment(_lib.ptr(K4), _lib.ptr(poses), _lib.ptr(cf), _lib.ptr(cl), len(cf),
                                            _lib.ptr(pts), len(pts), _lib.ptr(oc), _lib.ptr(op), _lib.ptr(uv),
                                            _lib.ptr(isg), len(oc), _lib.ptr(stop_flag) if stop_flag is not None else None,
                                            int(duplicate_blocks), _lib.ptr(er), C.byref(ab), C.byref(s1), C.byref(s2)),
               "ba_local_bundle_adjustment")
    return ab.value, poses, pts, er[:len(oc)], s1.as_dict(), s2.as_dict()


def optimize_sim3(K1, K2, s12, P3D2c, obs1, inv_sigma2_1, P3D1c, obs2, inv_sigma2_2, th2=10.0, fix_scale=False):
    """CeresOptimizer::OptimizeSim3 (src/CeresOptimizer.cc:601-735) on flattened correspondences.
    s12 = Sophus::Sim3d::data() layout [qx,qy,qz,qw (|q|^2 = scale), tx,ty,tz]; th2 = 10 at the only call site
    (src/LoopClosing.cc:324).  fix_scale is ignored exactly as the reference ignores bFixScale.
    Returns (n_inliers, s12, outlier_flags, summary)."""
    L = _lib.load()
    K1 = _f64(K1); K2 = _f64(K2); S = _f64(s12).copy()
    P2 = _f64(P3D2c).reshape(-1, 3); o1 = _f64(obs1).reshape(-1, 2); w1 = np.ascontiguousarray(inv_sigma2_1, np.float32)
    P1 = _f64(P3D1c).reshape(-1, 3); o2 = _f64(obs2).reshape(-1, 2); w2 = np.ascontiguousarray(inv_sigma2_2, np.float32)
    n = len(P2)
    assert len(P1) == n and len(o1) == n and len(o2) == n and len(w1) == n and len(w2) == n
    out = np.zeros(max(n, 1), np.uint8)
    ninl = C.c_int(0); s = _lib.BaSummary()
    _lib.check(L.ba_optimize_sim3(_lib.ptr(K1), _lib.ptr(K2), _lib.ptr(S), _lib.ptr(P2), _lib.ptr(o1), _lib.ptr(w1), _lib.ptr(P1),
                                  _lib.ptr(o2), _lib.ptr(w2), n, float(th2), int(bool(fix_scale)), _lib.ptr(out), C.byref(ninl),
                                  C.byref(s)), "ba_optimize_sim3")
    return ninl.value, S, out[:n], s.as_dict()


def optimize_sim3_batch(d_K1, d_K2, d_s12, d_P3D2c, d_obs1, d_w1, d_P3D1c, d_obs2, d_w2, d_offsets, d_th2, stream=None):
    """Batched device-resident OptimizeSim3 (torch CUDA tensors, one loop candidate per problem); d_s12 is updated in
    place.  Returns (outlier[total] uint8, n_inliers[np] int32, summaries[np, sizeof(ba_summary)] bytes)."""
    import torch
    L = _lib.load()
    npb = d_s12.shape[0]
    total = d_P3D2c.shape[0]
    dev = d_s12.device
    outl = torch.empty((max(total, 1),), dtype=torch.uint8, device=dev)
    ninl = torch.empty((npb,), dtype=torch.int32, device=dev)
    summ = torch.empty((npb, C.sizeof(_lib.BaSummary)), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
    _lib.check(L.ba_optimize_sim3_batch_device(_lib.ptr(d_K1), _lib.ptr(d_K2), _lib.ptr(d_s12), _lib.ptr(d_P3D2c), _lib.ptr(d_obs1),
                                               _lib.ptr(d_w1), _lib.ptr(d_P3D1c), _lib.ptr(d_obs2), _lib.ptr(d_w2), _lib.ptr(d_offsets),
                                               _lib.ptr(d_th2), npb, _lib.ptr(outl), _lib.ptr(ninl), _lib.ptr(summ), C.c_void_p(st)),
               "ba_optimize_sim3_batch_device")
    return outl[:total], ninl, summ


def sim3_exp(tangent7):
    """Sophus::Sim3d::exp in the qt7 layout (host arithmetic of the library)."""
    L = _lib.load()
    a = _f64(tangent7); out = np.zeros(7)
    _lib.check(L.ba_sim3_exp(_lib.ptr(a), _lib.ptr(out)), "ba_sim3_exp")
    return out


def sim3_log(s12):
    L = _lib.load()
    a = _f64(s12); out = np.zeros(7)
    _lib.check(L.ba_sim3_log(_lib.ptr(a), _lib.ptr(out)), "ba_sim3_log")
    return out


def matrix4d_to_pose7(T):
    """MatEigenConverter::Matrix4dToMatrix_7_1 (src/MatEigenConverter.cc:66-75): 4x4 [R t; 0 1] -> [t, qx,qy,qz,qw]."""
    L = _lib.load()
    a = _f64(T).reshape(4, 4); out = np.zeros(7)
    _lib.check(L.ba_matrix4d_to_pose7(_lib.ptr(a), _lib.ptr(out)), "ba_matrix4d_to_pose7")
    return out


def pose7_to_matrix4d(pose7):
    """MatEigenConverter::Matrix_7_1_ToMatrix4d (src/MatEigenConverter.cc:77-85): normalises the quaternion."""
    L = _lib.load()
    a = _f64(pose7); out = np.zeros((4, 4))
    _lib.check(L.ba_pose7_to_matrix4d(_lib.ptr(a), _lib.ptr(out)), "ba_pose7_to_matrix4d")
    return out


def set_profiling(enable=True):
    _lib.check(_lib.load().ba_set_profiling(int(enable)))


def get_profile():
    """(device ms, problems solved, LM iterations) of THIS host thread's solves since the last call."""
    ms, n, it = C.c_double(), C.c_int(), C.c_int()
    _lib.check(_lib.load().ba_get_profile(C.byref(ms), C.byref(n), C.byref(it)))
    return ms.value, n.value, it.value


LA_FORMS = {0: "none", 1: "k_chol_persist", 2: "k_chol_wg", 3: "k_chol_la steps"}
TL_FORMS = {0: "none", 1: "k_chol_persist_blk", 2: "hybrid step kernels", 3: "classic", 4: "one launch"}


def get_last_plan():
    """Which member of the factorisation family THIS host thread's last solve took (ba_get_last_plan)."""
    v = (C.c_int32 * 12)()
    _lib.check(_lib.load().ba_get_last_plan(v))
    return {"problems": v[0], "lookahead_form": LA_FORMS.get(v[1], v[1]), "two_level_form": TL_FORMS.get(v[2], v[2]),
            "backward_substitution": {1: "k_chol_bsolve_sky", 2: "per super-block"}.get(v[3], v[3]), "npad_lookahead": v[4], "npad_two_level": v[5],
            "band_tiles": v[6], "persist_workgroups": v[7], "persist_mode": v[8], "lookahead_above_1024": bool(v[9]), "fits_k_chol_wg": bool(v[10])}


def _addr(a):
    return a.ctypes.data if a is not None and a.size else None


def bundle_adjustment_batch(problems, n_iterations=200, huber_delta=HUBER_DELTA, fix_points=False, stop_flag=None):
    """ba_solve_batch: independent problems solved in lockstep (one grid row per problem).  `problems` is a list of
    tuples (K4, poses7, cam_fixed, pts3, obs_cam, obs_pt, obs_uv, obs_weight, obs_robust), as bundle_adjustment takes them.
    Returns a list of (poses7, pts3, summary)."""
    L = _lib.load()
    keep, arr = [], (_lib.BaProblem * len(problems))()
    for q, (K4, poses7, cam_fixed, pts3, obs_cam, obs_pt, obs_uv, obs_weight, obs_robust) in enumerate(problems):
        K4 = _f64(K4).reshape(-1, 4); poses = _f64(poses7).copy(); pts = _f64(pts3).reshape(-1, 3).copy()
        cf = np.ascontiguousarray(cam_fixed, np.uint8)
        oc = np.ascontiguousarray(obs_cam, np.int32); op = np.ascontiguousarray(obs_pt, np.int32)
        uv = _f64(obs_uv).reshape(-1, 2); w = _f64(obs_weight); rb = np.ascontiguousarray(obs_robust, np.uint8)
        keep.append((K4, poses, cf, pts, oc, op, uv, w, rb))
        arr[q] = _lib.BaProblem(_addr(K4), _addr(poses), _addr(cf), len(cf), _addr(pts), len(pts), _addr(oc), _addr(op), _addr(uv),
                                _addr(w), _addr(rb), len(oc))
    o = _lib.BaOptions(int(n_iterations), float(huber_delta), int(fix_points), _addr(stop_flag) if stop_flag is not None else None)
    summ = (_lib.BaSummary * len(problems))()
    _lib.check(L.ba_solve_batch(C.cast(arr, C.c_void_p), len(problems), C.byref(o), C.cast(summ, C.c_void_p)), "ba_solve_batch")
    return [(k[1], k[3], summ[q].as_dict()) for q, k in enumerate(keep)]


def local_bundle_adjustment_batch(problems, stop_flag=None, duplicate_blocks=True):
    """ba_local_bundle_adjustment_batch: `problems` is a list of tuples (K4, poses7, cam_fixed, cam_local, pts3, obs_cam,
    obs_pt, obs_uv, obs_inv_sigma2), as local_bundle_adjustment takes them.
    Returns (aborted, [(poses7, pts3, obs_erase, summary_pass1, summary_pass2), ...])."""
    L = _lib.load()
    keep, arr = [], (_lib.BaLocalProblem * len(problems))()
    for q, (K4, poses7, cam_fixed, cam_local, pts3, obs_cam, obs_pt, obs_uv, obs_inv_sigma2) in enumerate(problems):
        K4 = _f64(K4).reshape(-1, 4); poses = _f64(poses7).copy(); pts = _f64(pts3).reshape(-1, 3).copy()
        cf = np.ascontiguousarray(cam_fixed, np.uint8); cl = np.ascontiguousarray(cam_local, np.uint8)
        oc = np.ascontiguousarray(obs_cam, np.int32); op = np.ascontiguousarray(obs_pt, np.int32)
        uv = _f64(obs_uv).reshape(-1, 2); isg = np.ascontiguousarray(obs_inv_sigma2, np.float32)
        er = np.zeros(max(len(oc), 1), np.uint8)
        keep.append((K4, poses, cf, cl, pts, oc, op, uv, isg, er))
        arr[q] = _lib.BaLocalProblem(_addr(K4), _addr(poses), _addr(cf), _addr(cl), len(cf), _addr(pts), len(pts), _addr(oc), _addr(op),
                                     _addr(uv), _addr(isg), len(oc), er.ctypes.data)
    ab = C.c_int(0)
    s1 = (_lib.BaSummary * len(problems))(); s2 = (_lib.BaSummary * len(problems))()
    _lib.check(L.ba_local_bundle_adjustment_batch(C.cast(arr, C.c_void_p), len(problems), _lib.ptr(stop_flag) if stop_flag is not None else None,
                                                  int(duplicate_blocks), C.byref(ab), C.cast(s1, C.c_void_p), C.cast(s2, C.c_void_p)),
               "ba_local_bundle_adjustment_batch")
    return ab.value, [(k[1], k[4], k[9][:len(k[5])], s1[q].as_dict(), s2[q].as_dict()) for q, k in enumerate(keep)]


def optimize_essential_graph(lie7, kf_fixed, edge_j, edge_i, edge_Sji, max_iterations=100, stop_flag=None):
    """CeresOptimizer::OptimizeEssentialGraph, the solve (src/CeresOptimizer.cc:737-914) on flattened vertices / edges.
    lie7 = Scw.log() per keyframe; edges (j, i, Sji qt7) in the reference's insertion order.  Returns (lie7, summary)."""
    L = _lib.load()
    x = _f64(lie7).reshape(-1, 7).copy(); fx = np.ascontiguousarray(kf_fixed, np.uint8)
    ej = np.ascontiguousarray(edge_j, np.int32); ei = np.ascontiguousarray(edge_i, np.int32); S = _f64(edge_Sji).reshape(-1, 7)
    s = _lib.BaSummary()
    _lib.check(L.ba_optimize_essential_graph(_lib.ptr(x), _lib.ptr(fx), len(x), _lib.ptr(ej), _lib.ptr(ei), _lib.ptr(S), len(ej),
                                             int(max_iterations), _lib.ptr(stop_flag) if stop_flag is not None else None, C.byref(s)),
               "ba_optimize_essential_graph")
    return x, s.as_dict()


def essential_graph_correct(lie7_orig, lie7_opt, pt_ref_kf, pts3):
    """Write-back arithmetic of OptimizeEssentialGraph (:916-956).  Returns (Tiw [n, 3, 4], corrected points)."""
    L = _lib.load()
    a = _f64(lie7_orig).reshape(-1, 7); b = _f64(lie7_opt).reshape(-1, 7); n = len(a)
    T = np.zeros((n, 12)); pr = np.ascontiguousarray(pt_ref_kf, np.int32); P = _f64(pts3).reshape(-1, 3).copy()
    _lib.check(L.ba_essential_graph_correct(_lib.ptr(a), _lib.ptr(b), n, _lib.ptr(T), _lib.ptr(pr), _lib.ptr(P), len(P)),
               "ba_essential_graph_correct")
    return T.reshape(n, 3, 4), P
