"""ctypes loader for the in-tree HIP C-ABI library (include/orbslam_hip.h).

There is no CPU fallback: if the library is missing or a call fails the product raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORBHIP_LIB") or os.path.join(_HERE, "lib", "liborbslam_hip.so")   # (ORBHIP_LIB: a profiling / experiment build)

# every symbol include/orbslam_hip.h declares (tests check that the library exports them all)
SYMBOLS = [
    "orbhip_last_error", "orbhip_device_count", "orbhip_version", "orbhip_set_default_device", "orbhip_get_default_device", "orbhip_set_thread_priority", "orbhip_copy_pinned_async", "orbl_create_new_map_points", "orbl_fuse_batch", "orbt_relocalization_search_by_bow",
    "orbx_create", "orbx_destroy", "orbx_get_levels", "orbx_set_opencv_variant", "orbx_get_tables", "orbx_max_keypoints", "orbx_extract",
    "orbx_extract_batch_device", "orbx_set_profiling", "orbx_get_stage_ms", "orbx_get_level_image", "orbx_get_level_candidates", "orbx_get_level_selected",
    "orbm_descriptor_distance", "orbm_hamming_best2_device", "orbm_hamming_best2", "orbm_match_frames_batch_device",
    "orbm_search_for_initialization", "orbm_search_by_projection", "orbm_search_by_sim3", "orbm_search_by_bow", "orbm_search_for_triangulation",
    "orbv_create", "orbv_destroy", "orbv_load_text", "orbv_parse_text", "orbv_free_parsed", "orbv_transform", "orbv_descend_device", "orbv_score_l1",
    "orbm_undistort_keypoints", "orbm_assign_features_to_grid", "orbm_features_in_area", "orbm_is_in_frustum", "orbm_is_in_frustum_gates",
    "orbm_triangulate_matches", "orbt_track_with_motion_model", "orbt_track_local_map", "orbt_track_reference_keyframe", "orbt_last_call_ms",
    "ba_pose_optimization", "ba_pose_optimization_batch_device", "ba_solve", "ba_check_outlier",
    "ba_local_bundle_adjustment", "ba_optimize_sim3", "ba_optimize_sim3_batch_device", "ba_sim3_exp", "ba_sim3_log", "ba_sim3_mul", "ba_sim3_inverse",
    "ba_solve_batch", "ba_local_bundle_adjustment_batch", "ba_optimize_essential_graph", "ba_essential_graph_correct",
    "ba_matrix4d_to_pose7", "ba_pose7_to_matrix4d", "ba_set_profiling", "ba_get_profile", "ba_get_last_plan", "ba_set_wait_limit_ms",
    "orbl_fuse_batch_sim3", "orbhip_comm_get_unique_id", "orbhip_comm_create", "orbhip_comm_adopt", "orbhip_comm_info", "orbhip_comm_destroy", "orbhip_allgather_landmarks",
]


class BaProblem(C.Structure):                 # ba_problem
    _fields_ = [("K4", C.c_void_p), ("poses7", C.c_void_p), ("cam_fixed", C.c_void_p), ("ncam", C.c_int32),
                ("pts3", C.c_void_p), ("npts", C.c_int32), ("obs_cam", C.c_void_p), ("obs_pt", C.c_void_p),
                ("obs_uv", C.c_void_p), ("obs_weight", C.c_void_p), ("obs_robust", C.c_void_p), ("nobs", C.c_int32)]


class BaLocalProblem(C.Structure):            # ba_local_problem
    _fields_ = [("K4", C.c_void_p), ("poses7", C.c_void_p), ("cam_fixed", C.c_void_p), ("cam_local", C.c_void_p),
                ("ncam", C.c_int32), ("pts3", C.c_void_p), ("npts", C.c_int32), ("obs_cam", C.c_void_p),
                ("obs_pt", C.c_void_p), ("obs_uv", C.c_void_p), ("obs_inv_sigma2", C.c_void_p), ("nobs", C.c_int32),
                ("obs_erase", C.c_void_p)]


class OrbHipError(RuntimeError):
    pass


class BaOptions(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("huber_delta", C.c_double), ("fix_points", C.c_int32),
                ("stop_flag", C.c_void_p)]


class BaSummary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("iterations", C.c_int32),
                ("successful_steps", C.c_int32), ("termination", C.c_int32), ("final_radius", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def load():
    """Load liborbslam_hip.so (built by __graft_entry__.build()); raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OrbHipError("HIP library %s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % LIB_PATH)
    # PyTorch (device memory / streams / torch.distributed plumbing) bundles its own libamdhip64.so.7;
    # load it FIRST so this library binds to the same HIP runtime instead of a second copy from
    # /opt/rocm (two runtimes in one process cannot both own the GPU).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, f32, f64, sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t
    L.orbhip_last_error.restype = C.c_char_p
    L.orbhip_version.restype = C.c_char_p
    L.orbx_create.argtypes = [i32, f32, i32, i32, i32, i32, C.POINTER(vp)]
    L.orbx_destroy.argtypes = [vp]
    L.orbx_get_levels.argtypes = [vp]
    L.orbx_get_tables.argtypes = [vp, vp, vp, vp, vp, vp]
    L.orbx_max_keypoints.argtypes = [vp]
    L.orbx_extract.argtypes = [vp, vp, i32, i32, i32, vp, vp, i32, C.POINTER(i32)]
    L.orbx_extract_batch_device.argtypes = [vp, vp, i32, i32, i32, sz, i32, vp, vp, i32, vp, vp]
    L.orbx_set_profiling.argtypes = [vp, i32]
    L.orbx_get_stage_ms.argtypes = [vp, vp, C.POINTER(i32)]
    L.orbx_get_level_image.argtypes = [vp, i32, i32, i32, vp, C.POINTER(i32), C.POINTER(i32)]
    L.orbx_get_level_candidates.argtypes = [vp, i32, i32, vp, i32, C.POINTER(i32)]
    L.orbx_get_level_selected.argtypes = [vp, i32, i32, vp, i32, C.POINTER(i32)]
    L.orbm_descriptor_distance.argtypes = [vp, vp]
    L.orbm_hamming_best2_device.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp]
    L.orbm_hamming_best2.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, vp]
    L.orbm_match_frames_batch_device.argtypes = [vp, vp, vp, i32, vp, vp, i32, f32, i32, i32, vp, vp, vp]
    L.orbm_search_for_initialization.argtypes = [vp, vp, i32, vp, vp, i32, vp, vp, i32, f32, i32, vp, C.POINTER(i32)]
    L.orbm_search_by_projection.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, f32, vp, i32, f32, i32, i32,
                                            vp, vp, C.POINTER(i32)]
    L.orbm_search_by_sim3.argtypes = [vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(i32)]
    L.orbm_search_by_bow.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, f32, i32, i32, i32, vp,
                                     C.POINTER(i32)]
    L.orbm_search_for_triangulation.argtypes = [vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, i32, vp, f32, f32, vp, vp,
                                                i32, vp, C.POINTER(i32)]
    L.orbv_create.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, C.POINTER(vp)]
    L.orbv_destroy.argtypes = [vp]
    L.orbv_load_text.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
    L.orbv_parse_text.argtypes = [C.c_char_p] + [C.POINTER(i32)] * 6 + [C.POINTER(vp)] * 5
    L.orbv_free_parsed.argtypes = [vp]
    L.orbv_free_parsed.restype = None
    L.orbv_transform.argtypes = [vp, vp, i32, i32, vp, vp, C.POINTER(i32), vp, vp, vp, C.POINTER(i32)]
    L.orbv_descend_device.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp]
    L.orbv_score_l1.argtypes = [vp, vp, i32, vp, vp, i32]
    L.orbv_score_l1.restype = f64
    L.orbm_undistort_keypoints.argtypes = [vp, i32, vp, vp, vp]
    L.orbm_assign_features_to_grid.argtypes = [vp, i32, vp, vp, vp, C.POINTER(i32)]
    L.orbm_features_in_area.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, i32, C.POINTER(i32)]
    L.orbm_triangulate_matches.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, f32, vp, vp]
    L.orbm_is_in_frustum.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, i32, vp, vp, vp, vp]
    L.orbt_last_call_ms.argtypes = []; L.orbt_last_call_ms.restype = C.c_double
    L.orbt_track_with_motion_model.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, i32, vp, vp, i32, vp, vp, vp, vp]
    L.orbx_set_opencv_variant.argtypes = [vp, i32]
    L.orbt_track_reference_keyframe.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, f32, i32, vp, vp, i32, vp, vp, C.POINTER(i32), vp, vp, vp,
                                                C.POINTER(i32), vp, vp, vp, vp]
    L.orbt_track_local_map.argtypes = [vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, f32, f32, vp, vp, vp, vp, vp]
    L.orbm_is_in_frustum_gates.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, vp, vp, vp, vp]
    L.ba_pose_optimization.argtypes = [vp, vp, vp, vp, vp, i32, vp, C.POINTER(i32), C.POINTER(BaSummary)]
    L.ba_pose_optimization_batch_device.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp]
    L.ba_solve.argtypes = [vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, C.POINTER(BaOptions),
                           C.POINTER(BaSummary)]
    L.ba_check_outlier.argtypes = [vp, vp, vp, vp, f64, f64, vp]
    L.ba_local_bundle_adjustment.argtypes = [vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, i32, vp, i32, vp,
                                             C.POINTER(i32), C.POINTER(BaSummary), C.POINTER(BaSummary)]
    L.ba_optimize_sim3.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, f64, i32, vp, C.POINTER(i32), C.POINTER(BaSummary)]
    L.ba_optimize_sim3_batch_device.argtypes = [vp] * 11 + [i32, vp, vp, vp, vp]
    L.ba_sim3_exp.argtypes = [vp, vp]
    L.ba_sim3_log.argtypes = [vp, vp]
    L.ba_sim3_mul.argtypes = [vp, vp, vp]; L.ba_sim3_inverse.argtypes = [vp, vp]
    L.ba_optimize_essential_graph.argtypes = [vp, vp, i32, vp, vp, vp, i32, i32, vp, C.POINTER(BaSummary)]
    L.ba_essential_graph_correct.argtypes = [vp, vp, i32, vp, vp, vp, i32]
    L.ba_matrix4d_to_pose7.argtypes = [vp, vp]
    L.ba_pose7_to_matrix4d.argtypes = [vp, vp]
    L.ba_set_profiling.argtypes = [i32]
    L.ba_get_profile.argtypes = [C.POINTER(f64), C.POINTER(i32), C.POINTER(i32)]
    L.ba_get_last_plan.argtypes = [C.POINTER(i32)]
    L.orbhip_comm_get_unique_id.argtypes = [vp]
    L.orbhip_comm_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    L.orbhip_comm_adopt.argtypes = [vp, i32, C.POINTER(vp)]
    L.orbhip_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.orbhip_comm_destroy.argtypes = [vp]
    L.orbhip_allgather_landmarks.argtypes = [vp, vp, vp, i32, i32, vp, vp, i32, vp, C.POINTER(i32), vp]
    L.ba_set_wait_limit_ms.argtypes = [C.c_double]
    L.orbhip_copy_pinned_async.argtypes = [vp, vp, C.c_size_t, vp]
    L.ba_solve_batch.argtypes = [vp, i32, C.POINTER(BaOptions), vp]
    L.ba_local_bundle_adjustment_batch.argtypes = [vp, i32, vp, i32, C.POINTER(i32), vp, vp]
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        msg = load().orbhip_last_error().decode("utf-8", "replace")
        raise OrbHipError("%s failed (code %d): %s" % (what or "HIP call", rc, msg))


def ptr(a):
    """host numpy array or torch tensor -> void*"""
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return a.ctypes.data_as(C.c_void_p)
