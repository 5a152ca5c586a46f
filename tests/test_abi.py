"""CPU checks of the drop-in boundary: the in-tree C-ABI library builds, loads and exports every symbol
include/orbslam_hip.h declares; compute entry points fail LOUDLY without a GPU (no CPU fallback); the
reference-named C++ shims compile and link against it."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build_hip()
    from ceres_mono_orb_slam2_amd import _lib
    return _lib


def _declared_functions():
    txt = open(os.path.join(ROOT, "include", "orbslam_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|double|char\*|void)\s*\*?\s*([a-z_0-9]+)\s*\(", txt, flags=re.M)
    return sorted(set(n for n in names if n.startswith(("orbx_", "orbm_", "orbv_", "orbt_", "orbl_", "ba_", "orbhip_"))))


def test_header_symbols_are_exported(lib):
    L = lib.load()
    declared = _declared_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), "library does not export %s" % name
    assert sorted(lib.SYMBOLS) == declared, "ctypes SYMBOLS table out of sync with the header"
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH]).decode()
    exported = set(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert set(declared) <= exported


def test_landmark_merge_entry_points_validate_their_arguments(lib):
    """orbhip_comm_* / orbhip_allgather_landmarks (multi-GPU, csrc/orb_comm.hip): argument errors are reported before any device or RCCL
    call; without a GPU the communicator cannot be created (no CPU fallback)."""
    L = lib.load()
    h = C.c_void_p()
    ident = np.zeros(128, np.uint8)
    assert L.orbhip_comm_create(None, 1, 0, 0, C.byref(h)) == -1                       # ORBHIP_EINVAL
    assert L.orbhip_comm_create(lib.ptr(ident), 0, 0, 0, C.byref(h)) == -1
    assert L.orbhip_comm_create(lib.ptr(ident), 2, 2, 0, C.byref(h)) == -1
    assert L.orbhip_comm_create(lib.ptr(ident), 65, 0, 0, C.byref(h)) == -1
    assert L.orbhip_allgather_landmarks(None, None, None, 0, 1, None, None, 0, None, None, None) == -1
    assert L.orbhip_comm_info(None, None, None) == -1
    assert L.orbhip_comm_destroy(None) == 0
    if not os.path.exists("/dev/kfd"):
        assert L.orbhip_comm_create(lib.ptr(ident), 1, 0, 0, C.byref(h)) == -2               # ORBHIP_ENODEV
        assert b"no HIP device" in L.orbhip_last_error()


def test_struct_layouts(lib):
    from ceres_mono_orb_slam2_amd import KP_DTYPE
    assert KP_DTYPE.itemsize == 28                            # cv::KeyPoint
    assert C.sizeof(lib.BaSummary) == 40 and C.sizeof(lib.BaOptions) == 32


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present: the no-device error path cannot be exercised")
def test_compute_fails_loudly_without_gpu(lib):
    from ceres_mono_orb_slam2_amd import ORBextractor, ORBmatcher, optimizer
    from ceres_mono_orb_slam2_amd._lib import OrbHipError
    with pytest.raises(OrbHipError, match="no HIP device"):
        ORBextractor(1000, 1.2, 8, 20, 7)
    with pytest.raises(OrbHipError):
        ORBmatcher().hamming_best2(np.zeros((4, 32), np.uint8), np.zeros((4, 32), np.uint8))
    with pytest.raises(OrbHipError, match="no HIP device"):
        optimizer.pose_optimization(np.ones(4), np.array([0, 0, 0, 0, 0, 0, 1.0]), np.ones((5, 3)), np.ones((5, 2)), np.ones(5))
    # pure host helpers still work (no device needed, no oracle involved)
    a = np.arange(32, dtype=np.uint8); b = a[::-1].copy()
    assert ORBmatcher.DescriptorDistance(a, b) == int(np.unpackbits(a ^ b).sum())


def test_product_never_imports_oracle():
    """The shipped package and its native sources must not reference oracle/ (test infrastructure only)."""
    pkg = os.path.join(ROOT, "ceres_mono_orb_slam2_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".inc")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                assert "pyoracle" not in txt and "orc_" not in txt and "liborb_oracle" not in txt, os.path.join(dp, fn)
    # the BA legs of the benchmark (bench_ba.py at the repository root, not part of the package) touch the oracle in their
    # cpu_baseline leg only
    txt = open(os.path.join(ROOT, "bench_ba.py")).read()
    assert "if cpu:" in txt and "if a.cpu and a.oracle_lib:" in txt and txt.count("from oracle import") == 2


def test_compat_shims_compile_and_link(lib, tmp_path):
    exe = tmp_path / "test_compat"
    cmd = ["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "test_compat.cpp"), "-o", str(exe),
           lib.LIB_PATH, "-Wl,-rpath," + os.path.dirname(lib.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    assert exe.exists()


def test_dropin_test_program_and_reference_types_branch_compile(lib, tmp_path):
    """The drop-in call-site program links against the library (it runs on the GPU box: tests/test_gpu_compat_cpp.py), and the
    `#ifdef ORBSLAM_DROPIN_REFERENCE_TYPES` branch of orbslam_dropin.h - the one a maintainer of the reference uses - compiles
    with every member function instantiated (the reference's type NAMES bound to the mock data model: a spelling / type check
    of our header, not a build of the reference)."""
    inc = ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "cpp")]
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall"] + inc + [os.path.join(ROOT, "tests", "cpp", "test_reference_types_branch.cpp")])
    exe = tmp_path / "test_dropin"
    subprocess.check_call(["g++", "-std=c++17", "-O1"] + inc + [os.path.join(ROOT, "tests", "cpp", "test_dropin.cpp"), "-o", str(exe), lib.LIB_PATH, "-lpthread",
                           "-Wl,-rpath," + os.path.dirname(lib.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib"])
    assert exe.exists()
    # the `#ifdef ORBCOMPAT_HAVE_OPENCV` branch of orbslam_compat.h (cv::InputArray / cv::OutputArray signature of the reference's
    # ORBextractor::operator()) against the OpenCV API subset of tests/cpp/opencv_api_subset: compiled and linked here, run on
    # the GPU box (tests/test_gpu_compat_cpp.py::test_opencv_signature_branch)
    exe2 = tmp_path / "test_compat_opencv_branch"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "tests", "cpp", "opencv_api_subset"),
                           os.path.join(ROOT, "tests", "cpp", "test_compat_opencv_branch.cpp"), "-o", str(exe2), lib.LIB_PATH,
                           "-Wl,-rpath," + os.path.dirname(lib.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib"])
    assert exe2.exists()


def test_tracking_step_validates_arguments_and_fails_loudly_without_gpu(lib):
    import numpy as np
    from ceres_mono_orb_slam2_amd._lib import OrbHipError
    L = lib.load()
    res = (C.c_double * 16)()
    z = np.zeros(64, np.uint8)
    rc = L.orbt_track_with_motion_model(None, lib.ptr(z), 8, 8, 8, None, None, None, None, None, None, None, None, 0, C.c_float(15.0), 1, None, None, 0, None, None, None, res)
    assert rc != 0 and b"NULL" in L.orbhip_last_error()


def test_new_entry_points_fail_loudly_without_gpu_and_validate_arguments(lib):
    """The widened rows (Sim3, frame-side steps, vocabulary, triangulation, essential graph, batched BA) obey the same
    contract: argument errors are ORBHIP_EINVAL before any device work, valid calls without a device are ORBHIP_ENODEV."""
    import numpy as np
    from ceres_mono_orb_slam2_amd import frame, optimizer, synth
    from ceres_mono_orb_slam2_amd._lib import OrbHipError
    from ceres_mono_orb_slam2_amd.vocabulary import ORBVocabulary
    b = np.array([0, 640, 0, 480], np.float32)
    k = np.zeros((10, 4), np.float32)
    with pytest.raises(OrbHipError, match="no HIP device"):
        frame.AssignFeaturesToGrid(k, b)
    with pytest.raises(OrbHipError, match="no HIP device"):
        frame.GetFeaturesInArea(k, b, np.zeros((3, 2), np.float32), np.ones(3, np.float32))
    with pytest.raises(OrbHipError, match="no HIP device"):
        frame.UndistortKeyPoints(np.zeros((4, 2), np.float32), [500, 500, 320, 240], [0.1, 0, 0, 0, 0])
    assert np.array_equal(frame.UndistortKeyPoints(np.ones((4, 2), np.float32), [500, 500, 320, 240], [0, 0, 0, 0, 0]), np.ones((4, 2)))   # k1 == 0: host copy
    voc = synth.make_vocabulary(0, k=3, L=2)
    with pytest.raises(OrbHipError, match="no HIP device"):
        ORBVocabulary(*[voc[x] for x in ("node_desc", "child_off", "children", "word_id", "weight", "L")])
    bad = voc["children"].copy(); bad[0] = 10 ** 6
    with pytest.raises(OrbHipError, match="child index out of range"):
        ORBVocabulary(voc["node_desc"], voc["child_off"], bad, voc["word_id"], voc["weight"], voc["L"])
    x0 = np.zeros((3, 7)); S = np.tile([0, 0, 0, 1, 0, 0, 0.0], (2, 1))
    with pytest.raises(OrbHipError, match="edge vertex out of range"):
        optimizer.optimize_essential_graph(x0, [1, 0, 0], [0, 5], [1, 2], S)
    with pytest.raises(OrbHipError, match="no HIP device"):
        optimizer.optimize_essential_graph(x0, [1, 0, 0], [0, 1], [1, 2], S)
    p = synth.make_sim3_problem(0, n=20)
    with pytest.raises(OrbHipError, match="no HIP device"):
        optimizer.optimize_sim3(p["K1"], p["K2"], p["s12_0"], p["P3D2c"], p["obs1"], p["inv_sigma2_1"], p["P3D1c"], p["obs2"], p["inv_sigma2_2"])
    g = synth.make_ba_graph(1, ncam=3, npts=20, nobs=60, n_fixed=1)
    prob = (g["K4"], g["poses0"], g["cam_fixed"], np.ones(3, np.uint8), g["pts0"], g["obs_cam"], g["obs_pt"], g["obs_uv"], g["obs_inv_sigma2"])
    with pytest.raises(OrbHipError, match="no HIP device"):
        optimizer.local_bundle_adjustment_batch([prob, prob])
    badp = list(prob); badp[5] = g["obs_cam"].copy(); badp[5][0] = 99
    with pytest.raises(OrbHipError, match="observation index out of range"):
        optimizer.local_bundle_adjustment_batch([prob, tuple(badp)])
    with pytest.raises(OrbHipError, match="octave out of range"):
        frame.TriangulateMatches(np.eye(3, 4), np.eye(3, 4), [500, 500, 320, 240], [500, 500, 320, 240], [[1, 1, 9]], [[1, 1, 0]], np.ones(8), np.ones(8), 1.8)
