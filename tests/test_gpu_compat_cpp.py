"""GPU: the reference-named C++ shims (ORB_SLAM2::ORBextractor / ORBmatcher in csrc/compat) driven from a
C++ program must reproduce the oracle bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from ceres_mono_orb_slam2_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_shims_vs_oracle(oracle, tmp_path):
    from ceres_mono_orb_slam2_amd import _lib
    exe = tmp_path / "test_compat"
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "test_compat.cpp"), "-o", str(exe),
                           _lib.LIB_PATH, "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib"])
    img = synth.make_frame(77, 640, 480, "blocks")
    raw = tmp_path / "img.raw"; out = tmp_path / "out.bin"
    img.tofile(raw)
    pp = synth.make_pose_problem(5, n=400)
    posef = tmp_path / "pose.bin"
    with open(posef, "wb") as f:
        f.write(np.int32(400).tobytes())
        for k, dt in (("K4", np.float64), ("pose0", np.float64), ("Xw", np.float64), ("uv", np.float64), ("inv_sigma2", np.float32)):
            f.write(np.ascontiguousarray(pp[k], dt).tobytes())
    sp = synth.make_sim3_problem(6, n=150, scale=1.0)
    sim3f = tmp_path / "sim3.bin"
    SK = (("K1", np.float64), ("K2", np.float64), ("s12_0", np.float64), ("P3D2c", np.float64), ("obs1", np.float64), ("P3D1c", np.float64),
          ("obs2", np.float64), ("inv_sigma2_1", np.float32), ("inv_sigma2_2", np.float32))
    with open(sim3f, "wb") as f:
        f.write(np.int32(150).tobytes())
        for k, dt in SK:
            f.write(np.ascontiguousarray(sp[k], dt).tobytes())
    from tests.test_oracle_essential_graph import build_problem
    rng = np.random.default_rng(3)
    nq = 64
    qxy = np.stack([rng.uniform(0, 640, nq), rng.uniform(0, 480, nq)], 1).astype(np.float32); qr = rng.uniform(5, 40, nq).astype(np.float32)
    qmn = rng.integers(-1, 3, nq).astype(np.int32); qmx = (qmn + rng.integers(0, 4, nq)).astype(np.int32)
    voc = synth.make_vocabulary(5, k=6, L=3)
    eg = synth.make_essential_graph(9, n=25)
    x0, ej, ei, Sji = build_problem(eg)
    auxf = tmp_path / "aux.bin"
    with open(auxf, "wb") as f:
        f.write(np.int32(nq).tobytes()); f.write(qxy.tobytes()); f.write(qr.tobytes()); f.write(qmn.tobytes()); f.write(qmx.tobytes())
        f.write(np.array([len(voc["word_id"]), voc["L"], len(voc["children"])], np.int32).tobytes())
        for k in ("node_desc", "child_off", "children", "word_id", "weight"):
            f.write(np.ascontiguousarray(voc[k]).tobytes())
        f.write(np.array([len(x0), len(ej)], np.int32).tobytes())
        f.write(x0.tobytes()); f.write(eg["fixed"].tobytes()); f.write(ej.tobytes()); f.write(ei.tobytes()); f.write(Sji.tobytes())
    subprocess.check_call([str(exe), str(raw), "640", "480", "1000", str(out), str(posef), str(sim3f), str(auxf)])
    buf = open(out, "rb").read()
    n = int(np.frombuffer(buf, np.int32, 1)[0])
    off = 4
    kps = np.frombuffer(buf, np.uint8, n * 28, off).reshape(n, 28); off += n * 28
    desc = np.frombuffer(buf, np.uint8, n * 32, off).reshape(n, 32); off += n * 32
    bi = np.frombuffer(buf, np.int32, n, off); off += 4 * n
    bd = np.frombuffer(buf, np.int32, n, off); off += 4 * n
    sd = np.frombuffer(buf, np.int32, n, off); off += 4 * n
    d01 = int(np.frombuffer(buf, np.int32, 1, off)[0]); off += 4
    inl = int(np.frombuffer(buf, np.int32, 1, off)[0]); off += 4
    pose = np.frombuffer(buf, np.float64, 7, off); off += 56
    pout = np.frombuffer(buf, np.uint8, 400, off); off += 400
    sinl = int(np.frombuffer(buf, np.int32, 1, off)[0]); off += 4
    s12 = np.frombuffer(buf, np.float64, 7, off); off += 56
    sout = np.frombuffer(buf, np.uint8, 150, off); off += 150
    oinl, opose, oout, _ = oracle.pose_optimization(pp["K4"], pp["pose0"], pp["Xw"], pp["uv"], pp["inv_sigma2"])
    assert inl == oinl and np.array_equal(pout, oout) and np.abs(pose - opose).max() < 1e-7
    on, oS, oo, _ = oracle.optimize_sim3(sp["K1"], sp["K2"], sp["s12_0"], sp["P3D2c"], sp["obs1"], sp["inv_sigma2_1"], sp["P3D1c"], sp["obs2"],
                                         sp["inv_sigma2_2"])
    assert sinl == on and np.array_equal(sout, oo) and np.abs(s12 - oS).max() < 1e-6
    # widened rows through the C++ shims
    kk = np.ascontiguousarray(kps).view(oracle.KP_DTYPE).reshape(-1)          # (checked bit-exact against the oracle below)
    kps4 = np.stack([kk["x"], kk["y"], kk["octave"].astype(np.float32), kk["angle"]], 1).astype(np.float32)
    ooff, oidx = oracle.features_in_area(kps4, np.array([0, 640, 0, 480], np.float32), qxy, qr, qmn, qmx)
    for q in range(nq):
        m = int(np.frombuffer(buf, np.int32, 1, off)[0]); off += 4
        got = np.frombuffer(buf, np.int32, m, off); off += 4 * m
        assert np.array_equal(got, oidx[ooff[q]:ooff[q + 1]].astype(np.int32))
    obw, obv, ofn, ofo, ofi = oracle.bow_transform(voc, desc, 2)
    nw = int(np.frombuffer(buf, np.int32, 1, off)[0]); off += 4
    rec = np.frombuffer(buf, np.dtype([("w", "<u4"), ("v", "<f8")]), nw, off); off += 12 * nw
    assert np.array_equal(rec["w"], obw) and np.array_equal(rec["v"], obv)
    nfv = int(np.frombuffer(buf, np.int32, 1, off)[0]); off += 4
    assert nfv == len(ofn)
    for mI in range(nfv):
        node, cnt = np.frombuffer(buf, np.uint32, 2, off); off += 8
        feats = np.frombuffer(buf, np.uint32, int(cnt), off); off += 4 * int(cnt)
        assert node == ofn[mI] and np.array_equal(feats, ofi[ofo[mI]:ofo[mI + 1]])
    self_score = np.frombuffer(buf, np.float64, 1, off)[0]; off += 8
    assert abs(self_score - 1.0) < 1e-12
    gx = np.frombuffer(buf, np.float64, 7 * len(x0), off).reshape(-1, 7); off += 56 * len(x0)
    ogx, _ = oracle.optimize_essential_graph(x0, eg["fixed"], ej, ei, Sji)
    assert np.abs(gx - ogx).max() < 1e-7 and off == len(buf)
    okps, odesc = oracle.OracleExtractor(1000).extract(img)
    assert n == len(okps)
    assert np.array_equal(kps, okps.view(np.uint8).reshape(n, 28)) and np.array_equal(desc, odesc)
    obi, obd, osd = oracle.hamming_best2(odesc, odesc)
    assert np.array_equal(bi, obi) and np.array_equal(bd, obd) and np.array_equal(sd, osd)
    assert d01 == oracle.descriptor_distance(odesc[0], odesc[1])


def test_reference_signature_dropin_classes(oracle, tmp_path):
    """(b) boundary: csrc/compat/orbslam_dropin.h - ORBmatcher's twelve methods (include/ORBmatcher.h:43-97), every static of
    CeresOptimizer (include/CeresOptimizer.h:351-388: PoseOptimization, GlobalBundleAdjustemnt / BundleAdjustment,
    LocalBundleAdjustment, OptimizeSim3, OptimizeEssentialGraph) and the Frame-side member bodies (ComputeBoW, isInFrustum,
    GetFeaturesInArea, the per-match body of CreateNewMapPoints) with the reference's own signatures, instantiated over mock
    structs that copy the member names of Frame.h / KeyFrame.h / MapPoint.h / Map.h.  tests/cpp/test_dropin.cpp writes every
    call as the reference's call site writes it and runs it through the HIP library, dumping the whole map state before and
    after; tests/dropin_checker.py replays each entry point on the "before" state - an independent Python restatement of the
    reference's semantics over the CPU oracle's flat solves - and the complete "after" state must agree."""
    from ceres_mono_orb_slam2_amd import _lib
    from tests import dropin_checker
    exe = tmp_path / "test_dropin"
    out = tmp_path / "cases"
    out.mkdir()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "cpp"),
                           os.path.join(ROOT, "tests", "cpp", "test_dropin.cpp"), "-o", str(exe), _lib.LIB_PATH, "-lpthread",
                           "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([str(exe), str(out)], capture_output=True, text=True, timeout=600)
    print(r.stdout[-4000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    cases = sorted(os.listdir(out))
    assert len(cases) == 26, cases
    failures = []
    for c in cases:
        fails, info = dropin_checker.check_case(str(out / c), oracle)
        print("%-34s %s%s" % (c[:-4], info, "" if not fails else "   FAIL"))
        failures += ["%s: %s" % (c[:-4], f) for f in fails]
    assert not failures, "\n".join(failures[:40])


def test_opencv_signature_branch(oracle, tmp_path):
    """The branch of csrc/compat/orbslam_compat.h a maintainer with OpenCV builds - ORBextractor::operator()(cv::InputArray image,
    cv::InputArray mask, std::vector<cv::KeyPoint>&, cv::OutputArray descriptors), src/ORBextractor.cc:1040 - compiled against the
    OpenCV API subset under tests/cpp/opencv_api_subset and run: keypoints (cv::KeyPoint's 28-byte layout) and descriptors bit
    for bit the oracle's; an empty image returns silently; the descriptor matrix is n x 32 CV_8U; mvImagePyramid is filled."""
    from ceres_mono_orb_slam2_amd import _lib
    exe = tmp_path / "test_compat_opencv_branch"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "tests", "cpp", "opencv_api_subset"),
                           os.path.join(ROOT, "tests", "cpp", "test_compat_opencv_branch.cpp"), "-o", str(exe), _lib.LIB_PATH,
                           "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib"])
    img = synth.make_frame(78, 752, 480, "blocks")
    raw = tmp_path / "img.raw"; out = tmp_path / "out.bin"
    img.tofile(raw)
    subprocess.check_call([str(exe), str(raw), "752", "480", "1200", str(out)])
    buf = open(out, "rb").read()
    n = int(np.frombuffer(buf, np.int32, 1)[0])
    kps = np.frombuffer(buf, np.uint8, n * 28, 4).reshape(n, 28)
    desc = np.frombuffer(buf, np.uint8, n * 32, 4 + n * 28).reshape(n, 32)
    okps, odesc = oracle.OracleExtractor(1200).extract(img)
    assert n == len(okps) and n > 500
    assert np.array_equal(kps, okps.view(np.uint8).reshape(n, 28)) and np.array_equal(desc, odesc)
