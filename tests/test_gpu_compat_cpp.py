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
    subprocess.check_call([str(exe), str(raw), "640", "480", "1000", str(out)])
    buf = open(out, "rb").read()
    n = int(np.frombuffer(buf, np.int32, 1)[0])
    off = 4
    kps = np.frombuffer(buf, np.uint8, n * 28, off).reshape(n, 28); off += n * 28
    desc = np.frombuffer(buf, np.uint8, n * 32, off).reshape(n, 32); off += n * 32
    bi = np.frombuffer(buf, np.int32, n, off); off += 4 * n
    bd = np.frombuffer(buf, np.int32, n, off); off += 4 * n
    sd = np.frombuffer(buf, np.int32, n, off); off += 4 * n
    d01 = int(np.frombuffer(buf, np.int32, 1, off)[0])
    okps, odesc = oracle.OracleExtractor(1000).extract(img)
    assert n == len(okps)
    assert np.array_equal(kps, okps.view(np.uint8).reshape(n, 28)) and np.array_equal(desc, odesc)
    obi, obd, osd = oracle.hamming_best2(odesc, odesc)
    assert np.array_equal(bi, obi) and np.array_equal(bd, obd) and np.array_equal(sd, osd)
    assert d01 == oracle.descriptor_distance(odesc[0], odesc[1])
