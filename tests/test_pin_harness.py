"""tools/pin (the harness that pins this library against the reference's own ORBextractor.cc / cost functors on a machine that has
OpenCV + Eigen + Ceres) cannot run here - no OpenCV, no Ceres - but it must not rot: the CMake project
configures in its PIN_SYNTAX_ONLY mode (the harness sources are parsed against tests/cpp/opencv_api_subset and include/), and the case
generator writes its frames and problems in the layouts the harness reads."""
import os
import shutil
import struct
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("cmake") is None, reason="cmake not installed")
def test_pin_project_configures_and_parses(tmp_path):
    b = str(tmp_path / "build")
    r = subprocess.run(["cmake", "-S", os.path.join(ROOT, "tools", "pin"), "-B", b, "-DPIN_SYNTAX_ONLY=ON"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    r = subprocess.run(["cmake", "--build", b], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    # without the switch the project demands the reference tree - it never falls back to anything in this repository
    r = subprocess.run(["cmake", "-S", os.path.join(ROOT, "tools", "pin"), "-B", str(tmp_path / "b2")], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "ORB_SLAM2_ROOT" in (r.stdout + r.stderr)


def test_pin_cases_have_the_layout_the_harness_reads(tmp_path):
    out = str(tmp_path / "cases")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pin", "make_cases.py"), out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    with open(os.path.join(out, "frame_000.pgm"), "rb") as f:
        assert f.readline() == b"P5\n" and f.readline() == b"1241 376\n" and f.readline() == b"255\n" and len(f.read()) == 1241 * 376
    sz = os.path.getsize(os.path.join(out, "pose_000.bin"))
    n = struct.unpack("<i", open(os.path.join(out, "pose_000.bin"), "rb").read(4))[0]
    assert n == 2000 and sz == 4 + 8 * (4 + 7 + 3 * n + 2 * n) + 4 * n
    hd = struct.unpack("<4i", open(os.path.join(out, "ba_000.bin"), "rb").read(16))
    ncam, npts, nobs, it = hd
    assert os.path.getsize(os.path.join(out, "ba_000.bin")) == 16 + 8 * (4 * ncam + 7 * ncam + 3 * npts + 2 * nobs) + ncam + 8 * nobs + 4 * nobs
    # nothing of the reference is stored under tools/pin
    for fn in os.listdir(os.path.join(ROOT, "tools", "pin")):
        assert fn in ("CMakeLists.txt", "pin_extractor.cpp", "pin_solver.cpp", "pin_matcher.cpp", "make_cases.py", "README.md"), fn
