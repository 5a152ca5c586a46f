"""csrc/wave_reduce.h on the device: the transposing 36-value wave reduction k_ba_schur and k_ba_cam_blocks rest on (gfx950's
v_permlane32_swap / v_permlane16_swap for the two widest steps) against a host sum in the same order, bit for bit, and every
value owned by exactly one lane (tools/ubench/wave_reduce36.hip, compiled here with hipcc)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_transposing_wave_reduction_is_exact(tmp_path):
    exe = tmp_path / "wave_reduce36"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-Wno-unused-value", "-o", str(exe),
                           os.path.join(ROOT, "tools", "ubench", "wave_reduce36.hip")])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "wave_reduce36 OK" in r.stdout
