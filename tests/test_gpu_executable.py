"""The raytrace_hip executable (the reference's main, raytrace.cpp:97-248, on top of the two C ABIs):
scene file in the reference's format in, BMP out."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_executable_renders_a_scene_file(tmp_path):
    sys.path.insert(0, ROOT)
    from raytrace_amd import scenes
    exe = os.path.join(ROOT, "raytrace_amd", "bin", "raytrace_hip")
    assert os.path.exists(exe), f"{exe} is missing: run make -C raytrace_amd/csrc"
    sizex, sizey = 96, 128
    scene_file = scenes.write_scene_files(scenes.cornell_demo(sizex, sizey), str(tmp_path), c_samps=1, g_samps=1)
    out = str(tmp_path / "o.bmp")
    r = subprocess.run([exe, scene_file, "-o", out, "--samps", "1", "--depth", "4", "--rr", "0"], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "1020 triangles, 9 spheres" in r.stdout
    data = open(out, "rb").read()
    assert data[:2] == b"BM" and len(data) >= 54 + sizex * sizey * 3
    # not a black frame: the Cornell demo is lit
    assert max(data[54:]) > 100
