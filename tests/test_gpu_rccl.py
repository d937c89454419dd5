"""The exchange step on real hardware: torch.distributed.gather on the `nccl` backend (RCCL) with one rank, in a
child process (tests/nccl_gather_check.py), and the native grouped ncclSend / ncclRecv exchange of librt_rccl.so
through the executable, whose two gather modes must write the same image."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frame_gather_over_a_one_rank_rccl_group():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_gather_check.py")], capture_output=True, text=True, timeout=110)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout[-2000:] + out.stderr[-4000:]


def test_executable_rccl_gather_equals_copy_gather(tmp_path):
    sys.path.insert(0, ROOT)
    from raytrace_amd import scenes
    exe = os.path.join(ROOT, "raytrace_amd", "bin", "raytrace_hip")
    scene_file = scenes.write_scene_files(scenes.cornell_demo(72, 100), str(tmp_path), c_samps=1, g_samps=1)
    outs = []
    for mode in ("rccl", "copy"):
        out = str(tmp_path / f"{mode}.bmp")
        r = subprocess.run([exe, scene_file, "-o", out, "--depth", "3", "--rr", "0", "--tile", "32", "--gather", mode], capture_output=True, text=True, timeout=100)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and max(outs[0][54:]) > 100
