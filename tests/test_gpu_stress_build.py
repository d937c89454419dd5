"""The HIP kernels built with a 2-level stack ring and 192 slots per wavefront (librt_hip_stress.so):
nearly every traversal-stack level then travels through the HBM overflow array.  And the default
library's second kernel build (140 slots, 8-level ring), which scenes with shallow trees do not
reach by themselves.  Same parity bars as the default build."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stress_build_matches_oracle():
    lib = os.path.join(ROOT, "raytrace_amd", "lib", "librt_hip_stress.so")
    assert os.path.exists(lib), f"{lib} is missing: run make -C raytrace_amd/csrc"
    env = dict(os.environ, RT_HIP_LIB=lib)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "stress_build_check.py")], env=env, capture_output=True,
                         text=True, timeout=100)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout[-2000:] + out.stderr[-4000:]


def test_deep_tree_kernel_build_matches_oracle():
    lib = os.path.join(ROOT, "raytrace_amd", "lib", "librt_hip.so")
    assert os.path.exists(lib), f"{lib} is missing: run make -C raytrace_amd/csrc"
    env = dict(os.environ, RT_HIP_LIB=lib, RT_HIP_DEEP_TREE="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "stress_build_check.py")], env=env, capture_output=True,
                         text=True, timeout=100)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout[-2000:] + out.stderr[-4000:]
