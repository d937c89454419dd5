"""Run by tests/test_gpu_stress_build.py in a child process whose RT_HIP_LIB points at the stress
build of the HIP library (2-level stack ring, 192 slots per wavefront), or whose RT_HIP_DEEP_TREE
makes the default library run its build for deep trees on these shallow ones: explicit rays and a small frame
against the CPU oracle, same bars as tests/test_gpu_parity.py.  Prints OK on success."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import pyoracle as po  # noqa: E402
import raytrace_amd as ra  # noqa: E402
from conftest import random_rays  # noqa: E402
from raytrace_amd import scenes  # noqa: E402


def main():
    assert "stress" in os.environ.get("RT_HIP_LIB", "") or "RT_HIP_DEEP_TREE" in os.environ, \
        "RT_HIP_LIB must point at the stress build, or RT_HIP_DEEP_TREE select the other kernel build"
    po.build()
    for sc, ctr, spr, n in ((scenes.sponza_class(40000, detail=0.3), np.array([50, 40, 130.0]), 60.0, 5000),
                            (scenes.bunny_class(5000), np.array([50, 40, 80.0]), 60.0, 8000)):
        hs = ra.HostScene(sc)
        dev = ra.DeviceScene(hs, 0)
        orc = po.OracleScene(sc)
        rays = random_rays(77, n, ctr, spr)
        tid, tt, sid, st, stats = dev.trace_rays(rays)
        otid, ott, osid, ost, ostats = orc.trace(rays)
        assert np.array_equal(tid, otid) and np.array_equal(tt, ott)
        assert np.array_equal(sid, osid) and np.array_equal(st, ost)
        assert {k: stats[k] for k in ostats} == ostats, (stats, ostats)
        # a small frame: primary ids bit-exact, colours within the parity bar
        sizex, sizey = 40, 56
        cam_o, cam_d = sc["camera"]["o"], sc["camera"]["d"]
        kw = dict(samps=1, rr_depth=3, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=5)
        rgb, ids, _ = dev.render_tile(ra.capi.make_camera(cam_o, cam_d, sizex, sizey), ra.capi.make_params(**kw), 0, 0, sizex, sizey)
        orgb, oids, _ = orc.render_tile(po.make_camera(cam_o, cam_d, sizex, sizey), po.make_params(**kw), 0, 0, sizex, sizey, nthreads=8)
        assert np.array_equal(ids, oids)
        close = np.all(np.abs(rgb - orgb) <= 1e-9, axis=-1)
        assert close.mean() >= 0.999, close.mean()
        dev.close()
    print("OK")


if __name__ == "__main__":
    main()
