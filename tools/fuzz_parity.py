#!/usr/bin/env python3
"""Wider parity sweep than the test suite: several generated scenes and seeds, explicit rays and a
two small frames each (the reference's integrator and explicit light sampling), HIP path against the CPU oracle (ids, distances and work counters bit-exact,
pixels within 1e-9).  Run on the GPU box: python3 tools/fuzz_parity.py [n_rays]; set
RT_HIP_DEEP_TREE=0 to exercise the kernel build for deep trees.  Prints one line per case, then OK."""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    po.build()
    cases = []
    for seed in (11, 12, 13):
        cases.append((f"bunny{seed}", scenes.bunny_class(3000 + 4000 * (seed - 10), seed=seed), np.array([50, 40, 80.0]), 60.0))
        cases.append((f"sponza{seed}", scenes.sponza_class(15000 + 10000 * (seed - 10), seed=seed, detail=0.3), np.array([50, 40, 130.0]), 70.0))
    cases.append(("spheres", scenes.random_spheres(300, seed=5), np.array([50, 40, 80.0]), 80.0))
    cases.append(("spheres3k", scenes.random_spheres(3000, seed=6), np.array([50, 50, 0.0]), 150.0))
    cases.append(("lamps", scenes.lamp_room(), np.array([50, 40, 80.0]), 60.0))
    cases.append(("texroom", scenes.textured_room(), np.array([20, 40, 40.0]), 60.0))
    cases.append(("deep", scenes.deep_tree(), np.array([300, 300, 300.0]), 500.0))
    # a sphere of 8 bands x 2 000 segments: two fans of 2 000 slivers, huge own lists (the wavefront-wide list scan of the deep-tree build)
    fan = {"name": "fan", "meshes": [scenes._mesh(scenes.blob((50, 30, 80), 20.0, 8, 2000, seed=7, amp=0.2), (.75, .7, .6), scenes.DIFF),
                                     scenes._mesh(scenes.grid((-100, 0, -100), (0, 0, 400), (400, 0, 0), 2, 2), (.6, .6, .6), scenes.DIFF)],
           "spheres": [scenes._sphere(40, (60, 150, 140), (12, 12, 12), (0, 0, 0), scenes.DIFF)],
           "camera": {"o": np.array([50.3, 230.0, 80.2]), "d": np.array([0.001, -1.0, 0.002])}, "sizex": 48, "sizey": 64}
    cases.append(("fan", fan, np.array([50, 45, 80.0]), 30.0))
    for name, sc, ctr, spr in cases:
        hs = ra.HostScene(sc)
        dev = ra.DeviceScene(hs, 0)
        orc = po.OracleScene(sc)
        rays = random_rays(1000 + len(name), n, ctr, spr)
        a = dev.trace_rays(rays)
        b = orc.trace(rays)
        for x, y in zip(a[:4], b[:4]):
            assert np.array_equal(x, y), name
        assert {k: a[4][k] for k in b[4]} == b[4], (name, a[4], b[4])
        sizex, sizey = 48, 64
        cam_o, cam_d = sc["camera"]["o"], sc["camera"]["d"]
        kw = dict(samps=1, rr_depth=4, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=3)
        rgb, ids, st = dev.render_tile(ra.capi.make_camera(cam_o, cam_d, sizex, sizey), ra.capi.make_params(**kw), 0, 0, sizex, sizey)
        orgb, oids, ost = orc.render_tile(po.make_camera(cam_o, cam_d, sizex, sizey), po.make_params(**kw), 0, 0, sizex, sizey, nthreads=16)
        assert np.array_equal(ids, oids), name
        assert np.abs(rgb - orgb).max() <= 1e-9, (name, np.abs(rgb - orgb).max())
        assert st["rays"] == ost["rays"] and st["node_visits"] == ost["node_visits"], (name, st, ost)
        # the same frame with explicit light sampling (the oracle's restatement of the same estimator)
        kw1 = dict(kw, light_mode=1, rr_mode=1, seed=4)
        rgb, ids, st = dev.render_tile(ra.capi.make_camera(cam_o, cam_d, sizex, sizey), ra.capi.make_params(**kw1), 0, 0, sizex, sizey)
        orgb, oids, ost = orc.render_tile(po.make_camera(cam_o, cam_d, sizex, sizey), po.make_params(**kw1), 0, 0, sizex, sizey, nthreads=16)
        assert np.array_equal(ids, oids), name
        assert np.abs(rgb - orgb).max() <= 1e-9, (name, np.abs(rgb - orgb).max())
        assert st["rays"] == ost["rays"] and st["node_visits"] == ost["node_visits"] and st["tri_tests"] == ost["tri_tests"], (name, st, ost)
        print(name, hs.num_tris, "tris:", n, "rays and two", sizex, "x", sizey, "frames agree;", a[4]["node_visits"], "visits", flush=True)
        dev.close()
    print("OK")


if __name__ == "__main__":
    main()
