"""Explicit light sampling (rt_params.light_mode 1) on the GPU against the CPU oracle's restatement of the same
estimator: first-hit ids and work counters bit-exact, pixels within 1e-9 -- the bar of every other render test.
The mode has no counterpart in the reference (its light.h is never used), so its parity is unpinned by
construction; tests/test_light_sampling.py shows that the oracle's version is the same integral as mode 0."""
import os

import numpy as np
import pytest

from raytrace_amd import scenes

pytestmark = pytest.mark.gpu


def compare(ra, po, sc, sizex, sizey, kw, dev=None):
    hs = ra.HostScene(sc)
    own = dev is None
    if own:
        dev = ra.DeviceScene(hs, 0)
    orc = po.OracleScene(sc)
    o, d = sc["camera"]["o"], sc["camera"]["d"]
    rgb, ids, st = dev.render_tile(ra.capi.make_camera(o, d, sizex, sizey), ra.capi.make_params(**kw), 0, 0, sizex, sizey)
    orgb, oids, ost = orc.render_tile(po.make_camera(o, d, sizex, sizey), po.make_params(**kw), 0, 0, sizex, sizey, nthreads=8)
    assert np.array_equal(ids, oids)
    for k in ("rays", "node_visits", "tri_tests", "sph_tests", "tri_shades"):
        assert st[k] == ost[k], (k, st[k], ost[k])
    assert np.abs(rgb - orgb).max() <= 1e-9, np.abs(rgb - orgb).max()
    if own:
        dev.close()
    return rgb, st


CASES = [
    ("lamp", dict(samps=2, rr_depth=5, rr_mode=1, split_depth=2, hard_depth=64, jitter=1, seed=3)),
    ("lamp", dict(samps=1, rr_depth=7, rr_mode=0, split_depth=4, hard_depth=64, jitter=0, seed=4)),  # deepest glass splitting
    ("dome", dict(samps=2, rr_depth=5, rr_mode=1, split_depth=2, hard_depth=64, jitter=1, seed=5)),  # a light that cannot be sampled
    ("cornell", dict(samps=1, rr_depth=5, rr_mode=1, split_depth=2, hard_depth=64, jitter=1, seed=6)),  # most of the light hidden behind the ceiling
    ("tiny", dict(samps=2, rr_depth=5, rr_mode=1, split_depth=2, hard_depth=64, jitter=1, seed=7)),
    ("bunny", dict(samps=1, rr_depth=3, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=8)),
    ("spheres", dict(samps=1, rr_depth=6, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=9)),  # four lights, sphere hierarchy
    ("texroom", dict(samps=2, rr_depth=4, rr_mode=1, split_depth=2, hard_depth=64, jitter=1, seed=10)),  # with textured triangles
]


def scene_of(name):
    return {"lamp": lambda: scenes.lamp_room(dome=False), "dome": lambda: scenes.lamp_room(dome=True),
            "cornell": lambda: scenes.cornell_demo(mesh_bands=8, mesh_segs=10), "tiny": lambda: scenes.tiny_mixed(2),
            "bunny": lambda: scenes.bunny_class(5000), "spheres": lambda: scenes.random_spheres(400, seed=3),
            "texroom": lambda: scenes.textured_room()}[name]()


@pytest.mark.parametrize("name,kw", CASES, ids=[f"{c[0]}-{i}" for i, c in enumerate(CASES)])
def test_light_sampling_matches_oracle(ra, po, name, kw):
    sc = scene_of(name)
    rgb, st = compare(ra, po, sc, 42, 56, dict(kw, light_mode=1))
    assert rgb.max() > 0.05
    # the mode changes the work: shadow rays are counted
    rgb0, st0 = compare(ra, po, sc, 42, 56, dict(kw, light_mode=0))
    assert st["rays"] > st0["rays"]


def test_light_sampling_on_the_deep_tree_kernel_build(ra, po, monkeypatch):
    monkeypatch.setenv("RT_HIP_DEEP_TREE", "0")
    kw = dict(samps=2, rr_depth=5, rr_mode=1, split_depth=2, hard_depth=64, jitter=1, seed=12, light_mode=1)
    compare(ra, po, scenes.lamp_room(dome=True), 36, 48, kw)
    compare(ra, po, scenes.random_spheres(300, seed=4), 36, 48, dict(kw, rr_mode=0))


def test_light_mode_without_emissive_spheres_is_mode_zero(ra, po):
    sc = scenes.lamp_room(dome=False)
    for s in sc["spheres"]:
        s["lig"] = np.zeros(3)
    sc["meshes"][1]["lig"] = np.array([3.0, 2.0, 1.0])
    kw = dict(samps=2, rr_depth=5, rr_mode=1, split_depth=2, hard_depth=64, jitter=1, seed=2)
    a, st_a = compare(ra, po, sc, 36, 48, dict(kw, light_mode=1))
    b, st_b = compare(ra, po, sc, 36, 48, dict(kw, light_mode=0))
    assert np.array_equal(a, b) and a.max() > 0
    assert all(st_a[k] == st_b[k] for k in ("rays", "node_visits", "tri_tests", "sph_tests", "tri_shades"))


def test_modes_alternate_on_one_scene_and_tiles_do_not_matter(ra, po):
    """One uploaded scene serves both modes (two kernel instances share its state arrays), and a frame rendered in
    tiles with light sampling is the frame rendered whole, bit for bit."""
    sc = scenes.lamp_room(dome=True)
    hs = ra.HostScene(sc)
    dev = ra.DeviceScene(hs, 0)
    kw = dict(samps=1, rr_depth=5, rr_mode=1, split_depth=2, hard_depth=64, jitter=1, seed=21)
    first = None
    for mode in (1, 0, 1):
        rgb, st = compare(ra, po, sc, 40, 48, dict(kw, light_mode=mode), dev=dev)
        if mode == 1:
            if first is None:
                first = rgb
            else:
                assert np.array_equal(first, rgb)
    cam = ra.capi.make_camera(sc["camera"]["o"], sc["camera"]["d"], 40, 48)
    prm = ra.capi.make_params(light_mode=1, **kw)
    parts = np.zeros_like(first)
    for x0 in (0, 24):
        for y0 in (0, 16, 32):
            r, _, _ = dev.render_tile(cam, prm, x0, y0, min(24, 40 - x0), 16)
            parts[x0:x0 + r.shape[0], y0:y0 + 16] = r
    assert np.array_equal(parts, first)
    # the device-buffer entry points (the allocating one and the one that allocates nothing) run the same instance
    import torch
    from raytrace_amd import frame
    tiles = frame.plan_tiles(40, 48, 24, 16)
    outs = []
    for fn in (dev.render_tiles_device, dev.render_tiles_noalloc):
        buf = torch.full((frame.tile_pixels(tiles), 3), -1.0, dtype=torch.float64, device="cuda:0")
        fn(cam, prm, tiles, buf.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs.append(buf.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    k = 0
    for (x0, y0, rows, cols) in tiles:
        assert np.array_equal(outs[0][k:k + rows * cols].reshape(rows, cols, 3), first[x0:x0 + rows, y0:y0 + cols])
        k += rows * cols
    bad = ra.capi.make_params(light_mode=2, **kw)
    with pytest.raises(Exception):
        dev.render_tile(cam, bad, 0, 0, 8, 8)
    dev.close()
