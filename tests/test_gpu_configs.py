"""GPU parity tests for the configurations and corner cases the basic parity file does not reach:
the 10 000-sphere scene at depth 8 (sphere-hierarchy kernel instance), a tree deep enough that the
library picks its deep-tree kernel build by itself, the traversal-stack limit, a multi-million
triangle scene, scenes far from / tiny against the coordinate origin (where the culling margins'
absolute-coordinate terms matter), rays grazing triangle edges and vertices, and a sweep over
generated scenes.  Same bars as tests/test_gpu_parity.py: ids, distances and work counters
bit-exact against the CPU oracle; full paths within 1e-9 on >= 99.9 % of pixels."""
import ctypes

import numpy as np
import pytest

from conftest import random_rays

pytestmark = pytest.mark.gpu


def _same_queries(dev, orc, rays):
    a = dev.trace_rays(rays)
    b = orc.trace(rays)
    assert np.array_equal(a[0], b[0])
    assert np.array_equal(a[1], b[1])
    assert np.array_equal(a[2], b[2])
    assert np.array_equal(a[3], b[3])
    assert {k: a[4][k] for k in b[4]} == b[4], (a[4], b[4])
    return a


def _same_frame(ra, po, dev, orc, sc, sizex, sizey, kw, nthreads=8, exact_rays=False):
    cam = ra.capi.make_camera(sc["camera"]["o"], sc["camera"]["d"], sizex, sizey)
    ocam = po.make_camera(sc["camera"]["o"], sc["camera"]["d"], sizex, sizey)
    rgb, pid, st = dev.render_tile(cam, ra.capi.make_params(**kw), 0, 0, sizex, sizey)
    orgb, opid, ost = orc.render_tile(ocam, po.make_params(**kw), 0, 0, sizex, sizey, nthreads=nthreads)
    assert np.array_equal(pid, opid)
    close = (np.abs(rgb - orgb).max(axis=-1) <= 1e-9).mean()
    assert close >= 0.999, close
    b, ob = ra.tonemap(rgb).astype(np.int32), po.draw_pixels(orgb).astype(np.int32)
    assert (np.abs(b - ob).max(axis=-1) <= 1).mean() >= 0.999
    if exact_rays:
        assert {k: st[k] for k in ost} == ost, (st, ost)
    else:
        assert abs(st["rays"] - ost["rays"]) <= 1e-3 * ost["rays"] + 2, (st, ost)
    return rgb, pid, st


def test_ten_thousand_spheres_depth_8(ra, po):
    """BASELINE configuration 3: 10 000 random spheres, mirror / glass mix, depth 8, no roulette.  The
    reference scans every sphere for every ray (sphere.cpp:3-14); the device searches a hierarchy."""
    from raytrace_amd import scenes
    sc = scenes.random_spheres(10000, 24, 32)
    hs = ra.HostScene(sc)
    dev = ra.DeviceScene(hs, 0)
    orc = po.OracleScene(sc)
    rays = random_rays(31, 6000, np.array([50, 50, 0.0]), 130.0)
    rays[:300, :3] = np.array([s["pos"] for s in sc["spheres"][:300]])  # from inside spheres
    a = _same_queries(dev, orc, rays)
    assert (a[2] >= 0).mean() > 0.3
    assert a[4]["sph_tests"] == 10000 * len(rays) and a[4]["sph_tests_done"] < 0.01 * a[4]["sph_tests"]
    kw = dict(samps=1, rr_depth=8, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=5)
    _, _, st = _same_frame(ra, po, dev, orc, sc, 24, 32, kw, nthreads=16)
    assert st["rays"] > 3 * 4 * 24 * 32  # long mirror / glass paths
    dev.close()


def _deep_rays(n, seed, size=1000.0):
    rs = np.random.RandomState(seed)
    o = rs.uniform(-0.2 * size, 1.2 * size, (n, 3))
    tgt = rs.uniform(0, 1, (n, 3)) ** 4 * size  # mostly into the corner where the tree is deep
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], 1)


def test_naturally_deep_tree_runs_the_deep_tree_kernel_build(ra, po):
    """A tree of 40-odd levels (the Lucy-class tree has 31): the library picks the kernel build with the
    8-level stack ring from the tree's depth alone, and most stack levels travel through HBM."""
    import os
    from raytrace_amd import scenes
    assert "RT_HIP_DEEP_TREE" not in os.environ
    sc = scenes.deep_tree()
    hs = ra.HostScene(sc)
    assert hs.tree_arrays()["depth"] >= 36
    dev = ra.DeviceScene(hs, 0)
    orc = po.OracleScene(sc)
    a = _same_queries(dev, orc, _deep_rays(8000, 2))
    assert (a[0] >= 0).mean() > 0.2 and a[4]["node_visits"] > 25 * 8000
    kw = dict(samps=1, rr_depth=4, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=8)
    _same_frame(ra, po, dev, orc, sc, 48, 64, kw)
    dev.close()
    # the same scene through the other build (env override) gives the same answers: both are exact
    os.environ["RT_HIP_DEEP_TREE"] = "1000"
    try:
        dev = ra.DeviceScene(hs, 0)
        _same_queries(dev, orc, _deep_rays(3000, 3))
        dev.close()
    finally:
        del os.environ["RT_HIP_DEEP_TREE"]


def test_wavefront_wide_list_scan_of_a_sliver_fan(ra, po):
    """The deep-tree build lets the whole wavefront scan ONE slot's own list once only a handful of slots are left
    (a ray through a fan of thousands of slivers would otherwise take thousands of steps alone).  A sphere of 8 bands
    x 3 000 segments has two such fans; rays straight down its axis meet both.  Same ids, distances and reference
    work counters as the oracle, the same as with the scan switched off, and a frame looking down the axis."""
    import os
    from raytrace_amd import scenes
    body = scenes.blob((50, 30, 80), 20.0, 8, 3000, seed=7, amp=0.2)
    floor = scenes.grid((-100, 0, -100), (0, 0, 400), (400, 0, 0), 2, 2)
    sc = {"name": "fan", "meshes": [scenes._mesh(floor, (.6, .6, .6), scenes.DIFF), scenes._mesh(body, (.75, .7, .6), scenes.DIFF)],
          "spheres": [scenes._sphere(40, (60, 150, 140), (12, 12, 12), (0, 0, 0), scenes.DIFF)],
          "camera": {"o": np.array([50.3, 230.0, 80.2]), "d": np.array([0.001, -1.0, 0.002])}, "sizex": 24, "sizey": 32}
    hs = ra.HostScene(sc)
    orc = po.OracleScene(sc)
    rs = np.random.RandomState(5)
    n = 400
    o = np.stack([50 + rs.normal(0, 0.02, n), np.full(n, 120.0), 80 + rs.normal(0, 0.02, n)], 1)
    d = np.stack([rs.normal(0, 1e-4, n), np.full(n, -1.0), rs.normal(0, 1e-4, n)], 1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([np.concatenate([o, d], 1), random_rays(9, 2000, np.array([50, 40, 80.0]), 60.0)])
    assert "RT_HIP_DEEP_TREE" not in os.environ and "RT_HIP_COOP" not in os.environ
    os.environ["RT_HIP_DEEP_TREE"] = "0"
    try:
        dev = ra.DeviceScene(hs, 0)
        os.environ["RT_HIP_COOP"] = "0"
        dev_off = ra.DeviceScene(hs, 0)
    finally:
        del os.environ["RT_HIP_DEEP_TREE"]
        os.environ.pop("RT_HIP_COOP", None)
    a = _same_queries(dev, orc, rays)
    b = _same_queries(dev_off, orc, rays)
    assert (a[0][:n] >= 0).all() and a[4]["tri_tests"] > 1000 * n  # the reference tests thousands of slivers per ray
    # the scan ran: far fewer scheduler steps than with one slot working through the fan alone
    assert a[4]["sched_steps"] < 0.75 * b[4]["sched_steps"], (a[4]["sched_steps"], b[4]["sched_steps"])
    kw = dict(samps=1, rr_depth=3, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=4)
    rgb, pid, _ = _same_frame(ra, po, dev, orc, sc, 24, 32, kw)
    rgb0, pid0, _ = _same_frame(ra, po, dev_off, orc, sc, 24, 32, kw)
    assert np.array_equal(pid, pid0) and np.array_equal(rgb, rgb0)
    dev.close()
    dev_off.close()


def _chain_desc(ra, depth):
    """A degenerate but valid description: `depth` nodes, each the left child of the one before, one
    triangle on the last node's own list."""
    n = depth
    arrs = dict(
        box=np.tile(np.array([-1, -1, -1, 1, 1, 1.0]), (n, 1)), split=np.zeros(n), axis=(np.arange(n) % 3).astype(np.int32),
        left=np.concatenate([np.arange(1, n), [-1]]).astype(np.int32), right=np.full(n, -1, np.int32),
        first=np.zeros(n, np.int32), count=np.concatenate([np.zeros(n - 1), [1]]).astype(np.int32), list=np.zeros(1, np.int32),
        verts=np.array([[-0.5, -0.5, 0, 0.5, -0.5, 0, 0, 0.5, 0.0]]), nc=np.array([[0, 0, 1.0]]), col=np.full((1, 3), 0.5),
        lig=np.zeros((1, 3)), type=np.zeros(1, np.int32))
    arrs["split"][:] = 2.0  # every origin lies left of the plane: the chain is walked to its end
    d = ra.capi.SceneDesc()
    dp, ip = ra.capi.dptr, ra.capi.iptr
    d.n_nodes = n
    d.node_box, d.node_split, d.node_axis = dp(arrs["box"]), dp(arrs["split"]), ip(arrs["axis"])
    d.node_left, d.node_right, d.node_first, d.node_count = ip(arrs["left"]), ip(arrs["right"]), ip(arrs["first"]), ip(arrs["count"])
    d.n_list, d.list_tri, d.tree_depth = 1, ip(arrs["list"]), n
    d.n_tris, d.tri_verts, d.tri_nc, d.tri_col, d.tri_lig, d.tri_type = 1, dp(arrs["verts"]), dp(arrs["nc"]), dp(arrs["col"]), dp(arrs["lig"]), ip(arrs["type"])
    d.n_spheres = 0
    return d, arrs


def test_stack_limit_is_an_error_and_the_deepest_allowed_tree_works(ra):
    """The traversal stack pointer is kept in 8 bits: a tree that needs more levels is rejected with
    RT_E_LIMIT at upload; one just below the limit uploads and is walked to its last node."""
    D = ra.capi.device_lib()
    d, keep = _chain_desc(ra, 400)
    h = ctypes.c_void_p()
    rc = D.rt_upload_scene(0, ctypes.byref(d), ctypes.byref(h))
    assert rc == -4 and "too deep" in D.rt_last_error().decode(), (rc, D.rt_last_error())
    assert not h
    d, keep = _chain_desc(ra, 240)
    rc = D.rt_upload_scene(0, ctypes.byref(d), ctypes.byref(h))
    assert rc == 0, D.rt_last_error()
    rays = np.array([[0.1, 0.1, 5, 0, 0, -1.0], [0.1, 0.1, -5, 0, 0, 1.0], [3, 3, 5, 0, 0, -1.0]])
    tid, tt = np.zeros(3, np.int32), np.zeros(3)
    sid, st = np.zeros(3, np.int32), np.zeros(3)
    stats = ra.capi.Stats()
    rc = D.rt_trace_rays(h, 3, ra.capi.dptr(rays), ra.capi.iptr(tid), ra.capi.dptr(tt), ra.capi.iptr(sid), ra.capi.dptr(st), ctypes.byref(stats))
    assert rc == 0, D.rt_last_error()
    assert list(tid) == [0, 0, -1] and tt[0] == 5.0 and tt[1] == 5.0
    assert stats.node_visits == 240 + 240 + 1  # the third ray misses the root's box
    D.rt_free_scene(h)


def test_four_million_triangles_same_pixels_and_work_for_any_tiling(ra, po):
    """A multi-million-triangle scene (27-level tree, 16 M traversal records): two tilings and a repeat
    give bit-identical frames and counters, single pixels match their own launches, and a sample of
    explicit rays matches the oracle."""
    import torch
    from raytrace_amd import frame, scenes
    sc = scenes.bunny_class(4_000_000, 270, 480)
    hs = ra.HostScene(sc)
    dev = ra.DeviceScene(hs, 0)
    sizex, sizey = 270, 480
    kw = dict(samps=1, rr_depth=2, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=4)
    cam, prm = ra.capi.make_camera(sc["camera"]["o"], sc["camera"]["d"], sizex, sizey), ra.capi.make_params(**kw)
    dev.stats()
    results = []
    for tx, ty in ((64, 64), (50, 120), (64, 64)):
        tiles = frame.plan_tiles(sizex, sizey, tx, ty)
        n = frame.tile_pixels(tiles)
        buf = torch.zeros((n, 3), dtype=torch.float64, device="cuda:0")
        ids = torch.zeros((n,), dtype=torch.int32, device="cuda:0")
        dev.render_tiles_device(cam, prm, tiles, buf.data_ptr(), ids.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        st = dev.stats()
        img = frame.assemble_host(sizex, sizey, [tiles], [buf.cpu().numpy()])
        pid = frame.assemble_host(sizex, sizey, [tiles], [ids.cpu().numpy()], channels=0, dtype=np.int32)
        results.append((img, pid, {k: st[k] for k in ("rays", "node_visits", "tri_tests", "tri_shades")}))
    for img, pid, st in results[1:]:
        assert np.array_equal(img, results[0][0]) and np.array_equal(pid, results[0][1]) and st == results[0][2]
    assert (results[0][1] >= 0).mean() > 0.3
    for x, y in ((3, 5), (135, 240), (200, 101), (269, 479)):
        rgb, one, _ = dev.render_tile(cam, prm, x, y, 1, 1)
        assert np.array_equal(rgb[0, 0], results[0][0][x, y]) and one[0, 0] == results[0][1][x, y]
    orc = po.OracleScene(sc)  # (the oracle's tree build and 300 queries take a few seconds at this size)
    _same_queries(dev, orc, random_rays(5, 300, np.array([50, 40, 80.0]), 60.0))
    dev.close()


@pytest.mark.parametrize("base", ["bunny5k", "sponza40k"])
@pytest.mark.parametrize("scale,shift", [(1.0, (1e5, -2e5, 3e5)), (1e-3, (0.0, 0.0, 0.0)), (1e3, (0.0, 0.0, 0.0)),
                                          (1.0, (3e6, 3e6, -3e6)), (1e-2, (-40.0, 25.0, 10.0))])
def test_translated_and_scaled_scenes(ra, po, base, scale, shift):
    """The same parity bars far from the coordinate origin and at other scales: the culling margins'
    terms in the absolute coordinates (and, at small scale, the eps-sized constants of the exact
    tests against tiny triangles) decide there.  The oracle runs on the transformed scene."""
    from raytrace_amd import scenes
    if base == "bunny5k":
        sc0, ctr, spr = scenes.bunny_class(5000), np.array([50, 40, 80.0]), 60.0
    else:
        sc0, ctr, spr = scenes.sponza_class(40000, detail=0.3), np.array([50, 40, 130.0]), 60.0
    sc = scenes.transformed(sc0, scale, shift)
    hs = ra.HostScene(sc)
    dev = ra.DeviceScene(hs, 0)
    orc = po.OracleScene(sc)
    n = 6000 if base == "bunny5k" else 3000
    rays = random_rays(41, n, ctr * scale + np.asarray(shift), spr * scale)
    a = _same_queries(dev, orc, rays)
    if scale >= 1.0:
        assert (a[0] >= 0).mean() > 0.2
    if scale <= 1.0:  # (the margins grow with the cube of the scale -- the exact test's det cut is absolute -- and cull little at 1e3)
        assert a[4]["tri_tests_done"] < 0.5 * a[4]["tri_tests"] or a[4]["tri_tests"] == 0
    # rays from far outside the scene towards it (the list's largest margin coefficient applies)
    rs = np.random.RandomState(42)
    tgt = ctr * scale + np.asarray(shift) + rs.uniform(-0.5, 0.5, (1500, 3)) * spr * scale
    away = rs.normal(size=(1500, 3))
    away /= np.linalg.norm(away, axis=1, keepdims=True)
    o = tgt + away * (10.0 ** rs.uniform(0.5, 3.0, (1500, 1))) * spr * scale
    d = (tgt - o) / np.linalg.norm(tgt - o, axis=1, keepdims=True)
    _same_queries(dev, orc, np.concatenate([o, d], 1))
    if scale == 1.0:  # the camera's 140-unit start offset does not scale (raytrace.cpp:203)
        kw = dict(samps=1, rr_depth=3, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=6)
        _same_frame(ra, po, dev, orc, sc, 27, 48, kw)
    dev.close()


def test_rays_at_triangle_edges_vertices_and_in_plane(ra, po):
    """Rays aimed exactly at vertices and edge points of triangles, and rays almost in a triangle's
    plane (det close to the eps cut of polygon.h:48): where the exact test's rounding decides, the
    conservative boxes must not have dropped the triangle."""
    from raytrace_amd import scenes
    for sc, spr in ((scenes.bunny_class(5000), 60.0), (scenes.sponza_class(40000, detail=0.3), 80.0)):
        hs = ra.HostScene(sc)
        dev = ra.DeviceScene(hs, 0)
        orc = po.OracleScene(sc)
        verts = np.concatenate([m["verts"] for m in sc["meshes"]]).reshape(-1, 3, 3)
        rs = np.random.RandomState(9)
        pick = rs.choice(len(verts), 1500, replace=False)
        T = verts[pick]
        rays = []
        for k, t in enumerate(T):
            e1, e2 = t[1] - t[0], t[2] - t[0]
            nrm = np.cross(e1, e2)
            ln = np.linalg.norm(nrm)
            if ln == 0:
                continue
            nrm /= ln
            targets = [t[0], t[1], t[2], 0.5 * (t[0] + t[1]), t[0] + 0.3 * e2, t[1] + 0.7 * (t[2] - t[1])]
            o = t.mean(0) + rs.normal(size=3) * spr * 0.3
            for p in targets:
                d = p - o
                rays.append(np.concatenate([o, d / np.linalg.norm(d)]))
            # nearly in the plane: direction along an edge tilted by ~1e-5..1e-3 towards the normal, through the interior
            c = t.mean(0)
            for tilt in (3e-6, 1e-5, 3e-5, 1e-3):
                d = e1 / np.linalg.norm(e1) + tilt * nrm * (1 if k % 2 else -1)
                d /= np.linalg.norm(d)
                rays.append(np.concatenate([c - d * rs.uniform(0.5, 20.0), d]))
        rays = np.array(rays)
        a = _same_queries(dev, orc, rays)
        assert (a[0] >= 0).mean() > 0.5
        dev.close()


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_generated_scene_sweep(ra, po, seed):
    """Several generated scenes per seed: explicit rays and a small frame each (ids, distances and
    counters bit-exact, pixels within 1e-9)."""
    from raytrace_amd import scenes
    cases = [(scenes.bunny_class(3000 + 4000 * (seed - 10), seed=seed), np.array([50, 40, 80.0]), 60.0),
             (scenes.sponza_class(15000 + 10000 * (seed - 10), seed=seed, detail=0.3), np.array([50, 40, 130.0]), 70.0),
             (scenes.random_spheres(100 * (seed - 8), seed=seed), np.array([50, 40, 80.0]), 80.0),
             (scenes.tiny_mixed(seed, n_tris=60 * (seed - 9), n_sph=seed), np.zeros(3), 15.0)]
    for sc, ctr, spr in cases:
        hs = ra.HostScene(sc)
        dev = ra.DeviceScene(hs, 0)
        orc = po.OracleScene(sc)
        _same_queries(dev, orc, random_rays(1000 + seed, 8000, ctr, spr))
        kw = dict(samps=1, rr_depth=4, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=3)
        _same_frame(ra, po, dev, orc, sc, 48, 64, kw)
        dev.close()
