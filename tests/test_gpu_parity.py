"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Bars: bit-exact for primitive ids, hit distances and work counters of explicit
rays and of primary rays; linear RGB within 1e-9 (absolute) on >= 99.9 % of pixels and 8-bit
output within +-1 on >= 99.9 % of pixels for full paths (the only sources of difference are the
last ulp of sin/cos and front-to-back instead of back-to-front summation)."""
import numpy as np
import pytest

from conftest import random_rays

pytestmark = pytest.mark.gpu


def _scenes():
    from raytrace_amd import scenes
    return {
        "tiny": (scenes.tiny_mixed(0), np.zeros(3), 15.0),
        "tiny2": (scenes.tiny_mixed(5, n_tris=200, n_sph=9), np.zeros(3), 15.0),
        "cornell": (scenes.cornell_demo(), np.array([50, 40, 80.0]), 40.0),
        "bunny5k": (scenes.bunny_class(5000), np.array([50, 40, 80.0]), 60.0),
        "sponza40k": (scenes.sponza_class(40000, detail=0.3), np.array([50, 40, 130.0]), 60.0),
    }


@pytest.fixture(scope="module")
def world(ra, po):
    out = {}
    for k, (sc, ctr, spr) in _scenes().items():
        hs = ra.HostScene(sc)
        out[k] = dict(scene=sc, host=hs, dev=ra.DeviceScene(hs, 0), orc=po.OracleScene(sc), center=ctr, spread=spr)
    yield out
    for v in out.values():
        v["dev"].close()


def test_device_present(ra):
    assert ra.device_count() >= 1
    info = ra.device_info(0)
    assert "gfx950" in info, info


def test_fp64_div_sqrt_correctly_rounded(ra):
    rs = np.random.RandomState(0)
    a = np.concatenate([rs.uniform(-1e3, 1e3, 200000), rs.uniform(-1, 1, 100000) * 10.0 ** rs.randint(-30, 30, 100000)])
    b = np.concatenate([rs.uniform(-1e3, 1e3, 200000), rs.uniform(-1, 1, 100000) * 10.0 ** rs.randint(-30, 30, 100000)])
    b[b == 0] = 1.0
    got = ra.debug_fp64("div", a, b)
    assert np.array_equal(got, a / b)
    s = np.abs(a)
    got = ra.debug_fp64("sqrt", s)
    assert np.array_equal(got, np.sqrt(s))


@pytest.mark.parametrize("name", ["tiny", "tiny2", "cornell", "bunny5k", "sponza40k"])
def test_trace_rays_bit_exact(world, name):
    w = world[name]
    rays = random_rays(11, 20000 if name != "sponza40k" else 6000, w["center"], w["spread"])
    tid, tt, sid, st, stats = w["dev"].trace_rays(rays)
    otid, ott, osid, ost, ostats = w["orc"].trace(rays)
    assert np.array_equal(tid, otid)
    assert np.array_equal(tt, ott)
    assert np.array_equal(sid, osid)
    assert np.array_equal(st, ost)
    assert {k: stats[k] for k in ostats} == ostats
    assert (tid >= 0).sum() > 0 or name == "cornell"


@pytest.mark.parametrize("name", ["tiny0", "tiny5", "cornell", "bunny2k", "sponza12k", "texroom", "deeptree"])
def test_trace_rays_match_reference_golden_vectors(ra, name):
    """The HIP path against vectors recorded from the reference's own object code (tests/golden)."""
    from helpers import load_fixture, scene_from_fixture
    fx = load_fixture(name)
    hs = ra.HostScene(scene_from_fixture(fx))
    dev = ra.DeviceScene(hs, 0)
    tid, tt, sid, st, _ = dev.trace_rays(fx["rays"])
    dev.close()
    assert np.array_equal(tid, fx["trace_tri_id"])
    assert np.array_equal(tt, fx["trace_tri_t"])
    assert np.array_equal(sid, fx["trace_sph_id"])
    assert np.array_equal(st, fx["trace_sph_t"])


def test_trace_rays_non_unit_directions(world):
    """Directions far from unit length: the culling margin grows with max(1, |d|); the answers must
    still be the reference's (distances scale with 1/|d|)."""
    w = world["bunny5k"]
    rays = random_rays(21, 3000, w["center"], w["spread"])
    rays[:1000, 3:] *= 7.5
    rays[1000:2000, 3:] *= 0.01
    tid, tt, sid, st, stats = w["dev"].trace_rays(rays)
    otid, ott, osid, ost, ostats = w["orc"].trace(rays)
    assert np.array_equal(tid, otid) and np.array_equal(tt, ott)
    assert np.array_equal(sid, osid) and np.array_equal(st, ost)
    assert stats["tri_tests"] == ostats["tri_tests"] and stats["tri_tests_done"] < stats["tri_tests"]


def test_culling_skips_most_tests_and_keeps_answers(world):
    w = world["sponza40k"]
    rays = random_rays(33, 4000, w["center"], w["spread"])
    tid, tt, _, _, stats = w["dev"].trace_rays(rays)
    otid, ott, _, _, ostats = w["orc"].trace(rays)
    assert np.array_equal(tid, otid) and np.array_equal(tt, ott)
    assert stats["tri_tests"] == ostats["tri_tests"]            # the reference's count is still reported
    assert stats["tri_tests_done"] < 0.1 * stats["tri_tests"]   # but only a fraction is executed


def test_trace_rays_axis_aligned_and_on_plane(world):
    """Rays parallel to the axes, starting on split planes and box faces: the s[w]==split branch
    (kdtree.cpp:137-141) and the |v|>eps gates of check_aabb (kdtree.h:64)."""
    w = world["sponza40k"]
    t = w["host"].tree_arrays()
    rs = np.random.RandomState(3)
    rays = []
    inner = np.nonzero((t["left"] >= 0) | (t["right"] >= 0))[0][:400]
    for n in inner:
        ax = int(t["axis"][n])
        lo, hi = t["box"][n, :3], t["box"][n, 3:]
        o = lo + (hi - lo) * rs.uniform(0, 1, 3)
        o[ax] = t["split"][n]
        for sgn in (-1.0, 1.0):
            d = np.zeros(3); d[ax] = sgn
            rays.append(np.concatenate([o, d]))
            d2 = np.zeros(3); d2[(ax + 1) % 3] = sgn
            rays.append(np.concatenate([o, d2]))
            d3 = rs.normal(size=3); d3[ax] = 0; d3 /= np.linalg.norm(d3)
            rays.append(np.concatenate([o, d3]))
    rays = np.array(rays)
    tid, tt, sid, st, stats = w["dev"].trace_rays(rays)
    otid, ott, osid, ost, ostats = w["orc"].trace(rays)
    assert np.array_equal(tid, otid) and np.array_equal(tt, ott)
    assert np.array_equal(sid, osid) and np.array_equal(st, ost)
    assert {k: stats[k] for k in ostats} == ostats


def _cam(ra, sc, sizex, sizey):
    return ra.capi.make_camera(sc["camera"]["o"], sc["camera"]["d"], sizex, sizey)


def _ocam(po, sc, sizex, sizey):
    return po.make_camera(sc["camera"]["o"], sc["camera"]["d"], sizex, sizey)


@pytest.mark.parametrize("name,sizex,sizey", [("tiny", 48, 64), ("cornell", 60, 80), ("bunny5k", 54, 96), ("sponza40k", 54, 96)])
def test_primary_ids_bit_exact(ra, po, world, name, sizex, sizey):
    """Jitter off, one bounce budget: the id map and the work counters must be identical."""
    w = world[name]
    kw = dict(samps=1, rr_depth=0, rr_mode=0, split_depth=0, hard_depth=1, jitter=0, seed=1)
    rgb, pid, st = w["dev"].render_tile(_cam(ra, w["scene"], sizex, sizey), ra.capi.make_params(**kw), 0, 0, sizex, sizey)
    orgb, opid, ost = w["orc"].render_tile(_ocam(po, w["scene"], sizex, sizey), po.make_params(**kw), 0, 0, sizex, sizey, nthreads=8)
    assert np.array_equal(pid, opid)
    assert {k: st[k] for k in ost} == ost
    assert np.array_equal(rgb, orgb)  # emission of the first hit only: no trig, no summation order
    assert (pid != ra.capi.RT_ID_MISS).mean() > (0.3 if name != "tiny" else 0.01)


def _compare_frames(ra, po, rgb, orgb):
    diff = np.abs(rgb - orgb).max(axis=-1)
    frac_close = (diff <= 1e-9).mean()
    b = ra.tonemap(rgb).astype(np.int32)
    ob = po.draw_pixels(orgb).astype(np.int32)
    frac_8bit = (np.abs(b - ob).max(axis=-1) <= 1).mean()
    return frac_close, frac_8bit, diff


@pytest.mark.parametrize("name,sizex,sizey,kw", [
    ("tiny", 48, 64, dict(samps=2, rr_depth=5, rr_mode=1, split_depth=2, hard_depth=64, jitter=1, seed=7)),
    ("cornell", 60, 80, dict(samps=2, rr_depth=5, rr_mode=1, split_depth=2, hard_depth=64, jitter=1, seed=3)),
    ("cornell", 45, 60, dict(samps=1, rr_depth=8, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=4)),
    ("bunny5k", 54, 96, dict(samps=1, rr_depth=2, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=5)),
    ("sponza40k", 36, 64, dict(samps=1, rr_depth=5, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=6)),
])
def test_full_paths_match_oracle(ra, po, world, name, sizex, sizey, kw):
    w = world[name]
    rgb, pid, st = w["dev"].render_tile(_cam(ra, w["scene"], sizex, sizey), ra.capi.make_params(**kw), 0, 0, sizex, sizey)
    orgb, opid, ost = w["orc"].render_tile(_ocam(po, w["scene"], sizex, sizey), po.make_params(**kw), 0, 0, sizex, sizey, nthreads=8)
    assert np.array_equal(pid, opid)
    frac_close, frac_8bit, diff = _compare_frames(ra, po, rgb, orgb)
    assert frac_close >= 0.999, (frac_close, float(diff.max()))
    assert frac_8bit >= 0.999, frac_8bit
    # the ray count can differ only where a last-ulp trig difference flips a discrete branch
    assert abs(st["rays"] - ost["rays"]) <= 1e-3 * ost["rays"], (st, ost)
    assert np.isfinite(rgb).all() and rgb.min() >= 0 and rgb.max() <= 1


def test_tiles_and_ragged_edges_are_invariant(ra, po, world):
    """Any tiling gives the same pixels, and the packed multi-tile launch equals per-tile launches."""
    import torch
    from raytrace_amd import frame
    w = world["cornell"]
    sizex, sizey = 45, 70  # not multiples of 8 or of the tile size
    kw = dict(samps=1, rr_depth=5, rr_mode=1, split_depth=2, hard_depth=64, jitter=1, seed=9)
    cam, prm = _cam(ra, w["scene"], sizex, sizey), ra.capi.make_params(**kw)
    full, fpid, _ = w["dev"].render_tile(cam, prm, 0, 0, sizex, sizey)
    tiles = frame.plan_tiles(sizex, sizey, 16, 24)
    for world_size in (1, 2, 3):
        shards = [frame.shard_tiles(tiles, r, world_size) for r in range(world_size)]
        packed, packed_ids = [], []
        for sh in shards:
            n = frame.tile_pixels(sh)
            buf = torch.zeros((n, 3), dtype=torch.float64, device="cuda:0")
            ids = torch.zeros((n,), dtype=torch.int32, device="cuda:0")
            w["dev"].render_tiles_device(cam, prm, sh, buf.data_ptr(), ids.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            packed.append(buf.cpu().numpy())
            packed_ids.append(ids.cpu().numpy())
        got = frame.assemble_host(sizex, sizey, shards, packed)
        got_ids = frame.assemble_host(sizex, sizey, shards, packed_ids, channels=0, dtype=np.int32)
        assert np.array_equal(got, full)
        assert np.array_equal(got_ids, fpid)
    # a single ragged tile in the middle of the image
    rgb, pid, _ = w["dev"].render_tile(cam, prm, 7, 13, 11, 19)
    assert np.array_equal(rgb, full[7:18, 13:32]) and np.array_equal(pid, fpid[7:18, 13:32])


def test_large_frame_same_pixels_and_work_for_any_tiling(ra, world):
    """A frame with many more pieces of work than the chip has ray slots, so that every wavefront
    takes work many times and runs through all of the frame's piece counters: two different
    tilings and a repeat give bit-identical pixels, ids and work counters, and every pixel of a
    lattice matches its own one-pixel launch."""
    import torch
    from raytrace_amd import frame
    w = world["sponza40k"]
    sizex, sizey = 1000, 1500  # 6 M sub-pixels; not multiples of the tile sizes
    kw = dict(samps=1, rr_depth=3, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=21)
    cam, prm = _cam(ra, w["scene"], sizex, sizey), ra.capi.make_params(**kw)
    w["dev"].stats()
    results = []
    for tx, ty in ((64, 64), (96, 200), (64, 64)):
        tiles = frame.plan_tiles(sizex, sizey, tx, ty)
        n = frame.tile_pixels(tiles)
        buf = torch.zeros((n, 3), dtype=torch.float64, device="cuda:0")
        ids = torch.zeros((n,), dtype=torch.int32, device="cuda:0")
        w["dev"].render_tiles_device(cam, prm, tiles, buf.data_ptr(), ids.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        st = w["dev"].stats()
        img = frame.assemble_host(sizex, sizey, [tiles], [buf.cpu().numpy()])
        pid = frame.assemble_host(sizex, sizey, [tiles], [ids.cpu().numpy()], channels=0, dtype=np.int32)
        results.append((img, pid, {k: st[k] for k in ("rays", "node_visits", "tri_tests", "tri_shades")}))
    for img, pid, st in results[1:]:
        assert np.array_equal(img, results[0][0]) and np.array_equal(pid, results[0][1])
        assert st == results[0][2]
    assert results[0][2]["rays"] >= 4 * sizex * sizey
    for x in range(3, sizex, 331):
        for y in range(5, sizey, 373):
            rgb, one, _ = w["dev"].render_tile(cam, prm, x, y, 1, 1)
            assert np.array_equal(rgb[0, 0], results[0][0][x, y]) and one[0, 0] == results[0][1][x, y]


def test_frame_renderer_single_rank(ra, world):
    import torch
    from raytrace_amd import frame
    w = world["tiny"]
    sizex, sizey = 40, 56
    kw = dict(samps=1, rr_depth=3, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=2)
    cam, prm = _cam(ra, w["scene"], sizex, sizey), ra.capi.make_params(**kw)
    fr = frame.FrameRenderer(w["dev"], cam, prm, 16, 16, want_ids=True)
    out = fr.render()
    torch.cuda.synchronize()
    full, fpid, st = w["dev"].render_tile(cam, prm, 0, 0, sizex, sizey)
    assert np.array_equal(out.cpu().numpy(), full)
    assert np.array_equal(fr.ids.cpu().numpy(), fpid)
    assert w["dev"].last_kernel_ms() > 0


def test_edge_scenes(ra, po):
    """Empty scene, spheres only, triangles only, a single triangle, duplicate triangles (the
    first one in list order wins a tie) and coincident spheres (the LAST index wins, sphere.cpp:7)."""
    from raytrace_amd import scenes
    tri = np.array([[-5, -5, 0, 5, -5, 0, 0, 5, 0.0]])
    cases = {
        "empty": {"meshes": [], "spheres": []},
        "one_tri": {"meshes": [scenes._mesh(tri, (.5, .5, .5))], "spheres": []},
        "dup_tris": {"meshes": [scenes._mesh(np.repeat(tri, 6, 0), (.5, .5, .5)), scenes._mesh(tri, (.2, .2, .2))], "spheres": []},
        "dup_spheres": {"meshes": [], "spheres": [scenes._sphere(2, (0, 0, 0), (1, 1, 1), (.5, .5, .5), 0)] * 3},
        "tri_vs_sphere_tie": {"meshes": [scenes._mesh(tri, (.5, .5, .5))], "spheres": [scenes._sphere(3, (0, 0, -3), (0, 0, 0), (.5, .5, .5), 0)]},
    }
    rays = np.array([[0, 0, 10, 0, 0, -1.0], [0.5, -0.5, 10, 0, 0, -1.0], [0, 0, -10, 0, 0, 1.0], [50, 50, 10, 0, 0, -1.0],
                     [0, 0, 0, 0, 0, -1.0], [1, 1, 5, 0.1, 0.05, -1.0]])
    rays[:, 3:] /= np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
    for name, sc in cases.items():
        hs = ra.HostScene(sc)
        dev = ra.DeviceScene(hs, 0)
        orc = po.OracleScene(sc)
        a = dev.trace_rays(rays)
        b = orc.trace(rays)
        for x, y in zip(a[:4], b[:4]):
            assert np.array_equal(x, y), name
        assert {k: a[4][k] for k in b[4]} == b[4], name
        cam = ra.capi.make_camera((0, 0, 160), (0, 0, -1), 16, 24)
        ocam = po.make_camera((0, 0, 160), (0, 0, -1), 16, 24)
        kw = dict(samps=1, rr_depth=3, rr_mode=0, split_depth=2, hard_depth=8, jitter=0, seed=1)
        rgb, pid, st = dev.render_tile(cam, ra.capi.make_params(**kw), 0, 0, 16, 24)
        orgb, opid, ost = orc.render_tile(ocam, po.make_params(**kw), 0, 0, 16, 24)
        assert np.array_equal(pid, opid), name
        assert np.abs(rgb - orgb).max() <= 1e-9, name
        dev.close()
    # tie rules, stated explicitly
    hs = ra.HostScene(cases["dup_tris"]); dev = ra.DeviceScene(hs, 0)
    tid, tt, _, _, _ = dev.trace_rays(rays[:1])
    assert tid[0] == 0 and tt[0] == 10.0
    dev.close()
    hs = ra.HostScene(cases["dup_spheres"]); dev = ra.DeviceScene(hs, 0)
    _, _, sid, stt, _ = dev.trace_rays(rays[:1])
    assert sid[0] == 2 and stt[0] == 8.0
    dev.close()


def test_equal_distances_across_own_lists(ra, po):
    """Coplanar overlapping triangles of very different size sit in different own lists (the large
    ones straddle split planes and stay on interior nodes); an axis-parallel ray meets a small and
    a large one at exactly the same distance.  Within a list the earlier entry wins, across lists
    the list visited first (kdtree.cpp:105 with its strict <)."""
    from raytrace_amd import scenes
    rng = np.random.default_rng(11)
    n = 24
    small = []
    for i in range(n):
        for j in range(n):
            x, y = i - n / 2, j - n / 2
            small.append([x, y, 0, x + 1, y, 0, x, y + 1, 0.0])
            small.append([x + 1, y + 1, 0, x, y + 1, 0, x + 1, y, 0.0])
    small = np.array(small)
    big = []
    for k in range(40):  # large triangles in the same plane, at random places
        c = rng.uniform(-n / 2, n / 2, 2); r = rng.uniform(2, 9)
        ang = rng.uniform(0, 2 * np.pi) + np.array([0, 2.1, 4.2])
        big.append(np.stack([c[0] + r * np.cos(ang), c[1] + r * np.sin(ang), np.zeros(3)], 1).reshape(9))
    big = np.array(big)
    perm = rng.permutation(len(small) + len(big))  # list order mixes the two kinds
    sc = {"meshes": [scenes._mesh(np.concatenate([small, big])[perm], (.5, .5, .5))], "spheres": []}
    m = 3000
    rays = np.zeros((m, 6))
    rays[:, :2] = rng.uniform(-n / 2, n / 2, (m, 2))
    rays[:, 2] = np.where(np.arange(m) % 2 == 0, 8.0, -8.0)
    rays[:, 5] = np.where(np.arange(m) % 2 == 0, -1.0, 1.0)
    hs = ra.HostScene(sc)
    dev = ra.DeviceScene(hs, 0)
    orc = po.OracleScene(sc)
    a = dev.trace_rays(rays)
    b = orc.trace(rays)
    for x, y in zip(a[:4], b[:4]):
        assert np.array_equal(x, y)
    assert {k: a[4][k] for k in b[4]} == b[4]
    is_big = perm[np.maximum(a[0], 0)] >= len(small)
    hit = a[0] >= 0
    assert hit.mean() > 0.9 and is_big[hit].any() and (~is_big[hit]).any()  # both kinds win ties
    dev.close()


@pytest.mark.parametrize("name", ["bunny5k", "sponza40k"])
def test_far_origins_and_long_directions(world, name):
    """Rays that start many scene sizes away, and directions much longer or shorter than one: the
    culling boxes then use each list's largest margin coefficient instead of the working one (the
    inflation built into the boxes at upload only covers origins within a few scene sizes)."""
    w = world[name]
    rs = np.random.RandomState(23)
    n = 4000
    target = w["center"] + rs.uniform(-0.6, 0.6, (n, 3)) * w["spread"]
    away = rs.normal(size=(n, 3))
    away /= np.linalg.norm(away, axis=1, keepdims=True)
    dist = 10.0 ** rs.uniform(0.5, 4.0, (n, 1)) * w["spread"]  # 3 .. 10 000 scene sizes
    o = target + away * dist
    d = (target - o) / np.linalg.norm(target - o, axis=1, keepdims=True)
    d[n // 2:] *= 10.0 ** rs.uniform(-3, 3, (n - n // 2, 1))     # not unit length
    o[n // 2:] = target[n // 2:] + away[n // 2:] * rs.uniform(1, 3, (n - n // 2, 1)) * w["spread"]
    rays = np.concatenate([o, d], 1)
    tid, tt, sid, st, stats = w["dev"].trace_rays(rays)
    otid, ott, osid, ost, ostats = w["orc"].trace(rays)
    assert np.array_equal(tid, otid)
    assert np.array_equal(tt, ott)
    assert np.array_equal(sid, osid)
    assert np.array_equal(st, ost)
    assert {k: stats[k] for k in ostats} == ostats
    assert (tid >= 0).sum() > n // 10


def test_sphere_hierarchy_matches_brute_force_scan(ra, po):
    """Scenes with 16 or more spheres search them through a hierarchy after the kd-tree; the reference
    scans all of them (sphere.cpp:3-14).  Same ids, distances and work counters as the oracle for random
    rays, axis-parallel rays, rays that start inside spheres, coincident and touching spheres, and
    directions that are not unit length (those fall back to the scan); full paths agree too."""
    from raytrace_amd import scenes
    sc = scenes.random_spheres(400, 24, 32, seed=7)
    sph = sc["spheres"]
    sph += [scenes._sphere(3, (50, 50, 40), (0, 0, 0), (.5, .5, .5), 0)] * 3            # coincident: the last index wins
    sph += [scenes._sphere(2, (60, 50, 40), (0, 0, 0), (.5, .5, .5), 0), scenes._sphere(2, (64, 50, 40), (0, 0, 0), (.5, .5, .5), 0)]
    sph += [scenes._sphere(1e4, (50, -1e4 - 120, 0), (0, 0, 0), (.7, .7, .7), 0)]       # a huge one under everything
    tri = np.array([[-200, -200, -150, 300, -200, -150, 50, 300, -150.0]])
    for meshes in ([], [scenes._mesh(tri, (.5, .5, .5))]):
        scn = {"meshes": meshes, "spheres": sph}
        hs = ra.HostScene(scn)
        dev = ra.DeviceScene(hs, 0)
        orc = po.OracleScene(scn)
        rs = np.random.RandomState(5)
        o = rs.uniform(-120, 220, (6000, 3))
        d = rs.normal(size=(6000, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = np.concatenate([o, d], 1)
        rays[:200, 3:] = np.eye(3)[rs.randint(0, 3, 200)] * rs.choice([-1.0, 1.0], (200, 1))  # axis-parallel
        rays[200:300, :3] = np.array([s_["pos"] for s_ in sph[:100]])                               # from sphere centres
        rays[300:330] = [62, 50, 90, 0, 0, -1.0]                                                    # through the touching point
        rays[330:360] = [50, 50, 90, 0, 0, -1.0]                                                    # at the coincident spheres
        rays[360:460, 3:] *= rs.uniform(0.3, 3.0, (100, 1))                                         # not unit length
        a = dev.trace_rays(rays)
        b = orc.trace(rays)
        for x, y in zip(a[:4], b[:4]):
            assert np.array_equal(x, y)
        assert {k: a[4][k] for k in b[4]} == b[4]
        assert (a[2] >= 0).sum() > 1000
        assert a[2][330] == len(sph) - 4  # the last of the three coincident spheres
        cam = ra.capi.make_camera((50, 50, 300), (0, 0, -1), 24, 32)
        ocam = po.make_camera((50, 50, 300), (0, 0, -1), 24, 32)
        kw = dict(samps=1, rr_depth=4, rr_mode=0, split_depth=2, hard_depth=8, jitter=1, seed=3)
        rgb, pid, st = dev.render_tile(cam, ra.capi.make_params(**kw), 0, 0, 24, 32)
        orgb, opid, ost = orc.render_tile(ocam, po.make_params(**kw), 0, 0, 24, 32)
        assert np.array_equal(pid, opid)
        assert (np.abs(rgb - orgb).max(axis=-1) <= 1e-9).mean() >= 0.999
        assert {k: st[k] for k in ost} == ost
        dev.close()


def test_invalid_arguments_are_rejected(ra, world):
    w = world["tiny"]
    cam = _cam(ra, w["scene"], 16, 16)
    with pytest.raises(ra.capi.RtError):
        w["dev"].render_tile(cam, ra.capi.make_params(samps=0), 0, 0, 16, 16)
    with pytest.raises(ra.capi.RtError):
        w["dev"].render_tile(cam, ra.capi.make_params(), 8, 8, 16, 16)  # tile leaves the image
    with pytest.raises(ra.capi.RtError):
        w["dev"].render_tile(cam, ra.capi.make_params(split_depth=9), 0, 0, 16, 16)
    with pytest.raises(ra.capi.RtError):
        ra.DeviceScene(w["host"], 99)
