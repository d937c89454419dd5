"""Textured triangles on the GPU: Polygon::getcol with an image (polygon.h:30-35,73-89) inside full paths,
against the CPU oracle, whose getcol is pinned to the reference's own object code (tests/golden/texroom.npz,
tests/test_oracle_vs_ref.py).  Bars as in tests/test_gpu_parity.py."""
import numpy as np
import pytest

from conftest import random_rays

pytestmark = pytest.mark.gpu


def _frames(ra, po, sc, sizex, sizey, kw, nthreads=8):
    hs = ra.HostScene(sc)
    dev = ra.DeviceScene(hs, 0)
    orc = po.OracleScene(sc)
    cam = ra.capi.make_camera(sc["camera"]["o"], sc["camera"]["d"], sizex, sizey)
    ocam = po.make_camera(sc["camera"]["o"], sc["camera"]["d"], sizex, sizey)
    rgb, pid, st = dev.render_tile(cam, ra.capi.make_params(**kw), 0, 0, sizex, sizey)
    orgb, opid, ost = orc.render_tile(ocam, po.make_params(**kw), 0, 0, sizex, sizey, nthreads=nthreads)
    dev.close()
    return hs, (rgb, pid, st), (orgb, opid, ost)


@pytest.mark.parametrize("kw", [
    dict(samps=4, rr_depth=1, rr_mode=0, split_depth=0, hard_depth=2, jitter=0, seed=1),   # first hit + one bounce: colour = texel x light
    dict(samps=2, rr_depth=5, rr_mode=1, split_depth=2, hard_depth=64, jitter=1, seed=3),  # the reference's integrator settings
    dict(samps=1, rr_depth=6, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=9),
])
def test_textured_room_matches_oracle(ra, po, kw):
    from raytrace_amd import scenes
    sc = scenes.textured_room()
    hs, (rgb, pid, st), (orgb, opid, ost) = _frames(ra, po, sc, 60, 80, kw)
    assert np.array_equal(pid, opid)
    tt = hs.tree_arrays()["tri_tex"]
    hit = pid >= 0
    assert (tt[pid[hit]] >= 0).mean() > 0.5 and len(set(tt[pid[hit]])) == 4  # all three images and plain triangles are seen
    diff = np.abs(rgb - orgb).max(axis=-1)
    assert (diff <= 1e-9).mean() >= 0.999, float(diff.max())
    b, ob = ra.tonemap(rgb).astype(np.int32), po.draw_pixels(orgb).astype(np.int32)
    assert (np.abs(b - ob).max(axis=-1) <= 1).mean() >= 0.999
    assert abs(st["rays"] - ost["rays"]) <= 1e-3 * ost["rays"] + 2
    assert rgb.max() > 0.05


def test_texture_coordinates_outside_the_image(ra, po):
    """Negative, huge and non-finite texture coordinates: the reference indexes out of bounds there (undefined);
    the HIP path and the oracle share one policy (such a texel reads as 0) and must agree exactly."""
    from raytrace_amd import scenes
    sc = scenes.textured_room()
    rs = np.random.RandomState(4)
    for m in sc["meshes"]:
        if m.get("tri_ts") is not None:
            ts = np.array(m["tri_ts"], np.float64)
            ts[0::4] -= rs.uniform(0.5, 3.0, ts[0::4].shape)     # negative
            ts[1::4] *= 1e12                                     # beyond int
            ts[2::4, :2] = [np.inf, np.nan]
            m["tri_ts"] = ts
    kw = dict(samps=4, rr_depth=1, rr_mode=0, split_depth=0, hard_depth=2, jitter=0, seed=1)
    hs, (rgb, pid, st), (orgb, opid, ost) = _frames(ra, po, sc, 60, 80, kw)
    assert np.array_equal(pid, opid)
    assert (np.abs(rgb - orgb).max(-1) <= 1e-9).mean() >= 0.999 and np.isfinite(rgb).all() and rgb.max() > 0.05
    assert abs(st["rays"] - ost["rays"]) <= 1e-3 * ost["rays"] + 2


def test_textures_do_not_change_queries_or_untextured_scenes(ra, po):
    """Explicit-ray queries of a textured scene (they never shade), and the golden vectors of the fixture."""
    from helpers import load_fixture, scene_from_fixture
    fx = load_fixture("texroom")
    sc = scene_from_fixture(fx)
    hs = ra.HostScene(sc)
    dev = ra.DeviceScene(hs, 0)
    tid, tt, sid, st, _ = dev.trace_rays(fx["rays"])
    assert np.array_equal(tid, fx["trace_tri_id"]) and np.array_equal(tt, fx["trace_tri_t"])
    assert np.array_equal(sid, fx["trace_sph_id"]) and np.array_equal(st, fx["trace_sph_t"])
    orc = po.OracleScene(sc)
    rays = random_rays(3, 5000, np.array([20, 30, 20.0]), 60.0)
    a, b = dev.trace_rays(rays), orc.trace(rays)
    for x, y in zip(a[:4], b[:4]):
        assert np.array_equal(x, y)
    assert {k: a[4][k] for k in b[4]} == b[4]
    dev.close()
    # a description with images that no triangle uses runs the plain kernel and gives the plain answer
    sc2 = dict(sc, meshes=[{k: v for k, v in m.items() if k not in ("tri_tex", "tri_ts")} for m in sc["meshes"]])
    sc2["camera"] = {"o": np.array([20, 56, 190.0]), "d": np.array([0.0, -0.15, -1.0])}
    sc3 = dict(sc2, textures=[])
    kw = dict(samps=1, rr_depth=3, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=2)
    _, (r2, p2, s2), _ = _frames(ra, po, sc2, 24, 32, kw)
    _, (r3, p3, s3), (orgb, opid, _) = _frames(ra, po, sc3, 24, 32, kw)
    assert np.array_equal(r2, r3) and np.array_equal(p2, p3) and np.array_equal(p3, opid)
    assert (np.abs(r3 - orgb).max(-1) <= 1e-9).mean() >= 0.999


def test_textured_sponza_class_matches_oracle(ra, po):
    """The textured kernel instance on a real tree: the Sponza-class hall with images on floor, ceiling and the long
    walls (what `bench.py --workload sponza_tex` times, at a size the oracle finishes in seconds).  Same geometry as
    the untextured scene, so the primary ids are those of the untextured frame as well."""
    from raytrace_amd import scenes
    kw = dict(samps=1, rr_depth=5, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=5)
    sc = scenes.sponza_class(40000, 54, 96, detail=0.3, textured=True)
    hs, (rgb, pid, st), (orgb, opid, ost) = _frames(ra, po, sc, 54, 96, kw)
    assert hs.desc.contents.n_textures == 3
    assert np.array_equal(pid, opid)
    tt = hs.tree_arrays()["tri_tex"]
    hit = pid >= 0
    assert (tt[pid[hit]] >= 0).mean() > 0.3  # a good part of the primary hits is textured
    diff = np.abs(rgb - orgb).max(axis=-1)
    assert (diff <= 1e-9).mean() >= 0.999, float(diff.max())
    assert abs(st["rays"] - ost["rays"]) <= 1e-3 * ost["rays"] + 2
    plain = scenes.sponza_class(40000, 54, 96, detail=0.3)
    _, (rgb0, pid0, _), _ = _frames(ra, po, plain, 54, 96, kw)
    assert np.array_equal(pid0, pid) and np.abs(rgb0 - rgb).max() > 1e-3  # same hits, other colours
