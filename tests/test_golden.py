"""The oracle (and the product's host-side builder) against the golden vectors recorded from the
reference's own object code (tests/golden/make_golden.py).  Everything here is bit-exact."""
import numpy as np
import pytest

from helpers import SCENE_FIXTURES, load_fixture, scene_from_fixture

TREE_KEYS = ("box", "split", "axis", "left", "right", "first", "count", "list")


@pytest.mark.parametrize("name", SCENE_FIXTURES)
def test_oracle_tree_and_queries_match_reference(po, name):
    fx = load_fixture(name)
    orc = po.OracleScene(scene_from_fixture(fx))
    tree = orc.tree_dump()
    for k in TREE_KEYS:
        assert np.array_equal(tree[k], fx["tree_" + k]), k
    tid, tt, sid, st, _ = orc.trace(fx["rays"])
    assert np.array_equal(tid, fx["trace_tri_id"])
    assert np.array_equal(tt, fx["trace_tri_t"])
    assert np.array_equal(sid, fx["trace_sph_id"])
    assert np.array_equal(st, fx["trace_sph_t"])
    if "nc_gid" in fx:
        for g, nc, x, n, r, tuv in zip(fx["nc_gid"], fx["nc"], fx["getn_x"], fx["getn"], fx["tri_rays"], fx["tri_tuv"]):
            assert np.array_equal(orc.tri_nc(g), nc)
            assert np.array_equal(orc.tri_getn(g, x), n, equal_nan=True)
            t, u, v = orc.tri_intersect(g, r)
            assert t == tuv[0]
            if t < 1e19:  # u, v are only meaningful for an accepted hit
                assert (u, v) == (tuv[1], tuv[2])
    if "sph_idx" in fx:
        for i, r, t in zip(fx["sph_idx"], fx["sph_rays"], fx["sph_t"]):
            assert orc.sph_intersect(i, r) == t
    if "getcol" in fx:  # Polygon::getcol with and without an image (polygon.h:73-89)
        for g, uv, c in zip(fx["getcol_gid"], fx["getcol_uv"], fx["getcol"]):
            assert np.array_equal(orc.tri_getcol(g, uv[0], uv[1]), c), (g, uv)


def test_getn_is_the_face_normal(po):
    """polygon.h:67-69: the guard is true for every non-degenerate point, so getn() == nc."""
    for name in SCENE_FIXTURES:
        fx = load_fixture(name)
        if "nc" in fx:
            assert np.array_equal(fx["getn"], fx["nc"])


@pytest.mark.parametrize("name", SCENE_FIXTURES)
def test_host_builder_matches_reference_tree(ra, name):
    fx = load_fixture(name)
    hs = ra.HostScene(scene_from_fixture(fx))
    t = hs.tree_arrays()
    for k in TREE_KEYS:
        assert np.array_equal(t[k], fx["tree_" + k]), k
    if "nc_gid" in fx:
        assert np.array_equal(t["nc"][fx["nc_gid"]], fx["nc"])
    if "getcol" in fx:  # images and texture coordinates arrive in the description unchanged
        sc = scene_from_fixture(fx)
        for a, b in zip(hs.textures(), sc["textures"]):
            assert np.array_equal(a, b)
        tt = np.concatenate([m.get("tri_tex", np.full(len(m["verts"]), -1, np.int32)) for m in sc["meshes"]])
        assert np.array_equal(t["tri_tex"], tt)
        ts = np.concatenate([m.get("tri_ts", np.zeros((len(m["verts"]), 6))) for m in sc["meshes"]])
        assert np.array_equal(t["tri_ts"][tt >= 0], ts[tt >= 0])


def test_oracle_check_aabb_matches_reference(po):
    fx = load_fixture("aabb")
    got = np.array([po.check_aabb(b, r) for b, r in zip(fx["boxes"], fx["rays"])], np.int8)
    assert np.array_equal(got, fx["hit"])
    assert 0 < got.sum() < len(got)


def test_primary_ray_evaluation_order(po):
    """raytrace.cpp:203 `Ray(cam.o + d * 140, d.norm())`: g++ normalises d first."""
    fx = load_fixture("primary")
    for o, d, ray in zip(fx["camo"], fx["d"], fx["ray"]):
        assert np.array_equal(po.primary_ray(o, d), ray)
    d0 = fx["d"][0]
    assert not np.allclose(fx["ray"][0][:3], fx["camo"][0] + d0 * 140)  # the other order would give this


def test_obj_reader_matches_the_reference_reader_golden(ra, tmp_path):
    """rth_add_obj_file against what the reference's own Object::readfile (object.cpp:9-286, with Matrix::Rx/Ry/Rz and
    Matrix * Point3D, matrix.h:24-52) made of the same OBJ + MTL pair, recorded from its object code: vertices after
    scale -> Rx*Ry*Rz -> translate, the Kd / Ks pick per material type, usemtl (last of a name wins, unknown = object
    colour), texture coordinates, the dropped fourth vertex, face order -- bit for bit, for four transforms."""
    fx = load_fixture("objload")
    (tmp_path / "objload.obj").write_bytes(fx["obj"].tobytes())
    (tmp_path / "objload.mtl").write_bytes(fx["mtl"].tobytes())
    for k in range(int(fx["n_cases"])):
        p = fx["params"][k]
        hs = ra.HostScene()
        hs.add_obj_file(tmp_path / "objload.obj", p[0], p[1:4], p[4:7], p[7:10], p[10:13], int(p[13]))
        hs.build()
        t, d = hs.tree_arrays(), hs.desc.contents
        n = hs.num_tris
        assert n == len(fx[f"c{k}_verts"]) == 30
        assert np.array_equal(t["verts"], fx[f"c{k}_verts"])
        assert np.array_equal(t["col"], fx[f"c{k}_col"])
        assert np.array_equal(np.ctypeslib.as_array(d.tri_lig, shape=(n, 3)), fx[f"c{k}_lig"])
        assert np.array_equal(t["tri_ts"], fx[f"c{k}_ts"])
        assert np.array_equal(t["nc"], fx[f"c{k}_nc"])
        assert np.array_equal(np.ctypeslib.as_array(d.tri_type, shape=(n,)), fx[f"c{k}_type"])
