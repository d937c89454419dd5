"""Host library: builder properties, scene-file / OBJ readers, tone mapping, BMP output."""
import os
import struct

import numpy as np
import pytest


def test_tree_properties(ra):
    from raytrace_amd import scenes
    sc = scenes.sponza_class(20000, detail=0.18)
    hs = ra.HostScene(sc)
    t = hs.tree_arrays()
    n = len(t["axis"])
    assert sorted(t["list"].tolist()) == list(range(hs.num_tris))          # every triangle in exactly one own list
    assert np.array_equal(t["first"], np.concatenate([[0], np.cumsum(t["count"])[:-1]]))
    has_l = t["left"] >= 0
    assert np.array_equal(t["left"][has_l], np.nonzero(has_l)[0] + 1)      # pre-order: left child is node + 1
    leaf = (t["left"] < 0) & (t["right"] < 0) & (t["count"] < 4)
    assert leaf.sum() > 0
    inner = ~leaf
    ax = t["axis"][inner]
    box = t["box"][inner]
    mid = (box[np.arange(len(ax)), ax] + box[np.arange(len(ax)), 3 + ax]) * 0.5
    assert np.array_equal(t["split"][inner], mid)                          # kdtree.cpp:82
    kids = np.concatenate([t["left"][has_l], t["right"][t["right"] >= 0]])
    par = np.concatenate([np.nonzero(has_l)[0], np.nonzero(t["right"] >= 0)[0]])
    assert np.array_equal(t["axis"][kids], (t["axis"][par] + 1) % 3)       # kdtree.cpp:94-95
    assert t["axis"][0] == 0 and n == len(set(kids.tolist())) + 1


def test_empty_scene_has_one_empty_leaf(ra):
    hs = ra.HostScene({"meshes": [], "spheres": []})
    t = hs.tree_arrays()
    assert len(t["axis"]) == 1 and t["count"][0] == 0 and t["depth"] == 1
    assert np.array_equal(t["box"][0], [1e10] * 3 + [-1e10] * 3)           # kdtree.h:36-40


def test_tonemap_matches_drawpixel(ra, po):
    rs = np.random.RandomState(0)
    rgb = np.concatenate([rs.uniform(-0.2, 1.3, (5000, 3)), [[0, 0, 0], [1, 1, 1], [0.5, 0.25, 2.0]]])
    got = ra.tonemap(rgb)
    assert np.array_equal(got, po.draw_pixels(rgb))
    c = np.clip(rgb, 0, 1)
    want = (c ** (1 / 2.2) * 255 + .5).astype(np.int64)[:, ::-1]           # raytrace.cpp:33-39, BGR order
    assert np.abs(got.astype(np.int64) - want).max() <= 1                  # libm pow vs numpy pow: last-ulp cases only


def test_bmp_roundtrip(ra, tmp_path):
    rs = np.random.RandomState(1)
    img = rs.randint(0, 256, (5, 7, 3)).astype(np.uint8)                    # 7*3 = 21 bytes per row -> padded to 24
    p = tmp_path / "o.bmp"
    ra.write_bmp(p, img)
    raw = p.read_bytes()
    assert raw[:2] == b"BM" and struct.unpack("<I", raw[2:6])[0] == len(raw) == 54 + 24 * 5
    w, h, planes, bpp = struct.unpack("<iiHH", raw[18:30])
    assert (w, h, planes, bpp) == (7, 5, 1, 24)
    rows = [np.frombuffer(raw[54 + 24 * r: 54 + 24 * r + 21], np.uint8).reshape(7, 3) for r in range(5)]
    assert np.array_equal(np.array(rows[::-1]), img)                        # stored bottom-up


def test_scene_file_roundtrip(ra, tmp_path):
    from raytrace_amd import scenes
    sc = scenes.cornell_demo(sizex=30, sizey=40)
    path = scenes.write_scene_files(sc, str(tmp_path), c_samps=3, g_samps=5)
    a = ra.HostScene.from_file(path)
    b = ra.HostScene(sc)
    st = a.settings
    assert (st.sizex, st.sizey, st.c_samps, st.g_samps, st.num_per_node, st.cu_stack_size) == (30, 40, 3, 5, 1, 4096)
    assert list(st.view) == list(sc["camera"]["o"]) and list(st.dir) == list(sc["camera"]["d"])
    ta, tb = a.tree_arrays(), b.tree_arrays()
    for k in ta:
        assert np.array_equal(ta[k], tb[k]), k
    da, db = a.desc.contents, b.desc.contents
    assert da.n_spheres == db.n_spheres == 9
    for f in ("sph_rad", "sph_pos", "sph_lig", "sph_col", "tri_col", "tri_lig"):
        n = da.n_spheres * (1 if f == "sph_rad" else 3) if f.startswith("sph") else da.n_tris * 3
        assert np.array_equal(np.ctypeslib.as_array(getattr(da, f), shape=(n,)), np.ctypeslib.as_array(getattr(db, f), shape=(n,))), f
    assert np.array_equal(np.ctypeslib.as_array(da.tri_type, shape=(da.n_tris,)), np.ctypeslib.as_array(db.tri_type, shape=(db.n_tris,)))


OBJ = """# cube corner
mtllib m.mtl
v 0 0 0
v 1 0 0
v 0 1 0
v 0 0 1
vn 0 0 1
vt 0.5 0.5
usemtl red
f 1 2 3
usemtl blue
f 1/1 2/1 4/1
f 1//1 3//1 4//1
usemtl nosuch
f 2/1/1 3/1/1 4/1/1 1/1/1
"""
MTL = """newmtl red
Kd 1 0 0
Ks 0 1 0
newmtl blue
Ka 9 9 9
Kd 0 0 1
Ks 0.5 0.5 0.5
"""


def test_obj_reader(ra, tmp_path):
    (tmp_path / "a.obj").write_text(OBJ)
    (tmp_path / "m.mtl").write_text(MTL)
    hs = ra.HostScene()
    hs.add_obj_file(tmp_path / "a.obj", 2.0, (10, 20, 30), (0, 0, 90), (0, 0, 0), (.3, .3, .3), ra.capi.RT_DIFF)
    hs.add_obj_file(tmp_path / "a.obj", 1.0, (0, 0, 0), (0, 0, 0), (0, 0, 0), (.3, .3, .3), ra.capi.RT_SPEC)
    hs.build()
    t = hs.tree_arrays()
    assert hs.num_tris == 8                                                 # the 4th vertex of the last face is dropped
    d = hs.desc.contents
    col = np.ctypeslib.as_array(d.tri_col, shape=(8, 3))
    # diffuse object: Kd; unknown material falls back to the object's colour (object.cpp:44-49, 75-106)
    assert np.array_equal(col[:4], [[1, 0, 0], [0, 0, 1], [0, 0, 1], [.3, .3, .3]])
    assert np.array_equal(col[4:], [[0, 1, 0], [.5, .5, .5], [.5, .5, .5], [.3, .3, .3]])  # specular object: Ks
    # scale 2, rotate 90 degrees about z (x -> y, y -> -x), translate (object.cpp:22-23, 138-140)
    v = t["verts"][0].reshape(3, 3)
    want = np.array([[0, 0, 0], [0, 2, 0], [-2, 0, 0]]) + [10, 20, 30]
    assert np.abs(v - want).max() < 1e-12
    assert np.array_equal(t["verts"][4].reshape(3, 3), [[0, 0, 0], [1, 0, 0], [0, 1, 0]])


def test_reader_errors(ra, tmp_path):
    hs = ra.HostScene()
    with pytest.raises(ra.capi.RtError, match="Can't open"):
        hs.add_obj_file(tmp_path / "missing.obj", 1, (0, 0, 0), (0, 0, 0), (0, 0, 0), (1, 1, 1), 0)
    (tmp_path / "bad.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n")
    with pytest.raises(ra.capi.RtError, match="Error face"):
        hs.add_obj_file(tmp_path / "bad.obj", 1, (0, 0, 0), (0, 0, 0), (0, 0, 0), (1, 1, 1), 0)
    (tmp_path / "neg.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf -3 -2 -1\n")
    with pytest.raises(ra.capi.RtError, match="Error face"):  # negative indices are unsupported there too
        hs.add_obj_file(tmp_path / "neg.obj", 1, (0, 0, 0), (0, 0, 0), (0, 0, 0), (1, 1, 1), 0)
    (tmp_path / "t.txt").write_text("0\n1\n1 0 0 0 0 0 0 1 1 1 DIFF\n10 10 1\n")
    with pytest.raises(ra.capi.RtError, match="settings"):
        ra.HostScene.from_file(tmp_path / "t.txt")
    with pytest.raises(ra.capi.RtError):
        ra.HostScene().desc  # not built yet


def test_scene_generators_hit_their_triangle_counts():
    from raytrace_amd import scenes
    assert scenes.num_tris(scenes.bunny_class(69451)) == 69451
    assert scenes.num_tris(scenes.sponza_class(262267)) == 262267
    assert scenes.num_tris(scenes.sponza_class(30000, detail=0.25)) == 30000
    s = scenes.random_spheres(10000)
    assert len(s["spheres"]) == 10000 and not s["meshes"]
    a, b = scenes.sponza_class(30000, detail=0.25), scenes.sponza_class(30000, detail=0.25)
    assert all(np.array_equal(x["verts"], y["verts"]) for x, y in zip(a["meshes"], b["meshes"]))  # deterministic
    # the textured variant is the same geometry; its textured meshes carry one image index and six coordinates per triangle
    t = scenes.sponza_class(30000, detail=0.25, textured=True)
    assert all(np.array_equal(x["verts"], y["verts"]) for x, y in zip(a["meshes"], t["meshes"])) and len(t["textures"]) == 3
    tex = [m for m in t["meshes"] if m.get("tri_tex") is not None]
    assert len(tex) == 4 and all(m["tri_ts"].shape == (m["verts"].shape[0], 6) and (m["tri_ts"] >= 0).all() for m in tex)


def test_obj_with_materials_textures_and_images(ra, tmp_path):
    """OBJ + MTL + image files: vt lines, v/t faces, usemtl, map_Kd (object.cpp:44-49,112-119,164-171,254,280) and
    the image reader standing in for cv::imread (BMP and binary PPM, top row first, BGR)."""
    from raytrace_amd import scenes
    sc = scenes.textured_room()
    path = scenes.write_scene_files(sc, str(tmp_path))
    hs = ra.HostScene.from_file(path)
    want = ra.HostScene(sc).tree_arrays()
    got = hs.tree_arrays()
    assert np.array_equal(got["verts"], want["verts"])
    assert np.array_equal(got["tri_tex"] >= 0, want["tri_tex"] >= 0)
    assert np.array_equal(got["tri_ts"][got["tri_tex"] >= 0], want["tri_ts"][want["tri_tex"] >= 0])
    imgs, ref_imgs = hs.textures(), sc["textures"]
    for g, w in zip(got["tri_tex"], want["tri_tex"]):  # images are numbered in loading order: compare contents
        if w >= 0:
            assert np.array_equal(imgs[g], ref_imgs[w])
    # binary PPM (RGB in the file, BGR in memory), with a comment line
    img = ref_imgs[1]
    with open(tmp_path / "a.ppm", "wb") as f:
        f.write(b"P6\n# a comment\n%d %d\n255\n" % (img.shape[1], img.shape[0]) + img[..., ::-1].tobytes())
    k = hs.load_texture_file(tmp_path / "a.ppm")
    assert np.array_equal(hs.textures()[k], img)
    with pytest.raises(ra.capi.RtError, match="image"):
        hs.load_texture_file(tmp_path / "missing.bmp")
    # a material whose image is missing stays untextured (opencvlib.cpp:13-14) and keeps its Kd colour
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 0.1 0.2 0.3\nmap_Kd nothere.bmp\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nusemtl a\nf 1/1 2/2 3/3\n")
    h2 = ra.HostScene()
    h2.add_obj_file(tmp_path / "m.obj", 1, (0, 0, 0), (0, 0, 0), (0, 0, 0), (1, 1, 1), 0)
    h2.build()
    t2 = h2.tree_arrays()
    assert t2["tri_tex"][0] == -1 and np.array_equal(t2["col"][0], [0.1, 0.2, 0.3])


def test_texture_resize_follows_the_object_scale(ra, tmp_path):
    """loadImage resizes the image by the object's scale (opencvlib.cpp:16-20): sizes are truncated products, equal
    sizes copy, a constant image stays constant, and a 2x enlargement of a ramp stays monotonic between its ends.
    (The interpolation arithmetic itself has no OpenCV here to be compared with.)"""
    from raytrace_amd import scenes
    ramp = np.zeros((8, 10, 3), np.uint8)
    ramp[..., 0] = np.arange(10)[None, :] * 20
    ramp[..., 1] = 77
    ramp[..., 2] = np.arange(8)[:, None] * 30
    scenes.write_bmp24(tmp_path / "r.bmp", ramp)
    hs = ra.HostScene()
    a = hs.load_texture_file(tmp_path / "r.bmp", 1.0)
    b = hs.load_texture_file(tmp_path / "r.bmp", 2.0)
    c = hs.load_texture_file(tmp_path / "r.bmp", 0.55)
    hs.build()
    t = hs.textures()
    assert np.array_equal(t[a], ramp)
    assert t[b].shape == (16, 20, 3) and t[c].shape == (4, 5, 3)
    assert (t[b][..., 1] == 77).all() and (t[c][..., 1] == 77).all()
    assert (np.diff(t[b][..., 0].astype(int), axis=1) >= 0).all() and t[b][0, 0, 0] == 0 and t[b][0, -1, 0] == 180
    assert (np.diff(t[b][..., 2].astype(int), axis=0) >= 0).all() and t[b][-1, 0, 2] == 210


def test_obj_reader_extensions_are_opt_in(ra, tmp_path):
    """Quads and negative indices: by default the 4th vertex is dropped and a negative index is an error, as in the
    reference (object.cpp:186-257,271-277); the flags of rth_add_obj_file_ex triangulate and resolve them."""
    (tmp_path / "q.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 2 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvt .5 2\n"
                                    "f 1/1 2/2 3/3 4/4 5/5\nf -5/-5 -4/-4 -3/-3\n")
    args = (1, (0, 0, 0), (0, 0, 0), (0, 0, 0), (1, 1, 1), 0)
    hs = ra.HostScene()
    with pytest.raises(ra.capi.RtError, match="Error face"):
        hs.add_obj_file(tmp_path / "q.obj", *args)
    hs = ra.HostScene()
    hs.add_obj_file(tmp_path / "q.obj", *args, flags=ra.capi.RTH_OBJ_NEGATIVE)
    hs.build()
    assert hs.num_tris == 2  # the pentagon's first three vertices, and the relative face
    assert np.array_equal(hs.tree_arrays()["verts"][1].reshape(3, 3), [[0, 0, 0], [1, 0, 0], [1, 1, 0]])
    hs = ra.HostScene()
    hs.add_obj_file(tmp_path / "q.obj", *args, flags=ra.capi.RTH_OBJ_NEGATIVE | ra.capi.RTH_OBJ_TRIANGULATE)
    hs.build()
    t = hs.tree_arrays()
    assert hs.num_tris == 4  # a fan of three, and the relative face
    assert np.array_equal(t["verts"][:3].reshape(3, 3, 3)[:, 0], np.zeros((3, 3)))
    assert np.array_equal(t["verts"][2].reshape(3, 3), [[0, 0, 0], [0, 1, 0], [0.5, 2, 0]])
    assert np.array_equal(t["tri_ts"][2], [0, 0, 0, 1, .5, 2])


def test_image_files_that_cannot_be_decoded_are_errors_and_headers_are_not_trusted(ra, tmp_path):
    """The reference leaves a material untextured only when the image cannot be OPENED (opencvlib.cpp:13-14); a file
    that exists is decoded by cv::imread.  A format this reader does not know therefore fails the OBJ load by name
    instead of rendering flat colours; a header that states an absurd size is refused before anything is allocated."""
    (tmp_path / "pic.png").write_bytes(b"\x89PNG\r\n\x1a\n" + b"\0" * 64)
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 0.1 0.2 0.3\nmap_Kd pic.png\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nusemtl a\nf 1/1 2/2 3/3\n")
    hs = ra.HostScene()
    with pytest.raises(ra.capi.RtError, match=r"decode image .*pic\.png.*BMP"):
        hs.add_obj_file(tmp_path / "m.obj", 1, (0, 0, 0), (0, 0, 0), (0, 0, 0), (1, 1, 1), 0)
    with pytest.raises(ra.capi.RtError, match="Can't open image"):
        hs.load_texture_file(tmp_path / "nothere.bmp")

    def bmp_header(w, h, bpp=24):
        return (b"BM" + struct.pack("<IHHI", 54, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, w, h, 1, bpp, 0, 0, 0, 0, 0, 0))

    for w, h in ((0x7FFFFFF0, 0x7FFFFFF0), (3, -0x80000000), (70000, 2), (40000, 40000)):
        (tmp_path / "huge.bmp").write_bytes(bmp_header(w, h) + b"\0" * 64)
        with pytest.raises(ra.capi.RtError, match="pixels"):
            hs.load_texture_file(tmp_path / "huge.bmp")
    (tmp_path / "huge.ppm").write_bytes(b"P6\n60000 60000\n255\n" + b"\0" * 64)
    with pytest.raises(ra.capi.RtError, match="pixels"):
        hs.load_texture_file(tmp_path / "huge.ppm")
    (tmp_path / "short.bmp").write_bytes(bmp_header(64, 64) + b"\0" * 64)  # fewer bytes than the header promises
    with pytest.raises(ra.capi.RtError, match="decode"):
        hs.load_texture_file(tmp_path / "short.bmp")
    img = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)
    from raytrace_amd import scenes
    scenes.write_bmp24(tmp_path / "ok.bmp", img)
    with pytest.raises(ra.capi.RtError, match="exceeds"):
        hs.load_texture_file(tmp_path / "ok.bmp", 1e6)
    k = hs.load_texture_file(tmp_path / "ok.bmp")
    hs.build()
    assert np.array_equal(hs.textures()[k], img)
