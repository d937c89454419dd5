#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own object code
(oracle/_ref/libref_l3.so: the reference's kdtree.cpp, sphere.cpp and headers compiled unmodified).
Run in the development container, where /root/reference exists:

    python tests/golden/make_golden.py

Each .npz holds inputs (scene arrays, rays, points) and the reference's outputs for them:
    tree_*      pre-order dump of KdTree (boxes, splits, axes, children, own lists)
    trace_*     KdTree::check -> (hit id, t) and sphere_intersect -> (id, t) per ray
    nc / getn   Polygon::nc and Polygon::getn
    tri_*       Polygon::intersect (t, u, v) for explicit (triangle, ray) pairs
    sph_*       Sphere::intersect per (sphere, ray)
    aabb_*      KdTreeNode::check_aabb for explicit (box, ray) pairs
    primary_*   Ray(cam.o + d * 140, d.norm()) as g++ evaluates it (raytrace.cpp:203)
    getcol_*    Polygon::getcol(u, v) of textured and plain triangles (polygon.h:73-89), images included as data
    objload     Object::readfile (object.cpp:9-286): an OBJ + MTL pair written by this script (kept as bytes) read with
                several scale / rotation / translation / material-type settings -> per Polygon points3d, col, lig, ts, nc
Nothing of the reference's source text is stored, only numbers.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import pyoracle as po  # noqa: E402
from raytrace_amd import scenes  # noqa: E402


def rand_rays(seed, n, center, spread):
    rs = np.random.RandomState(seed)
    o = np.asarray(center, np.float64) + rs.uniform(-spread, spread, (n, 3))
    d = rs.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], 1)


def special_rays(tree, n_nodes=60, seed=9):
    """Rays starting exactly on split planes / box faces, axis-parallel and in-plane directions."""
    rs = np.random.RandomState(seed)
    out = []
    inner = np.nonzero((tree["left"] >= 0) | (tree["right"] >= 0))[0][:n_nodes]
    for n in inner:
        ax = int(tree["axis"][n])
        lo, hi = tree["box"][n, :3], tree["box"][n, 3:]
        o = lo + (hi - lo) * rs.uniform(0, 1, 3)
        o[ax] = tree["split"][n]
        for sgn in (-1.0, 1.0):
            d = np.zeros(3); d[ax] = sgn
            out.append(np.concatenate([o, d]))
            d = np.zeros(3); d[(ax + 1) % 3] = sgn
            out.append(np.concatenate([o, d]))
            d = rs.normal(size=3); d[ax] = 0; d /= np.linalg.norm(d)
            out.append(np.concatenate([o, d]))
            o2 = o.copy(); o2[(ax + 1) % 3] = lo[(ax + 1) % 3]  # also on a box face
            d = rs.normal(size=3); d /= np.linalg.norm(d)
            out.append(np.concatenate([o2, d]))
    return np.array(out)


def scene_arrays(sc):
    """Flat arrays describing a scene dict, so a fixture is self-contained."""
    out = {"n_meshes": np.int32(len(sc["meshes"]))}
    for i, m in enumerate(sc["meshes"]):
        out[f"m{i}_verts"] = m["verts"]
        out[f"m{i}_lig"] = m["lig"]
        out[f"m{i}_col"] = m["col"]
        out[f"m{i}_type"] = np.int32(m["type"])
        if "tri_cols" in m:
            out[f"m{i}_tri_cols"] = m["tri_cols"]
        if m.get("tri_tex") is not None:
            out[f"m{i}_tri_tex"] = np.asarray(m["tri_tex"], np.int32)
            out[f"m{i}_tri_ts"] = np.asarray(m["tri_ts"], np.float64)
    out["n_textures"] = np.int32(len(sc.get("textures", [])))
    for k, t in enumerate(sc.get("textures", [])):
        out[f"tex{k}"] = np.asarray(t, np.uint8)
    sp = sc["spheres"]
    out["sph_rad"] = np.array([s["rad"] for s in sp], np.float64)
    out["sph_pos"] = np.array([s["pos"] for s in sp], np.float64).reshape(-1, 3)
    out["sph_lig"] = np.array([s["lig"] for s in sp], np.float64).reshape(-1, 3)
    out["sph_col"] = np.array([s["col"] for s in sp], np.float64).reshape(-1, 3)
    out["sph_type"] = np.array([s["type"] for s in sp], np.int32)
    return out


def make(name, sc, center, spread, n_rays):
    ref = po.RefScene(sc)
    tree = ref.tree_dump()
    rays = np.concatenate([rand_rays(17, n_rays, center, spread), special_rays(tree)]) if len(tree["axis"]) > 1 else rand_rays(17, n_rays, center, spread)
    tid, tt, sid, st = ref.trace(rays)
    ntri = scenes.num_tris(sc)
    rs = np.random.RandomState(23)
    fx = dict(scene_arrays(sc))
    fx.update({f"tree_{k}": v for k, v in tree.items()})
    fx.update(rays=rays, trace_tri_id=tid, trace_tri_t=tt, trace_sph_id=sid, trace_sph_t=st)
    if ntri:
        pick = rs.randint(0, ntri, min(ntri, 200))
        fx["nc_gid"] = pick.astype(np.int32)
        fx["nc"] = np.array([ref.poly_nc(g) for g in pick])
        pts = center + rs.uniform(-spread, spread, (len(pick), 3))
        fx["getn_x"] = pts
        fx["getn"] = np.array([ref.poly_getn(g, x) for g, x in zip(pick, pts)])
        # triangle tests: rays aimed at (and near) the chosen triangles
        verts = np.concatenate([m["verts"] for m in sc["meshes"]])
        tr, tout = [], []
        for g in pick:
            v = verts[g].reshape(3, 3)
            w = rs.dirichlet([1, 1, 1]) * rs.choice([1.0, 1.0, 1.2])
            target = (w[:, None] * v).sum(0)
            o = target + rs.normal(size=3) * spread
            d = target - o
            d /= np.linalg.norm(d)
            r = np.concatenate([o, d])
            tr.append(r)
            tout.append(ref.poly_intersect(g, r))
        fx["tri_rays"] = np.array(tr)
        fx["tri_tuv"] = np.array(tout)
    if sc.get("textures"):  # getcol for (u, v) inside the triangle, on every triangle (textured or not)
        gid = np.repeat(np.arange(ntri, dtype=np.int32), 6)
        u = rs.uniform(0, 1, len(gid))
        v = rs.uniform(0, 1, len(gid)) * (1 - u)
        u[::6], v[::6] = 0.0, 0.0  # the corners and an edge point
        u[1::6], v[1::6] = 1.0, 0.0
        u[2::6], v[2::6] = 0.0, 1.0
        fx["getcol_gid"], fx["getcol_uv"] = gid, np.stack([u, v], 1)
        fx["getcol"] = np.array([ref.poly_getcol(g, a, b) for g, a, b in zip(gid, u, v)])
    if len(sc["spheres"]):
        ns = len(sc["spheres"])
        si = rs.randint(0, ns, 200).astype(np.int32)
        sr = rand_rays(31, 200, center, spread)
        fx["sph_idx"] = si
        fx["sph_rays"] = sr
        fx["sph_t"] = np.array([ref.sph_intersect(i, r) for i, r in zip(si, sr)])
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **fx)
    print(name, "rays", len(rays), "tri hits", int((tid >= 0).sum()), "sphere hits", int((sid >= 0).sum()), "nodes", len(tree["axis"]))


def make_aabb():
    rs = np.random.RandomState(5)
    boxes, rays, res = [], [], []
    for _ in range(3000):
        lo = rs.uniform(-10, 10, 3)
        hi = lo + rs.uniform(0, 8, 3) * rs.choice([0.0, 1.0], 3, p=[0.1, 0.9])  # some flat boxes
        kind = rs.randint(0, 5)
        o = rs.uniform(-20, 20, 3)
        d = rs.normal(size=3)
        if kind == 1:  # axis parallel
            d = np.zeros(3); d[rs.randint(0, 3)] = rs.choice([-1.0, 1.0])
        elif kind == 2:  # origin on a face plane
            a = rs.randint(0, 3); o[a] = (lo, hi)[rs.randint(0, 2)][a]
        elif kind == 3:  # origin inside, within eps of a face
            o = lo + (hi - lo) * rs.uniform(0, 1, 3); a = rs.randint(0, 3); o[a] = lo[a] - rs.choice([0.0, 5e-6, 1e-5, 2e-5])
        elif kind == 4:  # tiny direction component
            d[rs.randint(0, 3)] = rs.choice([1e-6, -1e-6, 1e-5, 9.9e-6, 1.1e-5])
        d = d / np.linalg.norm(d)
        b, r = np.concatenate([lo, hi]), np.concatenate([o, d])
        boxes.append(b); rays.append(r); res.append(po.ref_check_aabb(b, r))
    np.savez_compressed(os.path.join(HERE, "aabb.npz"), boxes=np.array(boxes), rays=np.array(rays), hit=np.array(res, np.int8))
    print("aabb", len(res), "hits", int(np.sum(res)))


def make_primary():
    rs = np.random.RandomState(8)
    o = rs.uniform(-100, 400, (200, 3))
    d = rs.normal(size=(200, 3)) * rs.uniform(0.5, 2.0, (200, 1))
    out = np.array([po.ref_primary_ray(a, b) for a, b in zip(o, d)])
    np.savez_compressed(os.path.join(HERE, "primary.npz"), camo=o, d=d, ray=out)
    print("primary", len(out))


def objload_files(seed=3):
    """An OBJ + MTL pair of this script's own making: comments, v / vn / vt lines, all four face forms the reference
    accepts with normals present (v//n, v/t/n; plain v and v/t faces index vn[-1] there, undefined behaviour), a face
    with a fourth vertex (dropped, object.cpp:186-257), groups, materials with Ka / Kd / Ks / Ns, a material named
    twice (the last one wins, object.cpp:48), an unknown material (slot 0 = the object's colour); no images."""
    rs = np.random.RandomState(seed)
    nv, nn, nt = 40, 17, 23
    v, vn, vt = rs.uniform(-3, 3, (nv, 3)), rs.normal(size=(nn, 3)), rs.uniform(0, 1, (nt, 2))
    f3 = lambda x: " ".join(repr(float(a)) for a in x)  # noqa: E731
    mtl = ("# materials\nnewmtl red\nKa 1 1 1\nKd 0.9 0.1 0.2\nKs 0.3 0.4 0.5\nNs 10\nd 1\n\nnewmtl blue\nKd 0.1 0.2 0.95\nKs 0.6 0.7 0.8\nillum 2\n"
           "newmtl red\nKs 0.25 0.5 0.75\nKd 0.125 0.0625 0.5\n")
    obj = "# an object\nmtllib objload.mtl\n" + "".join(f"v {f3(x)}\n" for x in v) + "".join(f"vn {f3(x)}\n" for x in vn)
    obj += "".join(f"vt {f3(x)}\n" for x in vt) + "o thing\ng part1\ns 1\n"
    for i in range(30):
        a, b, c, d = (int(x) + 1 for x in rs.randint(0, nv, 4))
        n = [int(x) + 1 for x in rs.randint(0, nn, 4)]
        t = [int(x) + 1 for x in rs.randint(0, nt, 4)]
        if i % 7 == 0:
            obj += ("usemtl " + ["red", "blue", "nothere", "red"][(i // 7) % 4] + "\n")
        if i == 12:
            obj += "g part2\n# half way\n"
        if i % 3 == 0:
            obj += f"f {a}//{n[0]} {b}//{n[1]} {c}//{n[2]}\n"
        elif i % 3 == 1:
            obj += f"f {a}/{t[0]}/{n[0]} {b}/{t[1]}/{n[1]} {c}/{t[2]}/{n[2]}\n"
        else:
            obj += f"f {a}/{t[0]}/{n[0]} {b}/{t[1]}/{n[1]} {c}/{t[2]}/{n[2]} {d}/{t[3]}/{n[3]}\n"
    return obj, mtl


OBJLOAD_CASES = [  # scale, translation, rotation in degrees (about x, y, z), lig, col, type
    (1.0, (0, 0, 0), (0, 0, 0), (0, 0, 0), (.3, .35, .4), 0),
    (2.5, (10, 20, 30), (0, 0, 90), (.5, .25, .125), (.3, .35, .4), 0),
    (0.37, (-1.5, 2.25, 7), (30, -45, 60), (0, 0, 0), (.999, .999, .999), 1),
    (12.0, (50, -3, 81.5), (12.5, 200, -33), (4, 4, 4), (.1, .2, .3), 2),
]


def make_objload():
    import tempfile
    obj, mtl = objload_files()
    d = tempfile.mkdtemp(prefix="objload_")
    with open(os.path.join(d, "objload.obj"), "w") as f:
        f.write(obj)
    with open(os.path.join(d, "objload.mtl"), "w") as f:
        f.write(mtl)
    fx = {"obj": np.frombuffer(obj.encode(), np.uint8), "mtl": np.frombuffer(mtl.encode(), np.uint8), "n_cases": np.int32(len(OBJLOAD_CASES)),
          "params": np.array([[c[0], *c[1], *c[2], *c[3], *c[4], c[5]] for c in OBJLOAD_CASES], np.float64)}
    for k, (times, loc, rot, lig, col, typ) in enumerate(OBJLOAD_CASES):
        r = po.RefScene().readfile(os.path.join(d, "objload.obj"), times, loc, rot, lig, col, typ)
        assert r is not None and not r["has_tex"].any()
        for name in ("verts", "col", "lig", "ts", "nc", "type"):
            fx[f"c{k}_{name}"] = r[name]
    np.savez_compressed(os.path.join(HERE, "objload.npz"), **fx)
    print("objload", len(OBJLOAD_CASES), "cases,", len(fx["c0_verts"]), "polygons each")


if __name__ == "__main__":
    po.build()
    assert po.ref_available(), "oracle/_ref/libref_l3.so is missing (needs /root/reference)"
    make("tiny0", scenes.tiny_mixed(0), np.zeros(3), 15.0, 1500)
    make("tiny5", scenes.tiny_mixed(5, n_tris=200, n_sph=9), np.zeros(3), 15.0, 1500)
    make("cornell", scenes.cornell_demo(), np.array([50, 40, 80.0]), 40.0, 1500)
    make("bunny2k", scenes.bunny_class(2000), np.array([50, 30, 80.0]), 50.0, 1500)
    make("sponza12k", scenes.sponza_class(12000, detail=0.12), np.array([50, 40, 130.0]), 60.0, 1000)
    make("texroom", scenes.textured_room(), np.array([20, 30, 20.0]), 50.0, 1000)
    make("deeptree", scenes.deep_tree(), np.array([250, 250, 250.0]), 450.0, 1500)  # a 42-level tree
    make_aabb()
    make_primary()
    make_objload()
