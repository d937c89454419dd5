"""Seeded procedural scenes in the reference's data model (mesh objects of triangles + spheres +
the settings lines of its scene file, raytrace.cpp:106-151).  The reference ships no scene file
and no assets, and there is no network, so these stand in for the demo / Bunny / random-spheres /
Sponza / Lucy configurations at the same primitive counts.

A scene is a dict:
    meshes  : list of {verts (n,9) float64, lig (3,), col (3,), type, [tri_cols (n,3)],
                       [tri_tex (n,) int32 image index or -1, tri_ts (n,6) texture coordinates]}
    textures: optional list of uint8 [rows, cols, 3] BGR images
    spheres : list of {rad, pos (3,), lig (3,), col (3,), type}
    camera  : {o (3,), d (3,)}      sizex (rows), sizey (columns), name
Everything is generated with numpy's legacy MT19937 RandomState, which is stable across numpy
versions, and in float64.
"""
import math

import numpy as np

DIFF, SPEC, REFR = 0, 1, 2


# ----------------------------------------------------------------------------- mesh helpers

def _tris(a, b, c):
    return np.concatenate([a, b, c], axis=-1).reshape(-1, 9)


def grid(p0, du, dv, nu, nv, height=None):
    """nu x nv quads spanning p0 + [0,1]*du + [0,1]*dv, 2 triangles each; `height(u,v)` optionally
    displaces along du x dv."""
    p0, du, dv = (np.asarray(x, dtype=np.float64) for x in (p0, du, dv))
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    U, Vv = np.meshgrid(u, v, indexing="ij")
    P = p0 + U[..., None] * du + Vv[..., None] * dv
    if height is not None:
        n = np.cross(du, dv)
        n = n / np.linalg.norm(n)
        P = P + height(U, Vv)[..., None] * n
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    return np.concatenate([_tris(a, b, c), _tris(a, c, d)], axis=0)


def box(lo, hi):
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    e = hi - lo
    ex, ey, ez = np.array([e[0], 0, 0]), np.array([0, e[1], 0]), np.array([0, 0, e[2]])
    return np.concatenate([
        grid(lo, ex, ey, 1, 1), grid(lo + ez, ex, ey, 1, 1),
        grid(lo, ey, ez, 1, 1), grid(lo + ex, ey, ez, 1, 1),
        grid(lo, ez, ex, 1, 1), grid(lo + ey, ez, ex, 1, 1)], axis=0)


def rotate(tris, deg_xyz):
    rx, ry, rz = (math.radians(a) for a in deg_xyz)
    Rx = np.array([[1, 0, 0], [0, math.cos(rx), -math.sin(rx)], [0, math.sin(rx), math.cos(rx)]])
    Ry = np.array([[math.cos(ry), 0, math.sin(ry)], [0, 1, 0], [-math.sin(ry), 0, math.cos(ry)]])
    Rz = np.array([[math.cos(rz), -math.sin(rz), 0], [math.sin(rz), math.cos(rz), 0], [0, 0, 1]])
    R = Rx @ Ry @ Rz
    return (tris.reshape(-1, 3) @ R.T).reshape(-1, 9)


def translate(tris, t):
    return (tris.reshape(-1, 3) + np.asarray(t, np.float64)).reshape(-1, 9)


def fan(center, radius, k, y_axis=1):
    """k triangles around `center` in the plane perpendicular to axis y_axis."""
    ang = 2 * math.pi * np.arange(k + 1) / k
    a0, a1 = [(1, 2), (2, 0), (0, 1)][y_axis]
    rim = np.zeros((k + 1, 3))
    rim[:, a0] = radius * np.cos(ang)
    rim[:, a1] = radius * np.sin(ang)
    rim += np.asarray(center, np.float64)
    c = np.broadcast_to(np.asarray(center, np.float64), (k, 3))
    return _tris(c, rim[:-1], rim[1:])


def _bumps(rs, n_terms, amp):
    f = rs.uniform(-1, 1, (n_terms, 3)) * rs.uniform(1.5, 9.0, (n_terms, 1))
    ph = rs.uniform(0, 2 * math.pi, n_terms)
    a = amp * rs.uniform(0.3, 1.0, n_terms) / n_terms

    def h(d):
        return (np.sin(d @ f.T + ph) * a).sum(-1)
    return h


def blob(center, radius, bands, segs, seed=0, amp=0.35, squash=(1.0, 1.0, 1.0)):
    """Sphere of 2*segs*(bands-1) triangles whose radius is modulated by a seeded sum of sines."""
    rs = np.random.RandomState(seed)
    h = _bumps(rs, 12, amp)
    th = np.pi * np.arange(bands + 1) / bands
    ph = 2 * np.pi * np.arange(segs + 1) / segs
    ph[-1] = 0.0  # close the seam exactly
    T, Pp = np.meshgrid(th, ph, indexing="ij")
    d = np.stack([np.sin(T) * np.cos(Pp), np.cos(T), np.sin(T) * np.sin(Pp)], -1)
    d[0] = [0, 1, 0]
    d[-1] = [0, -1, 0]
    P = np.asarray(center, np.float64) + radius * (1.0 + h(d))[..., None] * d * np.asarray(squash, np.float64)
    top = _tris(P[0, :-1], P[1, 1:], P[1, :-1])
    bot = _tris(P[-1, :-1], P[-2, :-1], P[-2, 1:])
    a, b, c, e = P[1:-2, :-1], P[2:-1, :-1], P[2:-1, 1:], P[1:-2, 1:]
    return np.concatenate([top, _tris(a, c, b), _tris(a, e, c), bot], axis=0)


def cylinder(base, radius, height, segs, stacks, flute=0.0, flutes=16):
    """Open cylinder along +y: 2*segs*stacks triangles; optional fluting of the radius."""
    ph = 2 * np.pi * np.arange(segs + 1) / segs
    ph[-1] = 0.0
    y = np.linspace(0, height, stacks + 1)
    Y, Pp = np.meshgrid(y, ph, indexing="ij")
    r = radius * (1.0 + flute * np.cos(flutes * Pp))
    P = np.stack([r * np.cos(Pp), Y, r * np.sin(Pp)], -1) + np.asarray(base, np.float64)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    return np.concatenate([_tris(a, b, c), _tris(a, c, d)], axis=0)


def arch(p_from, p_to, rise, tube, arc_segs, tube_segs):
    """Half-torus arch from p_from to p_to (same height), bulging up by `rise`."""
    p_from, p_to = np.asarray(p_from, np.float64), np.asarray(p_to, np.float64)
    mid, half = 0.5 * (p_from + p_to), 0.5 * (p_to - p_from)
    L = np.linalg.norm(half)
    ax = half / L
    up = np.array([0.0, 1.0, 0.0])
    side = np.cross(ax, up)
    t = np.pi * np.arange(arc_segs + 1) / arc_segs
    s = 2 * np.pi * np.arange(tube_segs + 1) / tube_segs
    s[-1] = 0.0
    T, S = np.meshgrid(t, s, indexing="ij")
    cx = -np.cos(T)[..., None] * ax * L + (np.sin(T) * rise)[..., None] * up
    nrm = (-np.cos(T) * rise)[..., None] * ax * (1.0 / L) + (np.sin(T))[..., None] * up
    nrm = nrm / np.linalg.norm(nrm, axis=-1, keepdims=True)
    P = mid + cx + tube * (np.cos(S)[..., None] * nrm + np.sin(S)[..., None] * side)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    return np.concatenate([_tris(a, b, c), _tris(a, c, d)], axis=0)


def _mesh(verts, col, type_=DIFF, lig=(0, 0, 0), tri_cols=None):
    m = {"verts": np.ascontiguousarray(verts, np.float64).reshape(-1, 9), "lig": np.asarray(lig, np.float64),
         "col": np.asarray(col, np.float64), "type": int(type_)}
    if tri_cols is not None:
        m["tri_cols"] = np.ascontiguousarray(tri_cols, np.float64)
    return m


def _sphere(rad, pos, lig, col, type_):
    return {"rad": float(rad), "pos": np.asarray(pos, np.float64), "lig": np.asarray(lig, np.float64),
            "col": np.asarray(col, np.float64), "type": int(type_)}


def num_tris(scene):
    return int(sum(m["verts"].shape[0] for m in scene["meshes"]))


# ----------------------------------------------------------------------------- scenes

def _cornell_spheres():
    """The nine spheres of the smallpt Cornell box the reference's result images show."""
    return [
        _sphere(1e5, (1e5 + 1, 40.8, 81.6), (0, 0, 0), (.75, .25, .25), DIFF),    # left
        _sphere(1e5, (-1e5 + 99, 40.8, 81.6), (0, 0, 0), (.25, .25, .75), DIFF),  # right
        _sphere(1e5, (50, 40.8, 1e5), (0, 0, 0), (.75, .75, .75), DIFF),          # back
        _sphere(1e5, (50, 40.8, -1e5 + 170), (0, 0, 0), (0, 0, 0), DIFF),         # front
        _sphere(1e5, (50, 1e5, 81.6), (0, 0, 0), (.75, .75, .75), DIFF),          # bottom
        _sphere(1e5, (50, -1e5 + 81.6, 81.6), (0, 0, 0), (.75, .75, .75), DIFF),  # top
        _sphere(16.5, (27, 16.5, 47), (0, 0, 0), (.999, .999, .999), SPEC),       # mirror
        _sphere(16.5, (73, 16.5, 78), (0, 0, 0), (.999, .999, .999), REFR),       # glass
        _sphere(600, (50, 681.6 - .27, 81.6), (12, 12, 12), (0, 0, 0), DIFF),     # light
    ]


def cornell_demo(sizex=600, sizey=800, mesh_bands=22, mesh_segs=24):
    """Cornell box of 9 spheres + a 12-triangle glass cube + a ~1k-triangle diffuse blob."""
    cube = translate(rotate(box((-7, -7, -7), (7, 7, 7)), (0, 30, 0)), (52, 7.01, 112))
    bl = blob((28, 11, 100), 8.0, mesh_bands, mesh_segs, seed=7, amp=0.5)
    return {"name": "cornell_demo", "meshes": [_mesh(cube, (.999, .999, .999), REFR), _mesh(bl, (.85, .75, .2), DIFF)],
            "spheres": _cornell_spheres(), "camera": {"o": np.array([50, 52, 295.6]), "d": np.array([0, -0.042612, -1.0])},
            "sizex": sizex, "sizey": sizey}


def lamp_room(sizex=48, sizey=64, dome=True, lamp_rad=5.0):
    """The Cornell box with its ceiling light replaced by a lamp hanging inside the room (the scene smallpt's
    explicit-light-sampling variant uses), a second, dimmer lamp near the floor, the demo's glass cube and blob,
    and -- with `dome` -- an emissive sphere around everything with the front wall taken out: every shaded
    point is INSIDE that light, which therefore cannot be sampled over a cone."""
    sp = _cornell_spheres()[:8]
    if dome:
        del sp[3]  # the front wall: the dome is seen through the opening
    sp.append(_sphere(lamp_rad, (50, 81.6 - 16.5, 81.6), (40, 40, 40), (0, 0, 0), DIFF))
    sp.append(_sphere(3.0, (80, 6, 120), (6, 14, 20), (0, 0, 0), DIFF))
    if dome:
        sp.append(_sphere(3e5, (50, 40, 80), (.35, .4, .5), (0, 0, 0), DIFF))
    cube = translate(rotate(box((-7, -7, -7), (7, 7, 7)), (0, 30, 0)), (52, 7.01, 112))
    bl = blob((28, 11, 100), 8.0, 10, 12, seed=7, amp=0.5)
    return {"name": "lamp_room" + ("_dome" if dome else ""), "meshes": [_mesh(cube, (.999, .999, .999), REFR), _mesh(bl, (.85, .75, .2), DIFF)],
            "spheres": sp, "camera": {"o": np.array([50, 52, 295.6]), "d": np.array([0, -0.042612, -1.0])},
            "sizex": sizex, "sizey": sizey}


def _blob_counts(n_body):
    bands = max(3, int(round(math.sqrt(n_body / 4.0))))
    segs = max(3, (n_body - 3) // (2 * (bands - 1)))
    rest = n_body - 2 * segs * (bands - 1)
    if rest < 3:
        segs -= 1
        rest = n_body - 2 * segs * (bands - 1)
    return bands, segs, rest


def bunny_class(n_tris=69451, sizex=1080, sizey=1920, seed=1):
    """A displaced-sphere 'statue' of exactly n_tris triangles in total (with its 2-triangle floor
    and a pedestal fan), lit by one emissive sphere: stand-in for the Stanford Bunny / Lucy."""
    bands, segs, rest = _blob_counts(n_tris - 2)
    body = blob((50, 30, 80), 20.0, bands, segs, seed=seed, amp=0.45, squash=(1.0, 1.25, 0.9))
    ped = fan((50, 0.5, 80), 30.0, rest)
    floor = grid((-400, 0, -300), (0, 0, 800), (900, 0, 0), 1, 1)
    scene = {"name": f"bunny_class_{n_tris}",
             "meshes": [_mesh(floor, (.6, .6, .6), DIFF), _mesh(ped, (.3, .5, .3), DIFF), _mesh(body, (.75, .7, .6), DIFF)],
             "spheres": [_sphere(40, (60, 150, 140), (12, 12, 12), (0, 0, 0), DIFF)],
             "camera": {"o": np.array([50, 52, 295.6]), "d": np.array([0, -0.12, -1.0])}, "sizex": sizex, "sizey": sizey}
    assert num_tris(scene) == n_tris
    return scene


def lucy_class(n_tris=28055742, sizex=2160, sizey=3840, seed=2):
    s = bunny_class(n_tris, sizex, sizey, seed)
    s["name"] = f"lucy_class_{n_tris}"
    return s


def random_spheres(n=10000, sizex=1080, sizey=1920, seed=42):
    """n spheres, centres uniform in a 200^3 box, radii uniform in [0.5, 2]: 50 % mirror, 40 % glass,
    10 % diffuse; the first four are large emitters."""
    rs = np.random.RandomState(seed)
    pos = rs.uniform(-100, 100, (n, 3)) + np.array([50.0, 50.0, 0.0])
    rad = rs.uniform(0.5, 2.0, n)
    kind = rs.uniform(0, 1, n)
    cols = rs.uniform(0.2, 0.9, (n, 3))
    sph = []
    for i in range(n):
        if i < 4:
            sph.append(_sphere(12.0, pos[i], (12, 12, 12), (0, 0, 0), DIFF))
        elif kind[i] < 0.5:
            sph.append(_sphere(rad[i], pos[i], (0, 0, 0), (.999, .999, .999), SPEC))
        elif kind[i] < 0.9:
            sph.append(_sphere(rad[i], pos[i], (0, 0, 0), (.999, .999, .999), REFR))
        else:
            sph.append(_sphere(rad[i], pos[i], (0, 0, 0), cols[i], DIFF))
    return {"name": f"random_spheres_{n}", "meshes": [], "spheres": sph,
            "camera": {"o": np.array([50, 50, 300.0]), "d": np.array([0, 0, -1.0])}, "sizex": sizex, "sizey": sizey}


def sponza_class(n_tris=262267, sizex=2160, sizey=3840, seed=3, detail=1.0, textured=False):
    """A closed colonnaded atrium of exactly n_tris triangles (axis-heavy walls, slabs and floor
    plus fluted columns, arches, hanging drapes, urns and a relief), lit by 2 emissive spheres:
    stand-in for Crytek Sponza.  `detail` scales the tessellation of everything but the floor,
    which absorbs the remainder so the total is exact.  `textured`: the same geometry with images
    (test_images) on the floor, the ceiling and the two long walls (Polygon::getcol, polygon.h:73-89)."""
    rs = np.random.RandomState(seed)
    q = lambda v: max(2, int(round(v * detail)))  # noqa: E731
    X0, X1, Y0, Y1, Z0, Z1 = -20.0, 120.0, 0.0, 80.0, 0.0, 280.0  # every camera ray starts inside (z <= 275.6)
    W, Hh, Ln = X1 - X0, Y1 - Y0, Z1 - Z0
    meshes = []
    stone, warm, dark = (.72, .68, .6), (.8, .6, .45), (.35, .3, .28)
    # shell
    meshes.append(_mesh(grid((X0, Y1, Z0), (0, 0, Ln), (W, 0, 0), q(40), q(24)), stone))            # ceiling
    meshes.append(_mesh(grid((X0, Y0, Z0), (0, Hh, 0), (0, 0, Ln), q(16), q(48)), (.7, .35, .3)))   # left wall
    meshes.append(_mesh(grid((X1, Y0, Z0), (0, 0, Ln), (0, Hh, 0), q(48), q(16)), (.3, .4, .7)))    # right wall
    meshes.append(_mesh(grid((X0, Y0, Z0), (W, 0, 0), (0, Hh, 0), q(24), q(16)), stone))            # back wall
    meshes.append(_mesh(grid((X0, Y0, Z1), (0, Hh, 0), (W, 0, 0), q(16), q(24)), dark))             # front wall
    # galleries (slabs) along both sides
    for xa, xb in ((X0, 14.0), (86.0, X1)):
        lo, hi = np.array([xa, 50.0, Z0 + 1]), np.array([xb, 54.0, Z1 - 1])
        e = hi - lo
        meshes.append(_mesh(np.concatenate([
            grid(lo, (e[0], 0, 0), (0, 0, e[2]), q(6), q(40)), grid(lo + (0, e[1], 0), (0, 0, e[2]), (e[0], 0, 0), q(40), q(6)),
            grid(lo + ((e[0], 0, 0) if xa == X0 else (0, 0, 0)), (0, e[1], 0), (0, 0, e[2]), 2, q(40))]), warm))
    # colonnades, arches, upper columns
    zs = np.arange(20.0, 241.0, 22.0)
    lower, upper, arches = [], [], []
    for x in (10.0, 90.0):
        for i, z in enumerate(zs):
            lower.append(cylinder((x, 0, z), 4.0, 50.0, q(48), q(32), flute=0.04, flutes=12))
            upper.append(cylinder((x, 54, z), 2.5, 26.0, q(32), q(16), flute=0.05, flutes=8))
            if i + 1 < len(zs):
                arches.append(arch((x, 44, z), (x, 44, zs[i + 1]), 6.0, 1.5, q(24), q(16)))
    meshes.append(_mesh(np.concatenate(lower), stone))
    meshes.append(_mesh(np.concatenate(upper), stone))
    meshes.append(_mesh(np.concatenate(arches), warm))
    # hanging drapes (wavy sheets across the nave)
    drape_cols = [(.75, .2, .2), (.2, .55, .3), (.25, .3, .7), (.8, .7, .25), (.6, .3, .6), (.3, .65, .7)]
    for k, z in enumerate(np.linspace(45.0, 215.0, 6)):
        ph = rs.uniform(0, 6.28)
        hfn = lambda U, V, ph=ph: 2.5 * np.sin(9 * U + ph) * (0.3 + V) + 1.2 * np.sin(5 * V + 2 * ph)  # noqa: E731
        meshes.append(_mesh(grid((22, 40, z), (56, 0, 0), (0, 32, 0), q(64), q(96), height=hfn), drape_cols[k]))
    # urns: glass and mirror
    for k, (x, z) in enumerate([(30, 40), (70, 62), (30, 106), (70, 128), (30, 172), (70, 194), (50, 84), (50, 150)]):
        t = (REFR, SPEC, DIFF)[k % 3]
        c = (.999, .999, .999) if t != DIFF else (.7, .5, .3)
        meshes.append(_mesh(blob((x, 6.2, z), 6.0, q(32), q(64), seed=100 + k, amp=0.25, squash=(1, 1.0, 1)), c, t))
    # relief on the back wall
    meshes.append(_mesh(blob((50, 38, 6), 18.0, q(72), q(140), seed=55, amp=0.6, squash=(1.2, 1.0, 0.25)), (.78, .72, .62)))
    used = sum(m["verts"].shape[0] for m in meshes)
    rest = n_tris - used
    if rest < 2 * 4 * 4 + 3:
        raise ValueError(f"n_tris={n_tris} is too small for detail={detail} ({used} triangles already)")
    nu = max(4, int(math.sqrt((rest - 3) / 2.0 * (W / Ln))))
    nv = (rest - 3) // (2 * nu)
    k = rest - 2 * nu * nv
    while k < 3:
        nv -= 1
        k = rest - 2 * nu * nv
    floor = grid((X0, Y0, Z0), (0, 0, Ln), (W, 0, 0), nv, nu)
    chk = ((np.arange(nv)[:, None] + np.arange(nu)[None, :]) % 2).reshape(-1)
    fc = np.where(np.concatenate([chk, chk])[:, None] == 0, np.array(stone), np.array((.45, .42, .4)))
    meshes.append(_mesh(floor, stone, DIFF, tri_cols=fc))
    meshes.append(_mesh(fan((50, 0.02, 131), 12.0, k), (.55, .25, .2)))
    if textured:  # (texture coordinates follow grid()'s own parametrisation; they stay non-negative)
        def put(i, nu_, nv_, tex, reps):
            n = meshes[i]["verts"].shape[0]
            meshes[i] = dict(meshes[i], tri_tex=np.full(n, tex, np.int32), tri_ts=_grid_uv(nu_, nv_, reps=reps, off=(0.25, 0.25)))
        put(0, q(40), q(24), 1, 6.0)            # ceiling
        put(1, q(16), q(48), 0, 9.0)            # left wall
        put(2, q(48), q(16), 0, 9.0)            # right wall
        put(len(meshes) - 2, nv, nu, 2, 40.0)   # floor
    scene = {"name": f"sponza_class_{n_tris}" + ("_textured" if textured else ""), "meshes": meshes,
             "spheres": [_sphere(6, (50, 70, 70), (40, 38, 34), (0, 0, 0), DIFF), _sphere(6, (50, 70, 190), (40, 38, 34), (0, 0, 0), DIFF)],
             "camera": {"o": np.array([50, 40, 400.0]), "d": np.array([0, -0.02, -1.0])}, "sizex": sizex, "sizey": sizey}
    if textured:
        scene["textures"] = test_images(seed)
    assert num_tris(scene) == n_tris, (num_tris(scene), n_tris)
    return scene


def tiny_mixed(seed=0, n_tris=40, n_sph=5):
    """Small random soup of triangles and spheres around the origin for known-answer tests."""
    rs = np.random.RandomState(seed)
    c = rs.uniform(-10, 10, (n_tris, 1, 3))
    v = (c + rs.uniform(-3, 3, (n_tris, 3, 3))).reshape(n_tris, 9)
    half = n_tris // 2
    meshes = [_mesh(v[:half], (.7, .7, .7), DIFF), _mesh(v[half:], (.999, .999, .999), REFR)]
    sph = [_sphere(rs.uniform(0.5, 3), rs.uniform(-10, 10, 3), (0, 0, 0), rs.uniform(.2, .9, 3), (DIFF, SPEC, REFR)[i % 3])
           for i in range(n_sph)]
    sph.append(_sphere(4.0, (0, 18, 0), (12, 12, 12), (0, 0, 0), DIFF))
    return {"name": f"tiny_mixed_{seed}", "meshes": meshes, "spheres": sph,
            "camera": {"o": np.array([0, 0, 165.0]), "d": np.array([0, 0, -1.0])}, "sizex": 48, "sizey": 64}


def deep_tree(levels=36, seed=4, size=1000.0, per=6, clutter=1500):
    """A few hundred triangles whose median-split kd-tree is `levels` + 3 or so nodes deep: cluster k
    sits at 0.9 of the remaining box along axis k mod 3 and everything after it below 0.4, so every
    split peels off one cluster (kdtree.cpp:82-95 splits at the middle of the node's box).  The
    clusters shrink with the box; `clutter` larger triangles spread over the whole box end up on the
    interior nodes' own lists and in the shallow leaves, so rays find hits on the way down."""
    rs = np.random.RandomState(seed)
    w = np.array([size, size, size], np.float64)
    tris = []
    for k in range(levels):
        a = k % 3
        c = rs.uniform(0.1, 0.9, 3) * w
        c[a] = 0.9 * w[a]
        h = 0.02 * w.min()
        cc = c + rs.uniform(-h, h, (per, 1, 3))
        tris.append((cc + rs.uniform(-h, h, (per, 3, 3))).reshape(per, 9))
        w[a] *= 0.4
    cc = rs.uniform(0.1, 0.9, (per, 1, 3)) * w
    tris.append((cc + rs.uniform(-0.02, 0.02, (per, 3, 3)) * w).reshape(per, 9))
    if clutter:
        cc = rs.uniform(0, 1, (clutter, 1, 3)) ** 3 * size  # denser towards the deep corner
        tris.append((cc + rs.uniform(-0.15, 0.15, (clutter, 3, 3)) * (cc + 1.0)).reshape(clutter, 9))
    return {"name": f"deep_tree_{levels}", "meshes": [_mesh(np.concatenate(tris), (.7, .7, .7), DIFF)],
            "spheres": [_sphere(0.3 * size, (0.5 * size, 1.6 * size, 0.5 * size), (12, 12, 12), (0, 0, 0), DIFF)],
            "camera": {"o": np.array([0.3 * size, 0.3 * size, 2.2 * size]), "d": np.array([-0.1, -0.1, -1.0])}, "sizex": 48, "sizey": 64}


def _grid_uv(nu, nv, reps=1.0, off=(0.0, 0.0)):
    """Texture coordinates (ts[0..2].xy per triangle) matching grid(): (u, v) of each corner times `reps` plus `off`."""
    u = np.linspace(0.0, 1.0, nu + 1) * reps + off[0]
    v = np.linspace(0.0, 1.0, nv + 1) * reps + off[1]
    U, Vv = np.meshgrid(u, v, indexing="ij")
    P = np.stack([U, Vv], -1)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    t1 = np.concatenate([a, b, c], axis=-1).reshape(-1, 6)
    t2 = np.concatenate([a, c, d], axis=-1).reshape(-1, 6)
    return np.concatenate([t1, t2], axis=0)


def test_images(seed=0):
    """Three small BGR images of different, non-square sizes (the reference's lookup transposes a non-square image,
    polygon.h:34 against opencvlib.cpp:19-29): a checker board, a colour ramp and noise."""
    rs = np.random.RandomState(seed)
    r, c = np.meshgrid(np.arange(37), np.arange(53), indexing="ij")
    checker = np.where((((r // 5) + (c // 7)) % 2)[..., None] == 0, np.array([230, 60, 40]), np.array([40, 200, 235])).astype(np.uint8)
    r, c = np.meshgrid(np.arange(64), np.arange(24), indexing="ij")
    ramp = np.stack([255 * r // 63, 255 * c // 23, 255 - 255 * r // 63], -1).astype(np.uint8)
    noise = rs.randint(0, 256, (16, 16, 3)).astype(np.uint8)
    return [checker, ramp, noise]


def textured_room(sizex=48, sizey=64, seed=0):
    """A small room whose floor, back wall and a tilted mirror panel carry images (Polygon::getcol with a texture,
    polygon.h:73-89), beside an untextured wall and a glass blob; one emissive sphere.  Texture coordinates run
    past 1 on the floor (the lookup wraps with %), all stay non-negative (a negative one is undefined in the reference)."""
    tex = test_images(seed)
    floor = grid((-40, 0, -40), (0, 0, 120), (120, 0, 0), 5, 4)
    back = grid((-40, 0, -40), (120, 0, 0), (0, 80, 0), 3, 3)
    left = grid((-40, 0, -40), (0, 80, 0), (0, 0, 120), 2, 2)
    panel = translate(rotate(grid((-15, -15, 0), (30, 0, 0), (0, 30, 0), 2, 2), (10, 35, 0)), (45, 30, 0))
    gl = blob((5, 12, 25), 10.0, 8, 10, seed=3, amp=0.2)
    n_f, n_b, n_p = floor.shape[0], back.shape[0], panel.shape[0]
    meshes = [
        dict(_mesh(floor, (.6, .6, .6), DIFF), tri_tex=np.zeros(n_f, np.int32), tri_ts=_grid_uv(5, 4, reps=2.5, off=(0.13, 0.4))),
        dict(_mesh(back, (.6, .6, .6), DIFF), tri_tex=np.where(np.arange(n_b) % 5 == 4, -1, 1).astype(np.int32), tri_ts=_grid_uv(3, 3)),
        _mesh(left, (.75, .25, .25), DIFF),
        dict(_mesh(panel, (.9, .9, .9), SPEC), tri_tex=np.full(n_p, 2, np.int32), tri_ts=_grid_uv(2, 2, reps=0.999)),
        _mesh(gl, (.999, .999, .999), REFR),
    ]
    return {"name": "textured_room", "meshes": meshes, "textures": tex,
            "spheres": [_sphere(25, (20, 120, 40), (12, 12, 12), (0, 0, 0), DIFF)],
            "camera": {"o": np.array([20, 56, 190.0]), "d": np.array([0.0, -0.15, -1.0])}, "sizex": sizex, "sizey": sizey}  # (rays start 140 units along d)


def transformed(scene, scale=1.0, shift=(0.0, 0.0, 0.0)):
    """The scene uniformly scaled about the origin, then translated (meshes, spheres and camera)."""
    shift = np.asarray(shift, np.float64)
    out = dict(scene)
    out["name"] = scene.get("name", "scene") + f"_x{scale:g}"
    out["meshes"] = []
    for m in scene["meshes"]:
        mm = dict(m)
        mm["verts"] = np.ascontiguousarray((m["verts"].reshape(-1, 3) * scale + shift).reshape(-1, 9))
        out["meshes"].append(mm)
    out["spheres"] = [dict(s, rad=s["rad"] * scale, pos=s["pos"] * scale + shift) for s in scene["spheres"]]
    if "camera" in scene:
        # (the reference's camera starts its rays 140 units along the view direction, raytrace.cpp:203; that
        # offset does not scale with the scene, so the transformed camera only sees the same picture for scale 1)
        out["camera"] = {"o": scene["camera"]["o"] * scale + shift, "d": scene["camera"]["d"].copy()}
    return out


def write_bmp24(path, bgr):
    """A 24-bit BMP (bottom-up rows padded to four bytes) from a uint8 [rows, cols, 3] BGR array."""
    import struct
    a = np.ascontiguousarray(bgr, np.uint8)
    rows, cols = a.shape[:2]
    stride = (cols * 3 + 3) & ~3
    with open(path, "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", 54 + stride * rows, 0, 0, 54))
        f.write(struct.pack("<IiiHHIIiiII", 40, cols, rows, 1, 24, 0, stride * rows, 2835, 2835, 0, 0))
        for r in range(rows - 1, -1, -1):
            f.write(a[r].tobytes() + b"\0" * (stride - cols * 3))


def write_scene_files(scene, directory, c_samps=4, g_samps=4):
    """Write `scene` as a reference-format scene file (test.txt) plus one OBJ per mesh; returns the
    path of test.txt.  Textured meshes get `vt` lines, `v/t` faces, an MTL library with one material
    per image (map_Kd for diffuse objects, map_Ks otherwise, object.cpp:113) and the images as BMP
    files.  Per-triangle colours of untextured meshes are not written (the object's colour applies)."""
    import os
    os.makedirs(directory, exist_ok=True)
    tname = {DIFF: "DIFF", SPEC: "SPEC", REFR: "REFR"}
    for k, t in enumerate(scene.get("textures", [])):
        write_bmp24(os.path.join(directory, f"tex{k}.bmp"), t)
    lines = [str(len(scene["meshes"]))]
    for i, m in enumerate(scene["meshes"]):
        v = m["verts"].reshape(-1, 3)
        tt = m.get("tri_tex")
        with open(os.path.join(directory, f"mesh{i}.obj"), "w") as f:
            f.write("# generated\n")
            if tt is not None:
                with open(os.path.join(directory, f"mesh{i}.mtl"), "w") as fm:
                    for k in sorted(set(int(x) for x in tt if x >= 0)):
                        fm.write(f"newmtl img{k}\n{'map_Kd' if m['type'] == DIFF else 'map_Ks'} tex{k}.bmp\n")
                f.write(f"mtllib mesh{i}.mtl\n")
            for p in v:
                f.write("v %s %s %s\n" % (repr(float(p[0])), repr(float(p[1])), repr(float(p[2]))))
            if tt is not None:
                for p in np.asarray(m["tri_ts"], np.float64).reshape(-1, 2):
                    f.write("vt %s %s\n" % (repr(float(p[0])), repr(float(p[1]))))
            cur = None
            for t in range(v.shape[0] // 3):
                if tt is None:
                    f.write(f"f {3 * t + 1} {3 * t + 2} {3 * t + 3}\n")
                    continue
                want = int(tt[t])
                if want != cur:
                    f.write(f"usemtl img{want}\n" if want >= 0 else "usemtl none\n")  # an unknown name selects the object's colour
                    cur = want
                f.write(f"f {3 * t + 1}/{3 * t + 1} {3 * t + 2}/{3 * t + 2} {3 * t + 3}/{3 * t + 3}\n")
        lig, col = m["lig"], m["col"]
        lines.append(f"mesh{i}.obj 1 0 0 0 0 0 0 " + " ".join(repr(float(x)) for x in list(lig) + list(col)) + " " + tname[m["type"]])
    lines.append(str(len(scene["spheres"])))
    for s in scene["spheres"]:
        lines.append(" ".join(repr(float(x)) for x in [s["rad"], *s["pos"], *s["lig"], *s["col"]]) + " " + tname[s["type"]])
    lines.append(f"{scene['sizex']} {scene['sizey']} {c_samps} {g_samps} 1 4096")
    lines.append(" ".join(repr(float(x)) for x in scene["camera"]["o"]))
    lines.append(" ".join(repr(float(x)) for x in scene["camera"]["d"]))
    path = os.path.join(directory, "test.txt")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return path
