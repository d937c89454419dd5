"""ctypes bindings for the CPU oracle (liboracle.so) and, where it was built, the reference's
own acceleration/intersection object code (_ref/libref_l3.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by bench.py's
cpu_baseline leg as the checker / the reported CPU baseline.  The product package never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MISS_ID = -2147483648

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


class OStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64), ("sph_tests", C.c_uint64),
                ("tri_shades", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class OParams(C.Structure):
    _fields_ = [("samps", C.c_int32), ("rr_depth", C.c_int32), ("rr_mode", C.c_int32), ("split_depth", C.c_int32),
                ("hard_depth", C.c_int32), ("jitter", C.c_int32), ("seed", C.c_uint64), ("light_mode", C.c_int32)]


class OCamera(C.Structure):
    _fields_ = [("o", C.c_double * 3), ("d", C.c_double * 3), ("sizex", C.c_int32), ("sizey", C.c_int32)]


def build(force=False):
    """Compile liboracle.so (and _ref/ when the reference checkout is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "rt_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/kdtree.cpp"):
        ref = os.path.join(_HERE, "_ref", "libref_l3.so")
        deps = [os.path.join(_HERE, "ref_l3_harness.cpp"), os.path.join(_HERE, "..", "integration", "rt_binding.h"),
                os.path.join(_HERE, "..", "include", "rt_hip.h")]
        if force or not os.path.exists(ref) or os.path.getmtime(ref) < max(os.path.getmtime(d) for d in deps):
            subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def build_fast():
    """The oracle compiled with the reference's own flags (-Ofast -march=native, CMakeLists.txt:30) for
    THIS machine's CPU, into a temporary directory (a -march=native binary must not travel).  Returns
    the loaded library.  Not IEEE-reproducible: used only for an informational CPU timing."""
    import tempfile
    d = tempfile.mkdtemp(prefix="oracle_fast_")
    so = os.path.join(d, "liboracle_fast.so")
    subprocess.check_call(["gcc", "-Ofast", "-march=native", "-fPIC", "-std=c11", "-D_GNU_SOURCE", "-shared", "-o", so,
                           os.path.join(_HERE, "rt_oracle.c"), "-lm", "-lpthread"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return _bind(C.CDLL(so))


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _lib = _bind(C.CDLL(so))
    return _lib


def _bind(L):
    L.orc_scene_new.restype = C.c_void_p
    L.orc_scene_free.argtypes = [C.c_void_p]
    L.orc_add_object.argtypes = [C.c_void_p, _dp, C.c_int, _dp, _dp, _dp, _dp, C.c_int]
    L.orc_add_sphere.argtypes = [C.c_void_p, C.c_double, _dp, _dp, _dp, C.c_int]
    L.orc_add_texture.argtypes = [C.c_void_p, C.POINTER(C.c_ubyte), C.c_int, C.c_int]
    L.orc_set_textures.argtypes = [C.c_void_p, C.c_int, _ip, _dp]
    L.orc_tri_getcol.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, _dp]
    L.orc_build.argtypes = [C.c_void_p]
    for f in ("orc_num_nodes", "orc_tree_depth", "orc_num_tris"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = C.c_int
    L.orc_tree_dump.argtypes = [C.c_void_p, _dp, _dp, _ip, _ip, _ip, _ip, _ip, _ip]
    L.orc_trace.argtypes = [C.c_void_p, C.c_int, _dp, _ip, _dp, _ip, _dp, C.POINTER(OStats)]
    L.orc_tri_intersect.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp]
    L.orc_tri_intersect.restype = C.c_double
    L.orc_tri_getn.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
    L.orc_tri_nc.argtypes = [C.c_void_p, C.c_int, _dp]
    L.orc_sph_intersect.argtypes = [C.c_void_p, C.c_int, _dp]
    L.orc_sph_intersect.restype = C.c_double
    L.orc_path_state.argtypes = [C.c_uint64, C.c_uint64]
    L.orc_path_state.restype = C.c_uint64
    L.orc_rng_draw.argtypes = [C.POINTER(C.c_uint64)]
    L.orc_rng_draw.restype = C.c_double
    L.orc_camera_basis.argtypes = [C.POINTER(OCamera), _dp, _dp, _dp]
    L.orc_render_tile.argtypes = [C.c_void_p, C.POINTER(OCamera), C.POINTER(OParams), C.c_int, C.c_int, C.c_int,
                                  C.c_int, _dp, _ip, C.POINTER(OStats), C.c_int]
    L.orc_radiance.argtypes = [C.c_void_p, C.POINTER(OParams), C.c_int, _dp, _dp, C.POINTER(OStats)]
    L.orc_draw_pixels.argtypes = [_dp, C.c_size_t, C.POINTER(C.c_ubyte)]
    L.orc_check_aabb.argtypes = [_dp, _dp]
    L.orc_primary_ray.argtypes = [_dp, _dp, _dp]
    return L


def make_params(samps=1, rr_depth=5, rr_mode=1, split_depth=2, hard_depth=64, jitter=1, seed=1, light_mode=0):
    return OParams(samps, rr_depth, rr_mode, split_depth, hard_depth, jitter, seed, light_mode)


def make_camera(o, d, sizex, sizey):
    cam = OCamera()
    cam.o[:] = [float(v) for v in o]
    cam.d[:] = [float(v) for v in d]
    cam.sizex, cam.sizey = int(sizex), int(sizey)
    return cam


def _c3(v):
    return np.ascontiguousarray(v, dtype=np.float64).reshape(3)


class OracleScene:
    """The oracle's scene: mesh objects (in order), spheres (in order), kd-tree."""

    def __init__(self, scene=None, library=None):
        self.L = library if library is not None else lib()
        self.h = C.c_void_p(self.L.orc_scene_new())
        self.built = False
        if scene is not None:
            for t in scene.get("textures", []):
                self.add_texture(t)
            for m in scene["meshes"]:
                o = self.add_object(m["verts"], m["lig"], m["col"], m["type"], m.get("tri_cols"), m.get("vnormals"))
                if m.get("tri_tex") is not None:
                    self.set_textures(o, m["tri_tex"], m["tri_ts"])
            for s in scene["spheres"]:
                self.add_sphere(s["rad"], s["pos"], s["lig"], s["col"], s["type"])
            self.build()

    def add_texture(self, bgr):
        """bgr: uint8 array [rows, cols, 3]"""
        a = np.ascontiguousarray(bgr, dtype=np.uint8)
        return self.L.orc_add_texture(self.h, a.ctypes.data_as(C.POINTER(C.c_ubyte)), a.shape[0], a.shape[1])

    def set_textures(self, obj, tri_tex, tri_ts):
        tt = np.ascontiguousarray(tri_tex, dtype=np.int32)
        ts = np.ascontiguousarray(tri_ts, dtype=np.float64).reshape(-1, 6)
        self.L.orc_set_textures(self.h, int(obj), _i(tt), _d(ts))

    def tri_getcol(self, gid, u, v):
        out = np.zeros(3)
        self.L.orc_tri_getcol(self.h, int(gid), float(u), float(v), _d(out))
        return out

    def __del__(self):
        try:
            self.L.orc_scene_free(self.h)
        except Exception:
            pass

    def add_object(self, verts, lig, col, type_, tri_cols=None, vnormals=None):
        v = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 9)
        tc = None if tri_cols is None else np.ascontiguousarray(tri_cols, dtype=np.float64).reshape(-1, 3)
        vn = None if vnormals is None else np.ascontiguousarray(vnormals, dtype=np.float64).reshape(-1, 9)
        return self.L.orc_add_object(self.h, _d(v), v.shape[0], _d(_c3(lig)), _d(_c3(col)),
                                     None if tc is None else _d(tc), None if vn is None else _d(vn), int(type_))

    def add_sphere(self, rad, pos, lig, col, type_):
        return self.L.orc_add_sphere(self.h, float(rad), _d(_c3(pos)), _d(_c3(lig)), _d(_c3(col)), int(type_))

    def build(self):
        self.L.orc_build(self.h)
        self.built = True

    @property
    def num_nodes(self):
        return self.L.orc_num_nodes(self.h)

    @property
    def tree_depth(self):
        return self.L.orc_tree_depth(self.h)

    @property
    def num_tris(self):
        return self.L.orc_num_tris(self.h)

    def tree_dump(self):
        nn, nt = self.num_nodes, self.num_tris
        out = dict(box=np.zeros((nn, 6)), split=np.zeros(nn), axis=np.zeros(nn, np.int32), left=np.zeros(nn, np.int32),
                   right=np.zeros(nn, np.int32), first=np.zeros(nn, np.int32), count=np.zeros(nn, np.int32),
                   list=np.zeros(max(nt, 1), np.int32))
        self.L.orc_tree_dump(self.h, _d(out["box"]), _d(out["split"]), _i(out["axis"]), _i(out["left"]), _i(out["right"]),
                             _i(out["first"]), _i(out["count"]), _i(out["list"]))
        out["list"] = out["list"][:nt]
        return out

    def trace(self, rays):
        r = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = r.shape[0]
        tid, tt = np.zeros(n, np.int32), np.zeros(n)
        sid, st = np.zeros(n, np.int32), np.zeros(n)
        stats = OStats()
        self.L.orc_trace(self.h, n, _d(r), _i(tid), _d(tt), _i(sid), _d(st), C.byref(stats))
        return tid, tt, sid, st, stats.as_dict()

    def tri_intersect(self, gid, ray):
        r = np.ascontiguousarray(ray, dtype=np.float64).reshape(6)
        u, v = C.c_double(), C.c_double()
        t = self.L.orc_tri_intersect(self.h, int(gid), _d(r), C.byref(u), C.byref(v))
        return t, u.value, v.value

    def tri_getn(self, gid, x):
        out = np.zeros(3)
        self.L.orc_tri_getn(self.h, int(gid), _d(_c3(x)), _d(out))
        return out

    def tri_nc(self, gid):
        out = np.zeros(3)
        self.L.orc_tri_nc(self.h, int(gid), _d(out))
        return out

    def sph_intersect(self, i, ray):
        r = np.ascontiguousarray(ray, dtype=np.float64).reshape(6)
        return self.L.orc_sph_intersect(self.h, int(i), _d(r))

    def render_tile(self, cam, params, x0, y0, rows, cols, nthreads=1):
        rgb = np.zeros((rows, cols, 3))
        pid = np.zeros((rows, cols), np.int32)
        st = OStats()
        self.L.orc_render_tile(self.h, C.byref(cam), C.byref(params), x0, y0, rows, cols, _d(rgb), _i(pid), C.byref(st),
                               int(nthreads))
        return rgb, pid, st.as_dict()

    def radiance(self, params, rays):
        r = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        rgb = np.zeros((r.shape[0], 3))
        st = OStats()
        self.L.orc_radiance(self.h, C.byref(params), r.shape[0], _d(r), _d(rgb), C.byref(st))
        return rgb, st.as_dict()


def check_aabb(box, ray):
    b = np.ascontiguousarray(box, dtype=np.float64).reshape(6)
    r = np.ascontiguousarray(ray, dtype=np.float64).reshape(6)
    return bool(lib().orc_check_aabb(_d(b), _d(r)))


def primary_ray(camo, d):
    out = np.zeros(6)
    lib().orc_primary_ray(_d(_c3(camo)), _d(_c3(d)), _d(out))
    return out


def camera_basis(cam):
    d, cx, cy = np.zeros(3), np.zeros(3), np.zeros(3)
    lib().orc_camera_basis(C.byref(cam), _d(d), _d(cx), _d(cy))
    return d, cx, cy


def draw_pixels(rgb):
    a = np.ascontiguousarray(rgb, dtype=np.float64).reshape(-1, 3)
    out = np.zeros((a.shape[0], 3), np.uint8)
    lib().orc_draw_pixels(_d(a), a.shape[0], out.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return out.reshape(np.shape(rgb))


# --------------------------------------------------------------------------- reference object code

_ref = None


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_l3.so"))


def ref_lib():
    global _ref
    if _ref is None:
        R = C.CDLL(os.path.join(_HERE, "_ref", "libref_l3.so"))
        R.ref_new.restype = C.c_void_p
        R.ref_free.argtypes = [C.c_void_p]
        R.ref_add_object.argtypes = [C.c_void_p, _dp, C.c_int, _dp, _dp, _dp, _dp, C.c_int]
        R.ref_add_sphere.argtypes = [C.c_void_p, C.c_double, _dp, _dp, _dp, C.c_int]
        R.ref_add_texture.argtypes = [C.c_void_p, C.POINTER(C.c_ubyte), C.c_int, C.c_int]
        R.ref_add_object_tex.argtypes = [C.c_void_p, _dp, C.c_int, _dp, _dp, _dp, _dp, C.c_int, _ip, _dp]
        R.ref_poly_getcol.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, _dp]
        R.ref_flatten_desc.argtypes = [C.c_void_p]
        R.ref_flatten_desc.restype = C.c_void_p
        R.ref_build.argtypes = [C.c_void_p]
        R.ref_check.argtypes = [C.c_void_p, _dp, _ip, _ip, _dp]
        R.ref_sphere_intersect.argtypes = [C.c_void_p, _dp, _ip, _dp]
        R.ref_sph_intersect.argtypes = [C.c_void_p, C.c_int, _dp]
        R.ref_sph_intersect.restype = C.c_double
        R.ref_poly_intersect.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp]
        R.ref_poly_intersect.restype = C.c_double
        R.ref_poly_getn.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp]
        R.ref_poly_nc.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp]
        R.ref_check_aabb.argtypes = [_dp, _dp]
        R.ref_num_nodes.argtypes = [C.c_void_p]
        R.ref_tree_dump.argtypes = [C.c_void_p, _dp, _dp, _ip, _ip, _ip, _ip, _ip, _ip, _ip]
        R.ref_primary_ray.argtypes = [_dp, _dp, _dp]
        R.ref_vec_norm.argtypes = [_dp, _dp]
        R.ref_readfile.argtypes = [C.c_void_p, C.c_char_p, C.c_double, _dp, _dp, _dp, _dp, C.c_int]
        R.ref_obj_num_polys.argtypes = [C.c_void_p, C.c_int]
        R.ref_obj_dump.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp, _ip, _ip]
        _ref = R
    return _ref


class RefScene:
    """The reference's own Object/Polygon/Sphere/KdTree instances behind a C shim."""

    def __init__(self, scene=None):
        self.R = ref_lib()
        self.h = C.c_void_p(self.R.ref_new())
        self.offsets = [0]
        if scene is not None:
            for t in scene.get("textures", []):
                a = np.ascontiguousarray(t, dtype=np.uint8)
                self.R.ref_add_texture(self.h, a.ctypes.data_as(C.POINTER(C.c_ubyte)), a.shape[0], a.shape[1])
            for m in scene["meshes"]:
                self.add_object(m["verts"], m["lig"], m["col"], m["type"], m.get("tri_cols"), m.get("vnormals"),
                                m.get("tri_tex"), m.get("tri_ts"))
            for s in scene["spheres"]:
                self.add_sphere(s["rad"], s["pos"], s["lig"], s["col"], s["type"])
            self.build()

    def add_object(self, verts, lig, col, type_, tri_cols=None, vnormals=None, tri_tex=None, tri_ts=None):
        v = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 9)
        tc = None if tri_cols is None else np.ascontiguousarray(tri_cols, dtype=np.float64).reshape(-1, 3)
        vn = None if vnormals is None else np.ascontiguousarray(vnormals, dtype=np.float64).reshape(-1, 9)
        self.offsets.append(self.offsets[-1] + v.shape[0])
        if tri_tex is not None:  # Polygons built by the reference's constructor with its tex / sizex / sizey / ts arguments
            tt = np.ascontiguousarray(tri_tex, dtype=np.int32)
            ts = np.ascontiguousarray(tri_ts, dtype=np.float64).reshape(-1, 6)
            return self.R.ref_add_object_tex(self.h, _d(v), v.shape[0], _d(_c3(lig)), _d(_c3(col)),
                                             None if tc is None else _d(tc), None if vn is None else _d(vn), int(type_), _i(tt), _d(ts))
        return self.R.ref_add_object(self.h, _d(v), v.shape[0], _d(_c3(lig)), _d(_c3(col)),
                                     None if tc is None else _d(tc), None if vn is None else _d(vn), int(type_))

    def readfile(self, path, times, loc, rot_deg, lig, col, type_):
        """Object::readfile (object.cpp:9-286) on one more Object, prepared as raytrace.cpp:110-127 does.  `mtllib` names are
        opened relative to the working directory, so the call runs inside the file's directory.  Returns what the reader
        made of the file -- per Polygon: verts[n,9], col[n,3], lig[n,3], ts[n,6], nc[n,3], type[n], has_tex[n] -- or None
        if readfile returned false."""
        path = os.path.abspath(str(path))
        cwd = os.getcwd()
        os.chdir(os.path.dirname(path))
        try:
            no = self.R.ref_readfile(self.h, path.encode(), float(times), _d(_c3(loc)), _d(_c3(rot_deg)), _d(_c3(lig)), _d(_c3(col)), int(type_))
        finally:
            os.chdir(cwd)
        if no < 0:
            return None
        n = self.R.ref_obj_num_polys(self.h, no)
        self.offsets.append(self.offsets[-1] + n)
        out = dict(verts=np.zeros((n, 9)), col=np.zeros((n, 3)), lig=np.zeros((n, 3)), ts=np.zeros((n, 6)), nc=np.zeros((n, 3)),
                   type=np.zeros(n, np.int32), has_tex=np.zeros(n, np.int32))
        if n:
            self.R.ref_obj_dump(self.h, no, _d(out["verts"]), _d(out["col"]), _d(out["lig"]), _d(out["ts"]), _d(out["nc"]), _i(out["type"]),
                                _i(out["has_tex"]))
        return out

    def flatten_desc(self):
        """Address of the rt_scene_desc that integration/rt_binding.h builds from the reference's objects."""
        return self.R.ref_flatten_desc(self.h)

    def poly_getcol(self, gid, u, v):
        no, nv = self.split_gid(gid)
        out = np.zeros(3)
        self.R.ref_poly_getcol(self.h, no, nv, float(u), float(v), _d(out))
        return out

    def add_sphere(self, rad, pos, lig, col, type_):
        return self.R.ref_add_sphere(self.h, float(rad), _d(_c3(pos)), _d(_c3(lig)), _d(_c3(col)), int(type_))

    def build(self):
        self.R.ref_build(self.h)

    def gid(self, no, nv):
        """Global triangle id = triangles of earlier objects + face index (cuda.cu:80,158-159)."""
        return self.offsets[no] + nv

    def split_gid(self, gid):
        no = int(np.searchsorted(np.asarray(self.offsets), gid, side="right") - 1)
        return no, gid - self.offsets[no]

    def check(self, ray):
        r = np.ascontiguousarray(ray, dtype=np.float64).reshape(6)
        no, nv, t = C.c_int(), C.c_int(), C.c_double()
        f = self.R.ref_check(self.h, _d(r), C.byref(no), C.byref(nv), C.byref(t))
        return bool(f), (self.gid(no.value, nv.value) if f else -1), t.value

    def sphere_intersect(self, ray):
        r = np.ascontiguousarray(ray, dtype=np.float64).reshape(6)
        i, t = C.c_int(), C.c_double()
        f = self.R.ref_sphere_intersect(self.h, _d(r), C.byref(i), C.byref(t))
        return bool(f), (i.value if f else -1), t.value

    def trace(self, rays):
        r = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = r.shape[0]
        tid, tt = np.zeros(n, np.int32), np.zeros(n)
        sid, st = np.zeros(n, np.int32), np.zeros(n)
        for k in range(n):
            _, tid[k], tt[k] = self.check(r[k])
            _, sid[k], st[k] = self.sphere_intersect(r[k])
        return tid, tt, sid, st

    def poly_intersect(self, gid, ray):
        no, nv = self.split_gid(gid)
        r = np.ascontiguousarray(ray, dtype=np.float64).reshape(6)
        u, v = C.c_double(), C.c_double()
        t = self.R.ref_poly_intersect(self.h, no, nv, _d(r), C.byref(u), C.byref(v))
        return t, u.value, v.value

    def poly_getn(self, gid, x):
        no, nv = self.split_gid(gid)
        out = np.zeros(3)
        self.R.ref_poly_getn(self.h, no, nv, _d(_c3(x)), _d(out))
        return out

    def poly_nc(self, gid):
        no, nv = self.split_gid(gid)
        out = np.zeros(3)
        self.R.ref_poly_nc(self.h, no, nv, _d(out))
        return out

    def sph_intersect(self, i, ray):
        r = np.ascontiguousarray(ray, dtype=np.float64).reshape(6)
        return self.R.ref_sph_intersect(self.h, int(i), _d(r))

    def tree_dump(self):
        nn = self.R.ref_num_nodes(self.h)
        nt = self.offsets[-1]
        out = dict(box=np.zeros((nn, 6)), split=np.zeros(nn), axis=np.zeros(nn, np.int32), left=np.zeros(nn, np.int32),
                   right=np.zeros(nn, np.int32), first=np.zeros(nn, np.int32), count=np.zeros(nn, np.int32))
        lno, lnv = np.zeros(max(nt, 1), np.int32), np.zeros(max(nt, 1), np.int32)
        self.R.ref_tree_dump(self.h, _d(out["box"]), _d(out["split"]), _i(out["axis"]), _i(out["left"]), _i(out["right"]),
                             _i(out["first"]), _i(out["count"]), _i(lno), _i(lnv))
        out["list"] = (np.asarray(self.offsets, np.int32)[lno[:nt]] + lnv[:nt]).astype(np.int32)
        return out


def ref_check_aabb(box, ray):
    b = np.ascontiguousarray(box, dtype=np.float64).reshape(6)
    r = np.ascontiguousarray(ray, dtype=np.float64).reshape(6)
    return bool(ref_lib().ref_check_aabb(_d(b), _d(r)))


def ref_primary_ray(camo, d):
    out = np.zeros(6)
    ref_lib().ref_primary_ray(_d(_c3(camo)), _d(_c3(d)), _d(out))
    return out
