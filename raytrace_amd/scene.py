"""Host-side mirror of the reference's scene model and of its offload boundary.

HostScene  = the reference's globals `objs`, `spheres`, `kdtree` (raytrace.cpp:24-26) held by
             librt_host.so: add meshes/spheres (or load a scene file), build the kd-tree, get the
             flattened description.
DeviceScene = what copyToDevice (cuda.cu:91-176) produced: the scene resident in one GPU's HBM,
             plus the render / query entry points of include/rt_hip.h.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import check_dev, check_host, c3, dptr, iptr


class HostScene:
    def __init__(self, scene=None):
        self.H = capi.host_lib()
        self.h = C.c_void_p(self.H.rth_scene_new())
        self.settings = None
        if scene is not None:
            for t in scene.get("textures", []):
                self.add_texture(t)
            for m in scene["meshes"]:
                i = self.add_mesh(m["verts"], m["lig"], m["col"], m["type"], m.get("tri_cols"))
                if m.get("tri_tex") is not None:
                    self.set_mesh_textures(i, m["tri_tex"], m["tri_ts"])
            for s in scene["spheres"]:
                self.add_sphere(s["rad"], s["pos"], s["lig"], s["col"], s["type"])
            self.build()

    @classmethod
    def from_file(cls, path, obj_flags=0):
        """Load the reference's scene-file format (raytrace.cpp:106-151) and build the tree."""
        self = cls()
        st = capi.Settings()
        check_host(self.H.rth_load_scene_file_ex(self.h, str(path).encode(), C.byref(st), int(obj_flags)), "rth_load_scene_file")
        self.settings = st
        self.build()
        return self

    def __del__(self):
        try:
            self.H.rth_scene_free(self.h)
        except Exception:
            pass

    def add_mesh(self, verts, lig, col, type_, tri_cols=None):
        v = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 9)
        tc = None if tri_cols is None else np.ascontiguousarray(tri_cols, dtype=np.float64).reshape(-1, 3)
        if tc is not None and tc.shape[0] != v.shape[0]:
            raise ValueError("tri_cols must have one colour per triangle")
        return check_host(self.H.rth_add_mesh(self.h, dptr(v), v.shape[0], dptr(c3(lig)), dptr(c3(col)),
                                              None if tc is None else dptr(tc), int(type_)), "rth_add_mesh")

    def add_sphere(self, rad, pos, lig, col, type_):
        return check_host(self.H.rth_add_sphere(self.h, float(rad), dptr(c3(pos)), dptr(c3(lig)), dptr(c3(col)), int(type_)),
                          "rth_add_sphere")

    def add_obj_file(self, path, scale, loc, rot_deg, lig, col, type_, flags=0):
        return check_host(self.H.rth_add_obj_file_ex(self.h, str(path).encode(), float(scale), dptr(c3(loc)), dptr(c3(rot_deg)),
                                                     dptr(c3(lig)), dptr(c3(col)), int(type_), int(flags)), "rth_add_obj_file")

    def add_texture(self, bgr):
        """One image, uint8 [rows, cols, 3] in BGR order (what loadImage keeps, opencvlib.cpp:21-29)."""
        a = np.ascontiguousarray(bgr, dtype=np.uint8)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("a texture is a [rows, cols, 3] uint8 array")
        return check_host(self.H.rth_add_texture(self.h, a.ctypes.data_as(C.POINTER(C.c_uint8)), a.shape[0], a.shape[1]), "rth_add_texture")

    def load_texture_file(self, path, times=1.0):
        return check_host(self.H.rth_load_texture_file(self.h, str(path).encode(), float(times)), "rth_load_texture_file")

    def set_mesh_textures(self, mesh, tri_tex, tri_ts):
        tt = np.ascontiguousarray(tri_tex, dtype=np.int32)
        ts = np.ascontiguousarray(tri_ts, dtype=np.float64).reshape(-1, 6)
        if ts.shape[0] != tt.shape[0]:
            raise ValueError("tri_ts must have six values per triangle")
        return check_host(self.H.rth_set_mesh_textures(self.h, int(mesh), iptr(tt), dptr(ts)), "rth_set_mesh_textures")

    def textures(self):
        """Copies of the images held by the scene, as [rows, cols, 3] uint8 arrays."""
        d = self.desc.contents
        out = []
        for k in range(d.n_textures):
            rows, cols = d.tex_sizey[k], d.tex_sizex[k]
            a = np.ctypeslib.as_array(d.tex_data, shape=(d.tex_offset[k] + 3 * rows * cols,))[d.tex_offset[k]:]
            out.append(a.reshape(rows, cols, 3).copy())
        return out

    def build(self):
        check_host(self.H.rth_build(self.h), "rth_build")

    @property
    def desc(self):
        p = self.H.rth_desc(self.h)
        if not p:
            raise capi.RtError("rth_desc: " + self.H.rth_last_error().decode())
        return p

    @property
    def num_tris(self):
        return int(self.H.rth_num_tris(self.h))

    @property
    def num_spheres(self):
        return int(self.H.rth_num_spheres(self.h))

    def tree_arrays(self):
        """Copy of the flattened tree and triangle arrays as numpy arrays."""
        d = self.desc.contents
        nn, nt = d.n_nodes, d.n_tris

        def arr(ptr, n, dt):
            return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt).copy() if n else np.zeros(0, dt)

        return dict(
            box=arr(d.node_box, nn * 6, np.float64).reshape(nn, 6), split=arr(d.node_split, nn, np.float64),
            axis=arr(d.node_axis, nn, np.int32), left=arr(d.node_left, nn, np.int32), right=arr(d.node_right, nn, np.int32),
            first=arr(d.node_first, nn, np.int32), count=arr(d.node_count, nn, np.int32), list=arr(d.list_tri, d.n_list, np.int32),
            depth=int(d.tree_depth), nc=arr(d.tri_nc, nt * 3, np.float64).reshape(nt, 3),
            verts=arr(d.tri_verts, nt * 9, np.float64).reshape(nt, 9), col=arr(d.tri_col, nt * 3, np.float64).reshape(nt, 3),
            tri_tex=arr(d.tri_tex, nt, np.int32), tri_ts=arr(d.tri_ts, nt * 6, np.float64).reshape(nt, 6))


def tonemap(rgb):
    """drawPixel (raytrace.cpp:33-39): returns uint8 BGR with the shape of `rgb`."""
    a = np.ascontiguousarray(rgb, dtype=np.float64).reshape(-1, 3)
    out = np.zeros((a.shape[0], 3), np.uint8)
    capi.host_lib().rth_tonemap(dptr(a), a.shape[0], out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.reshape(np.shape(rgb))


def write_bmp(path, bgr):
    a = np.ascontiguousarray(bgr, dtype=np.uint8)
    rows, cols = a.shape[0], a.shape[1]
    check_host(capi.host_lib().rth_write_bmp(str(path).encode(), a.ctypes.data_as(C.POINTER(C.c_uint8)), rows, cols),
               "rth_write_bmp")


def device_count():
    return check_dev(capi.device_lib().rt_device_count(), "rt_device_count")


def device_info(dev=0):
    buf = C.create_string_buffer(512)
    check_dev(capi.device_lib().rt_device_info(dev, buf, 512), "rt_device_info")
    return buf.value.decode()


class PreparedScene:
    """A scene laid out for the device in host memory (rt_prepare_scene): built once, uploaded to every GPU of the
    process.  No GPU is touched."""

    def __init__(self, host_scene):
        self.D = capi.device_lib()
        self.h = C.c_void_p()
        check_dev(self.D.rt_prepare_scene(host_scene.desc, C.byref(self.h)), "rt_prepare_scene")

    def close(self):
        if self.h:
            self.D.rt_free_prepared(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceScene:
    """A scene resident in one GPU's HBM (rt_upload_scene, or rt_upload_prepared for a PreparedScene)."""

    def __init__(self, host_scene, dev=0):
        self.D = capi.device_lib()
        self.h = C.c_void_p()
        self.dev = dev
        if isinstance(host_scene, PreparedScene):
            check_dev(self.D.rt_upload_prepared(int(dev), host_scene.h, C.byref(self.h)), "rt_upload_prepared")
        else:
            check_dev(self.D.rt_upload_scene(int(dev), host_scene.desc, C.byref(self.h)), "rt_upload_scene")

    def close(self):
        if self.h:
            self.D.rt_free_scene(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render_tile(self, cam, params, x0, y0, rows, cols):
        """rt_render_tile: host buffers, synchronous. Returns (rgb[rows,cols,3], prim_id[rows,cols], stats)."""
        rgb = np.zeros((rows, cols, 3))
        pid = np.zeros((rows, cols), np.int32)
        st = capi.Stats()
        tile = capi.Tile(x0, y0, rows, cols)
        check_dev(self.D.rt_render_tile(self.h, C.byref(cam), C.byref(tile), C.byref(params), dptr(rgb), iptr(pid), C.byref(st)),
                  "rt_render_tile")
        return rgb, pid, st.as_dict()

    def render_tiles_device(self, cam, params, tiles, d_rgb_ptr, d_prim_ptr=None, stream=None):
        """rt_render_tiles_device: `tiles` is a list of (x0, y0, rows, cols); outputs are raw device
        pointers (ints), e.g. torch tensors' data_ptr(); asynchronous on `stream` (int handle or None)."""
        arr = (capi.Tile * len(tiles))(*[capi.Tile(*t) for t in tiles])
        check_dev(self.D.rt_render_tiles_device(self.h, C.byref(cam), arr, len(tiles), C.byref(params), C.c_void_p(d_rgb_ptr),
                                                C.c_void_p(d_prim_ptr) if d_prim_ptr else None,
                                                C.c_void_p(stream) if stream else None), "rt_render_tiles_device")

    def render_tiles_noalloc(self, cam, params, tiles, d_rgb_ptr, d_prim_ptr=None, stream=None):
        """rt_render_tiles_noalloc: as render_tiles_device, but the call allocates nothing (bounded sizes)."""
        arr = (capi.Tile * len(tiles))(*[capi.Tile(*t) for t in tiles])
        check_dev(self.D.rt_render_tiles_noalloc(self.h, C.byref(cam), arr, len(tiles), C.byref(params), C.c_void_p(d_rgb_ptr),
                                                 C.c_void_p(d_prim_ptr) if d_prim_ptr else None,
                                                 C.c_void_p(stream) if stream else None), "rt_render_tiles_noalloc")

    def stats(self):
        st = capi.Stats()
        check_dev(self.D.rt_get_stats(self.h, C.byref(st)), "rt_get_stats")
        return st.as_dict()

    def last_kernel_ms(self):
        ms = C.c_float()
        check_dev(self.D.rt_last_kernel_ms(self.h, C.byref(ms)), "rt_last_kernel_ms")
        return float(ms.value)

    def trace_rays(self, rays):
        """rt_trace_rays: returns (tri_id, tri_t, sph_id, sph_t, stats) for rays[n,6]."""
        r = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = r.shape[0]
        tid, tt = np.zeros(n, np.int32), np.zeros(n)
        sid, st_ = np.zeros(n, np.int32), np.zeros(n)
        st = capi.Stats()
        check_dev(self.D.rt_trace_rays(self.h, n, dptr(r), iptr(tid), dptr(tt), iptr(sid), dptr(st_), C.byref(st)), "rt_trace_rays")
        return tid, tt, sid, st_, st.as_dict()


def debug_fp64(op, a, b=None, dev=0):
    a = np.ascontiguousarray(a, dtype=np.float64).ravel()
    b = a if b is None else np.ascontiguousarray(b, dtype=np.float64).ravel()
    out = np.zeros_like(a)
    check_dev(capi.device_lib().rt_debug_fp64(dev, {"div": 0, "sqrt": 1}[op], a.size, dptr(a), dptr(b), dptr(out)), "rt_debug_fp64")
    return out
