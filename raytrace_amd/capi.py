"""ctypes bindings of the two C ABIs: include/rt_host.h (librt_host.so, CPU only) and
include/rt_hip.h (librt_hip.so, the HIP kernels).  There is no CPU fallback for the device
library: if it cannot be loaded, or a call fails, an exception is raised."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

RT_DIFF, RT_SPEC, RT_REFR = 0, 1, 2
RTH_OBJ_TRIANGULATE, RTH_OBJ_NEGATIVE = 1, 2
RT_ID_MISS = -2147483648

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class RtError(RuntimeError):
    pass


class SceneDesc(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32), ("node_box", _dp), ("node_split", _dp), ("node_axis", _ip), ("node_left", _ip),
        ("node_right", _ip), ("node_first", _ip), ("node_count", _ip), ("n_list", C.c_int32), ("list_tri", _ip),
        ("tree_depth", C.c_int32),
        ("n_tris", C.c_int32), ("tri_verts", _dp), ("tri_nc", _dp), ("tri_col", _dp), ("tri_lig", _dp), ("tri_type", _ip),
        ("n_spheres", C.c_int32), ("sph_rad", _dp), ("sph_pos", _dp), ("sph_lig", _dp), ("sph_col", _dp), ("sph_type", _ip),
        ("n_textures", C.c_int32), ("tex_data", C.POINTER(C.c_uint8)), ("tex_offset", C.POINTER(C.c_int64)), ("tex_sizex", _ip),
        ("tex_sizey", _ip), ("tri_tex", _ip), ("tri_ts", _dp),
    ]


class Camera(C.Structure):
    _fields_ = [("o", C.c_double * 3), ("d", C.c_double * 3), ("sizex", C.c_int32), ("sizey", C.c_int32)]


class Tile(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("samps", C.c_int32), ("rr_depth", C.c_int32), ("rr_mode", C.c_int32), ("split_depth", C.c_int32),
                ("hard_depth", C.c_int32), ("jitter", C.c_int32), ("seed", C.c_uint64), ("light_mode", C.c_int32),
                ("reserved", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64), ("sph_tests", C.c_uint64),
                ("tri_shades", C.c_uint64), ("tri_tests_done", C.c_uint64), ("cull_tests", C.c_uint64),
                ("sched_steps", C.c_uint64), ("sched_lanes", C.c_uint64), ("node_fetches", C.c_uint64),
                ("sph_tests_done", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Settings(C.Structure):
    _fields_ = [("sizex", C.c_int32), ("sizey", C.c_int32), ("c_samps", C.c_int32), ("g_samps", C.c_int32),
                ("num_per_node", C.c_int32), ("cu_stack_size", C.c_int32), ("view", C.c_double * 3), ("dir", C.c_double * 3)]


HOST_SYMBOLS = ["rth_scene_new", "rth_scene_free", "rth_add_mesh", "rth_add_sphere", "rth_add_obj_file", "rth_add_obj_file_ex",
                "rth_add_texture", "rth_load_texture_file", "rth_set_mesh_textures",
                "rth_load_scene_file", "rth_load_scene_file_ex", "rth_build", "rth_desc", "rth_num_tris", "rth_num_spheres", "rth_tonemap",
                "rth_write_bmp", "rth_last_error"]
DEVICE_SYMBOLS = ["rt_device_count", "rt_device_info", "rt_upload_scene", "rt_free_scene", "rt_prepare_scene", "rt_upload_prepared",
                  "rt_free_prepared", "rt_render_tile",
                  "rt_render_tiles_device", "rt_render_tiles_noalloc", "rt_get_stats", "rt_last_kernel_ms", "rt_trace_rays", "rt_debug_fp64",
                  "rt_last_error"]

_host = None
_dev = None


def host_lib():
    global _host
    if _host is None:
        path = os.path.join(LIB_DIR, "librt_host.so")
        if not os.path.exists(path):
            raise RtError(f"{path} is missing: run `make -C raytrace_amd/csrc` (or __graft_entry__.build())")
        H = C.CDLL(path)
        H.rth_scene_new.restype = C.c_void_p
        H.rth_scene_free.argtypes = [C.c_void_p]
        H.rth_add_mesh.argtypes = [C.c_void_p, _dp, C.c_int64, _dp, _dp, _dp, C.c_int]
        H.rth_add_sphere.argtypes = [C.c_void_p, C.c_double, _dp, _dp, _dp, C.c_int]
        H.rth_add_obj_file.argtypes = [C.c_void_p, C.c_char_p, C.c_double, _dp, _dp, _dp, _dp, C.c_int]
        H.rth_add_obj_file_ex.argtypes = [C.c_void_p, C.c_char_p, C.c_double, _dp, _dp, _dp, _dp, C.c_int, C.c_uint]
        H.rth_add_texture.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int32, C.c_int32]
        H.rth_load_texture_file.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        H.rth_set_mesh_textures.argtypes = [C.c_void_p, C.c_int, _ip, _dp]
        H.rth_load_scene_file.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(Settings)]
        H.rth_load_scene_file_ex.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(Settings), C.c_uint]
        H.rth_build.argtypes = [C.c_void_p]
        H.rth_desc.argtypes = [C.c_void_p]
        H.rth_desc.restype = C.POINTER(SceneDesc)
        H.rth_num_tris.argtypes = [C.c_void_p]
        H.rth_num_tris.restype = C.c_int64
        H.rth_num_spheres.argtypes = [C.c_void_p]
        H.rth_num_spheres.restype = C.c_int32
        H.rth_tonemap.argtypes = [_dp, C.c_size_t, C.POINTER(C.c_uint8)]
        H.rth_write_bmp.argtypes = [C.c_char_p, C.POINTER(C.c_uint8), C.c_int32, C.c_int32]
        H.rth_last_error.restype = C.c_char_p
        _host = H
    return _host


def device_lib():
    """The HIP library. Raises if it is missing -- there is no other implementation to fall back to."""
    global _dev
    if _dev is None:
        path = os.environ.get("RT_HIP_LIB") or os.path.join(LIB_DIR, "librt_hip.so")  # RT_HIP_LIB: A/B builds
        if not os.path.exists(path):
            raise RtError(f"{path} is missing: run `make -C raytrace_amd/csrc` (or __graft_entry__.build())")
        # One HIP runtime per process: torch bundles its own libamdhip64 / libhsa-runtime64, and a
        # second copy loaded next to it cannot open the GPU.  Importing torch first makes this
        # library's libamdhip64.so.7 dependency resolve to the copy torch already loaded, so device
        # buffers and streams can be shared with torch (frame gather over RCCL).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        D = C.CDLL(path)
        D.rt_device_count.restype = C.c_int
        D.rt_device_info.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
        D.rt_upload_scene.argtypes = [C.c_int, C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
        D.rt_free_scene.argtypes = [C.c_void_p]
        D.rt_prepare_scene.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
        D.rt_upload_prepared.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        D.rt_free_prepared.argtypes = [C.c_void_p]
        D.rt_render_tile.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Tile), C.POINTER(Params), _dp, _ip,
                                     C.POINTER(Stats)]
        D.rt_render_tiles_device.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Tile), C.c_int32, C.POINTER(Params),
                                             C.c_void_p, C.c_void_p, C.c_void_p]
        D.rt_render_tiles_noalloc.argtypes = D.rt_render_tiles_device.argtypes
        D.rt_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        D.rt_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        D.rt_trace_rays.argtypes = [C.c_void_p, C.c_int64, _dp, _ip, _dp, _ip, _dp, C.POINTER(Stats)]
        D.rt_debug_fp64.argtypes = [C.c_int, C.c_int, C.c_int64, _dp, _dp, _dp]
        D.rt_last_error.restype = C.c_char_p
        _dev = D
    return _dev


def check_host(rc, what):
    if rc < 0:
        raise RtError(f"{what}: {host_lib().rth_last_error().decode()} (code {rc})")
    return rc


def check_dev(rc, what):
    if rc < 0:
        raise RtError(f"{what}: {device_lib().rt_last_error().decode()} (code {rc})")
    return rc


def dptr(a):
    return a.ctypes.data_as(_dp)


def iptr(a):
    return a.ctypes.data_as(_ip)


def c3(v):
    return np.ascontiguousarray(v, dtype=np.float64).reshape(3)


def make_camera(o, d, sizex, sizey):
    cam = Camera()
    cam.o[:] = [float(v) for v in o]
    cam.d[:] = [float(v) for v in d]
    cam.sizex, cam.sizey = int(sizex), int(sizey)
    return cam


def make_params(samps=1, rr_depth=5, rr_mode=1, split_depth=2, hard_depth=64, jitter=1, seed=1, light_mode=0):
    return Params(int(samps), int(rr_depth), int(rr_mode), int(split_depth), int(hard_depth), int(jitter), int(seed), int(light_mode), 0)
