"""The C-ABI libraries load and export every function include/*.h declares (no GPU needed)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_host_library_exports_every_declared_symbol(ra):
    names = declared("rt_host.h", "rth_")
    assert len(names) >= 12
    lib = ra.capi.host_lib()
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(ra.capi.HOST_SYMBOLS) == names


def test_device_library_exports_every_declared_symbol(ra):
    names = declared("rt_hip.h", "rt_")
    assert len(names) >= 10
    lib = ctypes.CDLL(os.path.join(ra.capi.LIB_DIR, "librt_hip.so"))
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(ra.capi.DEVICE_SYMBOLS) == names


def test_gather_library_exports_every_declared_symbol():
    """librt_rccl.so (it links librccl, so it is inspected, not loaded into this process)."""
    import subprocess
    names = declared("rt_gather.h", "rt_gather_")
    assert len(names) == 4
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "raytrace_amd", "lib", "librt_rccl.so")], text=True)
    have = {line.split()[-1] for line in out.splitlines() if line.strip()}
    for n in names:
        assert n in have, n


def test_struct_layouts_match_header(ra, tmp_path):
    """ctypes mirrors against what the C compiler makes of include/rt_hip.h and rt_host.h."""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "rt_host.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(rt_scene_desc),sizeof(rt_camera),sizeof(rt_tile),sizeof(rt_params),sizeof(rt_stats),"
                   "sizeof(rth_settings),offsetof(rt_scene_desc,sph_type));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    c = ra.capi
    assert sizes == [ctypes.sizeof(c.SceneDesc), ctypes.sizeof(c.Camera), ctypes.sizeof(c.Tile), ctypes.sizeof(c.Params),
                     ctypes.sizeof(c.Stats), ctypes.sizeof(c.Settings), c.SceneDesc.sph_type.offset]


def test_upload_rejects_inconsistent_descriptions_before_touching_the_gpu(ra):
    """rt_upload_scene validates the description first; these never reach a HIP call."""
    import numpy as np
    from raytrace_amd import scenes
    hs = ra.HostScene(scenes.tiny_mixed(0))
    D = ra.capi.device_lib()
    good = hs.desc.contents

    def try_upload(mutate):
        d = ra.capi.SceneDesc()
        ctypes.memmove(ctypes.byref(d), ctypes.byref(good), ctypes.sizeof(d))
        keep = mutate(d)
        h = ctypes.c_void_p()
        rc = D.rt_upload_scene(0, ctypes.byref(d), ctypes.byref(h))
        return rc, D.rt_last_error().decode(), keep

    rc, msg, _ = try_upload(lambda d: setattr(d, "n_list", d.n_tris - 1))
    assert rc == -1 and "n_list" in msg
    rc, msg, _ = try_upload(lambda d: setattr(d, "n_nodes", 0))
    assert rc == -1
    rc, msg, _ = try_upload(lambda d: setattr(d, "tri_verts", None))
    assert rc == -1 and "triangle arrays" in msg

    def bad_left(d):
        a = np.ctypeslib.as_array(good.node_left, shape=(good.n_nodes,)).copy()
        a[0] = 5
        d.node_left = a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        return a
    rc, msg, _ = try_upload(bad_left)
    assert rc == -1 and "pre-order" in msg

    def dup_list(d):
        a = np.ctypeslib.as_array(good.list_tri, shape=(good.n_list,)).copy()
        a[1] = a[0]
        d.list_tri = a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        return a
    rc, msg, _ = try_upload(dup_list)
    assert rc == -1 and "more than one" in msg
    # tree_depth sizes the traversal stacks: an understated value is rejected (a larger one is allowed)
    assert good.tree_depth > 1
    rc, msg, _ = try_upload(lambda d: setattr(d, "tree_depth", good.tree_depth - 1))
    assert rc == -1 and "tree_depth" in msg and str(good.tree_depth) in msg

    def shared_child(d):  # a node that is the right child of two nodes
        r = np.ctypeslib.as_array(good.node_right, shape=(good.n_nodes,)).copy()
        have = np.nonzero(r > 0)[0]
        assert len(have) >= 2
        r[have[0]] = r[have[-1]]
        d.node_right = r.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        return r
    rc, msg, _ = try_upload(shared_child)
    assert rc == -1 and ("more than one parent" in msg or "not reachable" in msg), msg

    def orphan(d):  # cut a subtree loose
        r = np.ctypeslib.as_array(good.node_right, shape=(good.n_nodes,)).copy()
        r[np.nonzero(r > 0)[0][0]] = -1
        d.node_right = r.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        return r
    rc, msg, _ = try_upload(orphan)
    assert rc == -1 and "not reachable" in msg, msg
    h = ctypes.c_void_p()
    assert D.rt_upload_scene(0, None, ctypes.byref(h)) == -1
