#!/usr/bin/env python3
"""Host-side build and upload times of a large scene: python3 tools/time_upload.py [n_triangles]
rt_upload_scene = rt_prepare_scene (host only: device layout, culling hierarchies) + rt_upload_prepared (copies);
a host with several GPUs prepares once and uploads once per GPU."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytrace_amd as ra  # noqa: E402
from raytrace_amd import scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 28055742
t0 = time.time(); sc = scenes.lucy_class(n, 270, 480); t1 = time.time()
hs = ra.HostScene(sc); t2 = time.time()
cam = ra.capi.make_camera(sc["camera"]["o"], sc["camera"]["d"], 270, 480)
prm = ra.capi.make_params(samps=1, rr_depth=2, rr_mode=0)
print(f"{n} triangles: generate {t1 - t0:.1f} s, host tree build {t2 - t1:.1f} s", flush=True)
for threads in (os.environ.get("OMP_NUM_THREADS", "16"), "1"):
    os.environ["OMP_NUM_THREADS"] = threads
    t3 = time.time(); dev = ra.DeviceScene(hs, 0); t4 = time.time()
    rgb, pid, st = dev.render_tile(cam, prm, 0, 0, 270, 480)
    print(f"rt_upload_scene with {threads} thread(s) {t4 - t3:.1f} s; frame checksum {rgb.sum():.9f} rays {st['rays']}", flush=True)
    dev.close()
t5 = time.time(); prep = ra.PreparedScene(hs); t6 = time.time()
print(f"rt_prepare_scene {t6 - t5:.1f} s", flush=True)
for k in range(2):
    t7 = time.time(); dev = ra.DeviceScene(prep, 0); t8 = time.time()
    rgb, pid, st = dev.render_tile(cam, prm, 0, 0, 270, 480)
    print(f"rt_upload_prepared #{k + 1} {t8 - t7:.2f} s; frame checksum {rgb.sum():.9f} rays {st['rays']}", flush=True)
    dev.close()
prep.close()
