#!/usr/bin/env python3
"""Host-side build and upload times of a large scene: python3 tools/time_upload.py [n_triangles]"""
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
for threads in (os.environ.get("OMP_NUM_THREADS", "16"), "1"):
    os.environ["OMP_NUM_THREADS"] = threads
    t3 = time.time(); dev = ra.DeviceScene(hs, 0); t4 = time.time()
    cam = ra.capi.make_camera(sc["camera"]["o"], sc["camera"]["d"], 270, 480)
    rgb, pid, st = dev.render_tile(cam, ra.capi.make_params(samps=1, rr_depth=2, rr_mode=0), 0, 0, 270, 480)
    print(f"{n} triangles: generate {t1 - t0:.1f} s, host tree build {t2 - t1:.1f} s (depth {hs.tree_arrays()['depth'] if n < 5e6 else '-'}), "
          f"rt_upload_scene with {threads} thread(s) {t4 - t3:.1f} s; frame checksum {rgb.sum():.9f} rays {st['rays']}", flush=True)
    dev.close()
