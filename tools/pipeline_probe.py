#!/usr/bin/env python3
"""How much do frames gain from overlapping one frame's last paths with the next frame's first?  frame.FramePipeline
with one, two and three frames in flight over one upload of the scene (the library overlaps launches on two streams;
a third waits).  Usage: python3 tools/pipeline_probe.py [workload] [sizex sizey] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raytrace_amd as ra  # noqa: E402
from raytrace_amd import frame  # noqa: E402
import bench  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "sponza"
    sx = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    sy = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    scene, mode = bench.build_workload(wl, sx, sy)
    host = ra.HostScene(scene)
    cam = ra.capi.make_camera(scene["camera"]["o"], scene["camera"]["d"], scene["sizex"], scene["sizey"])
    device = torch.device("cuda", 0)
    dev = ra.DeviceScene(host, 0)
    prm = ra.capi.make_params(samps=1, rr_depth=mode["rr_depth"], rr_mode=mode["rr_mode"], split_depth=2, hard_depth=64, jitter=1, seed=1)
    pipes = {n: frame.FramePipeline(dev, cam, prm, n, tile_rows=64, tile_cols=64, device=device) for n in (1, 2, 3)}

    def run(n):
        pipe = pipes[n]
        for i in range(4):
            pipe.submit(1 + i)
        torch.cuda.synchronize()
        dev.stats()
        t0 = time.perf_counter()
        for i in range(steps):
            pipe.submit(10 + i)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        return dev.stats()["rays"] / el / 1e6, el / steps * 1e3

    for n in (1, 2, 3, 1, 2, 3):
        r, ms = run(n)
        print(f"{wl} {scene['sizex']}x{scene['sizey']} frames in flight {n}: {r:8.1f} Mrays/s  {ms:7.3f} ms/frame", flush=True)
    dev.close()


if __name__ == "__main__":
    main()
