#!/usr/bin/env python3
"""What one rank of N sees on the headline frame: every one of the N real shares of the frame (64x64 tiles, tile i to
rank i mod N) rendered on this GPU, ONE frame in flight, against the whole frame -- rate from the kernel's own
duration (HIP events around the launch) and from the wall clock of back-to-back frames, each waited for.
Usage: python3 tools/shard_probe.py [workload] [N] [frames]"""
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
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    scene, mode = bench.build_workload(wl, 0, 0)
    sizex, sizey = scene["sizex"], scene["sizey"]
    host = ra.HostScene(scene)
    cam = ra.capi.make_camera(scene["camera"]["o"], scene["camera"]["d"], sizex, sizey)
    dev = ra.DeviceScene(host, 0)
    device = torch.device("cuda", 0)
    tile = int(os.environ.get("SHARD_TILE", "64"))
    tiles = frame.plan_tiles(sizex, sizey, tile, tile)
    out = torch.zeros((sizex * sizey, 3), dtype=torch.float64, device=device)
    stream = torch.cuda.current_stream(device).cuda_stream

    def run(my_tiles):
        kms = []
        for i in range(4):
            dev.render_tiles_device(cam, ra.capi.make_params(samps=1, rr_depth=mode["rr_depth"], rr_mode=mode["rr_mode"], split_depth=2,
                                                             hard_depth=64, jitter=1, seed=1 + i), my_tiles, out.data_ptr(), None, stream)
        torch.cuda.synchronize()
        dev.stats()
        t0 = time.perf_counter()
        for i in range(frames):
            dev.render_tiles_device(cam, ra.capi.make_params(samps=1, rr_depth=mode["rr_depth"], rr_mode=mode["rr_mode"], split_depth=2,
                                                             hard_depth=64, jitter=1, seed=10 + i), my_tiles, out.data_ptr(), None, stream)
            torch.cuda.synchronize()
            kms.append(dev.last_kernel_ms())
        el = time.perf_counter() - t0
        st = dev.stats()
        k = sum(kms) / len(kms)
        return st["rays"] / frames, k, el / frames * 1e3, st["sched_lanes"] / max(1, 64 * st["sched_steps"])

    rays, k, w, occ = run(tiles)
    full = rays / k / 1e3
    print(f"{wl} {sizex}x{sizey} whole frame : {rays / 1e6:8.2f} Mrays  kernel {k:8.3f} ms  {rays / k / 1e3:8.1f} Mrays/s   wall {w:8.3f} ms  {rays / w / 1e3:8.1f} Mrays/s  lanes/step {occ:.3f}", flush=True)
    tot = 0.0
    for r in range(world):
        mine = frame.shard_tiles(tiles, r, world)
        if os.environ.get("SHARD_REVERSE"):
            mine = mine[::-1]
        rays, k, w, occ = run(mine)
        tot += rays / k / 1e3
        print(f"  share {r} of {world}: {rays / 1e6:8.2f} Mrays  kernel {k:8.3f} ms  {rays / k / 1e3:8.1f} Mrays/s ({rays / k / 1e3 / full:.3f} of whole)  "
              f"wall {w:8.3f} ms  {rays / w / 1e3:8.1f} Mrays/s  lanes/step {occ:.3f}", flush=True)
    print(f"  mean share rate / whole-frame rate (kernel time): {tot / world / full:.3f}")
    dev.close()


if __name__ == "__main__":
    main()
