"""Run by tests/test_gpu_rccl.py in a child process: a ONE-rank `nccl` (= RCCL) process group, rendezvous through a
file, and a frame assembled through torch.distributed.gather on device tensors -- the exchange an N-GPU run does --
compared with the plain single-launch render; then the same with two frames in flight on two streams.  Prints OK on
success."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import raytrace_amd as ra
    from raytrace_amd import frame, scenes
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    store = os.path.join(tempfile.mkdtemp(prefix="rt_pg_"), "store")
    dist.init_process_group("nccl", init_method="file://" + store, rank=0, world_size=1, device_id=device)
    assert dist.get_backend() == "nccl"
    sc = scenes.tiny_mixed(0)
    hs = ra.HostScene(sc)
    dev = ra.DeviceScene(hs, 0)
    sizex, sizey = 70, 90
    cam = ra.capi.make_camera(sc["camera"]["o"], sc["camera"]["d"], sizex, sizey)
    prm = ra.capi.make_params(samps=1, rr_depth=3, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=2)
    fr = frame.FrameRenderer(dev, cam, prm, 16, 24, want_ids=True, collective=True)  # one rank gathers too
    for _ in range(2):
        out = fr.render()
        torch.cuda.synchronize()
    full, fpid, _ = dev.render_tile(cam, prm, 0, 0, sizex, sizey)
    assert np.array_equal(out.cpu().numpy(), full)
    assert np.array_equal(fr.ids.cpu().numpy(), fpid)
    # two frames in flight, each with its own stream and gather, over the same scene: frame i is the frame of seed i
    pipe = frame.FramePipeline(dev, cam, prm, 2, tile_rows=16, tile_cols=24, want_ids=True, collective=True)
    slots = [pipe.submit(seed) for seed in (7, 8, 9, 10)]
    assert slots == [0, 1, 0, 1]
    for slot, seed in ((0, 9), (1, 10)):
        prm.seed = seed
        want, wid, _ = dev.render_tile(cam, prm, 0, 0, sizex, sizey)
        assert np.array_equal(pipe.frame(slot).cpu().numpy(), want)
        assert np.array_equal(pipe.renderers[slot].ids.cpu().numpy(), wid)
    dist.barrier()
    dist.destroy_process_group()
    dev.close()
    print("OK")


if __name__ == "__main__":
    main()
