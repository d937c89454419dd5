"""Run by tests/test_gpu_rccl.py in a child process: a ONE-rank `nccl` (= RCCL) process group, rendezvous through a
file, and a frame assembled through torch.distributed.gather on device tensors -- the exchange an N-GPU run does --
compared with the plain single-launch render.  Prints OK on success."""
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
    dist.barrier()
    dist.destroy_process_group()
    dev.close()
    print("OK")


if __name__ == "__main__":
    main()
