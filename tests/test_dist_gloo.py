"""The multi-rank frame path on CPU: world_size-2 (and 3) gloo process groups, one process per
rank.  Each rank renders its interleaved tile share into a packed buffer, one gather brings the
shares to rank 0, which un-permutes them; the result must equal the single-process frame bit for
bit.  The renderer behind the boundary is replaced by the CPU oracle here (no GPU), everything
else -- tile plan, sharding, padding, gather, un-permute -- is the product's frame.py."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZEX, SIZEY = 37, 52  # ragged against the 16x16 tiles


class OracleBackedScene:
    """Stands where a DeviceScene would: same render_tiles_device signature, host pointers."""
    dev = 0

    def __init__(self, orc, po):
        self.orc, self.po = orc, po

    def render_tiles_device(self, cam, params, tiles, rgb_ptr, ids_ptr=None, stream=None):
        n = sum(t[2] * t[3] for t in tiles)
        rgb = np.ctypeslib.as_array((ctypes.c_double * (3 * n)).from_address(rgb_ptr)).reshape(n, 3)
        ids = np.ctypeslib.as_array((ctypes.c_int32 * n).from_address(ids_ptr)) if ids_ptr else None
        ocam = self.po.make_camera(list(cam.o), list(cam.d), cam.sizex, cam.sizey)
        oprm = self.po.make_params(params.samps, params.rr_depth, params.rr_mode, params.split_depth, params.hard_depth,
                                   params.jitter, params.seed)
        off = 0
        for (x0, y0, r, c) in tiles:
            a, b, _ = self.orc.render_tile(ocam, oprm, x0, y0, r, c)
            rgb[off:off + r * c] = a.reshape(-1, 3)
            if ids is not None:
                ids[off:off + r * c] = b.reshape(-1)
            off += r * c


def _worker(rank, world, port, out_path):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import pyoracle as po
    import raytrace_amd as ra
    from raytrace_amd import frame, scenes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenes.tiny_mixed(3, n_tris=60, n_sph=4)
    cam = ra.capi.make_camera(sc["camera"]["o"], sc["camera"]["d"], SIZEX, SIZEY)
    prm = ra.capi.make_params(samps=1, rr_depth=3, rr_mode=1, split_depth=2, hard_depth=16, jitter=1, seed=5)
    fr = frame.FrameRenderer(OracleBackedScene(po.OracleScene(sc), po), cam, prm, 16, 16, rank=rank, world=world,
                             device=torch.device("cpu"), want_ids=True)
    out = fr.render()
    # the same through the pipeline of frames in flight (on the CPU its slots share the caller's thread; what is
    # covered here is the slots' bookkeeping and one gather per slot, in the same order on every rank)
    pipe = frame.FramePipeline(OracleBackedScene(po.OracleScene(sc), po), cam, prm, 2, tile_rows=16, tile_cols=16, rank=rank,
                               world=world, device=torch.device("cpu"), want_ids=True)
    slots = [pipe.submit(seed) for seed in (5, 6, 7)]
    assert slots == [0, 1, 0] and prm.seed == 5
    if rank == 0:
        np.savez(out_path, rgb=out.numpy(), ids=fr.ids.numpy(), rgb6=pipe.frame(1).numpy(), rgb7=pipe.frame(0).numpy(),
                 ids7=pipe.renderers[0].ids.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_gather_assembles_the_same_frame(po, ra, tmp_path, world):
    import torch.multiprocessing as mp
    from raytrace_amd import scenes
    out_path = str(tmp_path / f"frame{world}.npz")
    mp.spawn(_worker, args=(world, _free_port(), out_path), nprocs=world, join=True)
    got = np.load(out_path)
    sc = scenes.tiny_mixed(3, n_tris=60, n_sph=4)
    orc = po.OracleScene(sc)
    rgb, ids, _ = orc.render_tile(po.make_camera(sc["camera"]["o"], sc["camera"]["d"], SIZEX, SIZEY),
                                  po.make_params(samps=1, rr_depth=3, rr_mode=1, split_depth=2, hard_depth=16, jitter=1, seed=5),
                                  0, 0, SIZEX, SIZEY)
    assert np.array_equal(got["rgb"], rgb)
    assert np.array_equal(got["ids"], ids)
    for seed in (6, 7):
        want, wids, _ = orc.render_tile(po.make_camera(sc["camera"]["o"], sc["camera"]["d"], SIZEX, SIZEY),
                                        po.make_params(samps=1, rr_depth=3, rr_mode=1, split_depth=2, hard_depth=16, jitter=1, seed=seed),
                                        0, 0, SIZEX, SIZEY)
        assert np.array_equal(got[f"rgb{seed}"], want)
    assert np.array_equal(got["ids7"], wids)
    assert (ids != po.MISS_ID).any()


def test_tile_plan_and_shards_cover_the_image_once():
    from raytrace_amd import frame
    for sizex, sizey, tr, tc in ((2160, 3840, 64, 64), (37, 52, 16, 16), (8, 8, 64, 64), (65, 1, 64, 64)):
        tiles = frame.plan_tiles(sizex, sizey, tr, tc)
        cover = np.zeros((sizex, sizey), np.int32)
        for (x0, y0, r, c) in tiles:
            cover[x0:x0 + r, y0:y0 + c] += 1
        assert (cover == 1).all()
        for world in (1, 2, 4, 8):
            shards = [frame.shard_tiles(tiles, r, world) for r in range(world)]
            assert sorted(sum(shards, [])) == sorted(tiles)
            idx = np.concatenate([frame.dest_indices(s, sizey) for s in shards])
            assert np.array_equal(np.sort(idx), np.arange(sizex * sizey))
            counts = [frame.tile_pixels(s) for s in shards]
            if len(tiles) >= 8 * world:
                assert max(counts) <= 1.3 * (sizex * sizey / world)  # interleaving keeps the shares balanced
