"""Launches of one scene on several streams overlap (the library keeps two sets of per-launch state per scene; a third
stream waits for the set used longest ago): frames queued back to back without a host synchronisation in between are
bit-identical to the same frames rendered one at a time, their work counters add up, and explicit rays and the entry
point that allocates nothing still work afterwards."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(ra, sc, jobs, kw):
    from raytrace_amd import frame
    out = []
    for (sizex, sizey, seed, tr, tc) in jobs:
        cam = ra.capi.make_camera(sc["camera"]["o"], sc["camera"]["d"], sizex, sizey)
        out.append((cam, ra.capi.make_params(seed=seed, **kw), frame.plan_tiles(sizex, sizey, tr, tc)))
    return out


@pytest.mark.parametrize("n_streams", [2, 3])
def test_overlapping_launches_equal_serial_ones(ra, n_streams):
    import torch
    from raytrace_amd import frame, scenes
    sc = scenes.cornell_demo()
    dev = ra.DeviceScene(ra.HostScene(sc), 0)
    kw = dict(samps=2, rr_depth=5, rr_mode=1, split_depth=2, hard_depth=64, jitter=1)
    # frames of different sizes and tilings, so that a set's tables and share buffers are re-sized and re-uploaded
    jobs = _frames(ra, sc, [(160, 200, 3, 32, 48), (90, 70, 4, 16, 16), (200, 260, 5, 64, 64), (64, 64, 6, 8, 8), (120, 90, 7, 40, 24), (33, 47, 8, 16, 16)], kw)
    keys = ("rays", "node_visits", "tri_tests", "tri_shades", "sph_tests")
    want, want_st = [], {k: 0 for k in keys}
    dev.stats()
    for cam, prm, tiles in jobs:
        n = frame.tile_pixels(tiles)
        buf = torch.zeros((n, 3), dtype=torch.float64, device="cuda:0")
        ids = torch.zeros((n,), dtype=torch.int32, device="cuda:0")
        dev.render_tiles_device(cam, prm, tiles, buf.data_ptr(), ids.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        st = dev.stats()
        for k in keys:
            want_st[k] += st[k]
        want.append((buf.cpu().numpy(), ids.cpu().numpy()))
    streams = [torch.cuda.Stream("cuda:0") for _ in range(n_streams)]
    bufs = []
    for cam, prm, tiles in jobs:
        n = frame.tile_pixels(tiles)
        bufs.append((torch.full((n, 3), -1.0, dtype=torch.float64, device="cuda:0"), torch.full((n,), -7, dtype=torch.int32, device="cuda:0")))
    torch.cuda.synchronize()
    for i, (cam, prm, tiles) in enumerate(jobs):  # no host synchronisation between the launches
        st = streams[i % n_streams]
        dev.render_tiles_device(cam, prm, tiles, bufs[i][0].data_ptr(), bufs[i][1].data_ptr(), st.cuda_stream)
    got_st = dev.stats()  # waits for every launch in flight
    for i in range(len(jobs)):
        assert np.array_equal(bufs[i][0].cpu().numpy(), want[i][0]), i
        assert np.array_equal(bufs[i][1].cpu().numpy(), want[i][1]), i
    assert {k: got_st[k] for k in keys} == want_st
    # the other entry points share the first set of per-launch state: they queue behind whatever used it last
    cam, prm, tiles = jobs[1]
    n = frame.tile_pixels(tiles)
    a = torch.zeros((n, 3), dtype=torch.float64, device="cuda:0")
    dev.render_tiles_device(jobs[0][0], jobs[0][1], jobs[0][2], bufs[0][0].data_ptr(), None, streams[0].cuda_stream)
    dev.render_tiles_noalloc(cam, prm, tiles, a.data_ptr(), None, streams[-1].cuda_stream)
    full, _, _ = dev.render_tile(cam, prm, 0, 0, cam.sizex, cam.sizey)
    torch.cuda.synchronize()
    assert np.array_equal(bufs[0][0].cpu().numpy(), want[0][0])
    got = np.zeros((cam.sizex, cam.sizey, 3))
    frame.scatter_to_frame(got, tiles, a.cpu().numpy())
    assert np.array_equal(got, full)
    dev.close()


def test_frame_pipeline_equals_frame_renderer(ra):
    import torch
    from raytrace_amd import frame, scenes
    sc = scenes.tiny_mixed(1)
    dev = ra.DeviceScene(ra.HostScene(sc), 0)
    sizex, sizey = 150, 130
    cam = ra.capi.make_camera(sc["camera"]["o"], sc["camera"]["d"], sizex, sizey)
    prm = ra.capi.make_params(samps=1, rr_depth=4, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=1)
    fr = frame.FrameRenderer(dev, cam, ra.capi.make_params(samps=1, rr_depth=4, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=1), 32, 32)
    want = []
    for seed in range(20, 27):
        fr.params.seed = seed
        want.append(fr.render().cpu().numpy().copy())
    for depth in (1, 2):
        pipe = frame.FramePipeline(dev, cam, prm, depth, tile_rows=32, tile_cols=32)
        got = []
        for i, seed in enumerate(range(20, 27)):
            if i >= depth:  # the slot about to be reused holds frame i - depth
                got.append(pipe.frame(i % depth).cpu().numpy().copy())
            pipe.submit(seed)
        for i in range(7 - depth, 7):
            got.append(pipe.frame(i % depth).cpu().numpy().copy())
        assert len(got) == 7
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
    assert prm.seed == 1  # the pipeline renders from copies of the parameters
    torch.cuda.synchronize()
    dev.close()
