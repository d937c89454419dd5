"""The scenes bench.py times, at their stated sizes, against the CPU oracle through the C ABI: bunny-class 69 451 triangles
(depth 2), Sponza-class 262 267 (depth 5) and Lucy-class 28 055 742 (depth 2, the deep-tree kernel build with its
three-triangle groups and a record pool near the index limit).  The other GPU tests use smaller stand-ins of the same
generators; these tie the checks to the very trees, lists and culling hierarchies that are timed.  Bars as everywhere:
ids, distances and work counters of explicit rays bit-exact; a frame's primary ids bit-exact and its colours within
1e-9 on >= 99.9 % of the pixels."""
import os
import time

import numpy as np
import pytest

from conftest import random_rays
from test_gpu_configs import _same_frame, _same_queries

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _progress(msg):
    """A line in gpurun_out/ now and then: the Lucy-class test runs for minutes without pytest printing anything."""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "timed_scenes_progress.log"), "a") as f:
            f.write(f"{time.strftime('%H:%M:%S')} {msg}\n")
    except OSError:
        pass


def _aimed_rays(seed, n, eye, eye_spread, target, target_spread):
    rs = np.random.RandomState(seed)
    o = np.asarray(eye, np.float64) + rs.uniform(-eye_spread, eye_spread, (n, 3))
    d = np.asarray(target, np.float64) + rs.uniform(-target_spread, target_spread, (n, 3)) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], 1)


def test_bunny_class_at_its_benchmark_size(ra, po):
    from raytrace_amd import scenes
    sc = scenes.bunny_class(69451, 27, 48)
    hs = ra.HostScene(sc)
    dev = ra.DeviceScene(hs, 0)
    orc = po.OracleScene(sc)
    rays = np.concatenate([_aimed_rays(3, 2000, sc["camera"]["o"], 4.0, (50, 30, 80), 26.0), random_rays(4, 1000, np.array([50, 40, 100.0]), 120.0)])
    a = _same_queries(dev, orc, rays)
    assert (a[0] >= 0).mean() > 0.5
    _same_frame(ra, po, dev, orc, sc, 27, 48, dict(samps=1, rr_depth=2, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=1), nthreads=16)
    dev.close()


def test_sponza_class_at_its_benchmark_size(ra, po):
    from raytrace_amd import scenes
    sc = scenes.sponza_class(262267, 27, 48)
    hs = ra.HostScene(sc)
    dev = ra.DeviceScene(hs, 0)
    orc = po.OracleScene(sc)
    rays = np.concatenate([_aimed_rays(5, 1500, sc["camera"]["o"], 10.0, (50, 35, 120), 70.0), random_rays(6, 1500, np.array([50, 40, 140.0]), 60.0)])
    a = _same_queries(dev, orc, rays)
    assert (a[0] >= 0).mean() > 0.9  # a closed room
    _same_frame(ra, po, dev, orc, sc, 27, 48, dict(samps=1, rr_depth=5, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=1), nthreads=16)
    dev.close()


def test_lucy_class_at_its_benchmark_size(ra, po):
    """28 055 742 triangles: explicit rays and a small frame against the oracle; the same frame as one tile, as ragged
    tiles, rendered twice, and as single-pixel launches must agree to the last bit."""
    from raytrace_amd import frame, scenes
    _progress("lucy: generating")
    sc = scenes.lucy_class(28055742, 24, 40)
    _progress("lucy: host tree")
    hs = ra.HostScene(sc)
    assert hs.num_tris == 28055742 and int(hs.desc.contents.tree_depth) >= 28  # the deep-tree kernel build
    _progress("lucy: upload")
    dev = ra.DeviceScene(hs, 0)
    kw = dict(samps=1, rr_depth=2, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=1)
    cam, prm = ra.capi.make_camera(sc["camera"]["o"], sc["camera"]["d"], 24, 40), ra.capi.make_params(**kw)
    whole, wid, wst = dev.render_tile(cam, prm, 0, 0, 24, 40)
    again, aid, ast = dev.render_tile(cam, prm, 0, 0, 24, 40)
    work = lambda st: {k: v for k, v in st.items() if not k.startswith("sched_")}  # noqa: E731  (scheduler steps depend on timing)
    assert np.array_equal(whole, again) and np.array_equal(wid, aid) and work(wst) == work(ast)
    import torch
    for tr, tc in ((7, 9), (16, 16)):  # ragged tilings through the device-buffer entry point
        tiles = frame.plan_tiles(24, 40, tr, tc)
        buf = torch.zeros((24 * 40, 3), dtype=torch.float64, device="cuda:0")
        ids = torch.zeros((24 * 40,), dtype=torch.int32, device="cuda:0")
        dev.render_tiles_device(cam, prm, tiles, buf.data_ptr(), ids.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = frame.scatter_to_frame(np.zeros((24, 40, 3)), tiles, buf.cpu().numpy())
        gid = frame.scatter_to_frame(np.zeros((24, 40), np.int32), tiles, ids.cpu().numpy())
        assert np.array_equal(got, whole) and np.array_equal(gid, wid)
    for (x, y) in ((0, 0), (11, 20), (13, 19), (23, 39)):  # launches of one pixel
        px, pid, _ = dev.render_tile(cam, prm, x, y, 1, 1)
        assert np.array_equal(px[0, 0], whole[x, y]) and pid[0, 0] == wid[x, y]
    _progress("lucy: oracle tree")
    orc = po.OracleScene(sc)
    _progress("lucy: oracle rays")
    rays = np.concatenate([_aimed_rays(7, 200, sc["camera"]["o"], 4.0, (50, 30, 80), 26.0), random_rays(8, 100, np.array([50, 40, 100.0]), 120.0)])
    a = _same_queries(dev, orc, rays)
    assert (a[0] >= 0).mean() > 0.5
    _progress("lucy: oracle frame")
    ocam = po.make_camera(sc["camera"]["o"], sc["camera"]["d"], 24, 40)
    orgb, opid, ost = orc.render_tile(ocam, po.make_params(**kw), 0, 0, 24, 40, nthreads=16)
    assert np.array_equal(wid, opid)
    assert (np.abs(whole - orgb).max(-1) <= 1e-9).mean() >= 0.999
    assert abs(wst["rays"] - ost["rays"]) <= 1e-3 * ost["rays"] + 2
    _progress("lucy: done")
    dev.close()


def test_one_prepared_scene_uploads_twice_to_identical_frames(ra):
    """rt_prepare_scene builds the device layout once on the host; every rt_upload_prepared of it gives a scene that
    renders the same frame as rt_upload_scene's, bit for bit (what a host with several GPUs relies on)."""
    from raytrace_amd import scenes
    sc = scenes.sponza_class(40000, 40, 64, detail=0.3)
    hs = ra.HostScene(sc)
    kw = dict(samps=1, rr_depth=5, rr_mode=0, split_depth=2, hard_depth=64, jitter=1, seed=3)
    cam, prm = ra.capi.make_camera(sc["camera"]["o"], sc["camera"]["d"], 40, 64), ra.capi.make_params(**kw)
    plain = ra.DeviceScene(hs, 0)
    want = plain.render_tile(cam, prm, 0, 0, 40, 64)
    plain.close()
    prep = ra.PreparedScene(hs)
    a, b = ra.DeviceScene(prep, 0), ra.DeviceScene(prep, 0)  # both alive at once, as on two GPUs
    prep.close()  # the uploads own their copies
    for d in (a, b):
        got = d.render_tile(cam, prm, 0, 0, 40, 64)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        assert {k: v for k, v in got[2].items() if not k.startswith("sched_")} == {k: v for k, v in want[2].items() if not k.startswith("sched_")}
        d.close()
    with pytest.raises(ra.capi.RtError, match="not in"):
        ra.DeviceScene(ra.PreparedScene(hs), 4096)


def test_kernel_time_is_reported_per_launch_with_two_frames_in_flight(ra):
    """rt_last_kernel_ms belongs to the launch queued last, even while another launch of the scene runs on a second stream."""
    import torch
    from raytrace_amd import frame, scenes
    sc = scenes.cornell_demo()
    hs = ra.HostScene(sc)
    dev = ra.DeviceScene(hs, 0)
    kw = dict(samps=1, rr_depth=5, rr_mode=1, split_depth=2, hard_depth=64, jitter=1, seed=2)
    prm = ra.capi.make_params(**kw)
    big, small = ra.capi.make_camera(sc["camera"]["o"], sc["camera"]["d"], 600, 800), ra.capi.make_camera(sc["camera"]["o"], sc["camera"]["d"], 16, 16)
    b1 = torch.zeros((600 * 800, 3), dtype=torch.float64, device="cuda:0")
    b2 = torch.zeros((16 * 16, 3), dtype=torch.float64, device="cuda:0")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    dev.render_tiles_device(small, prm, [(0, 0, 16, 16)], b2.data_ptr(), None, s2.cuda_stream)  # the second stream brings the second set
    torch.cuda.synchronize()
    ms_small_alone = dev.last_kernel_ms()
    dev.render_tiles_device(big, prm, frame.plan_tiles(600, 800), b1.data_ptr(), None, s1.cuda_stream)
    torch.cuda.synchronize()
    ms_big_alone = dev.last_kernel_ms()
    assert 0 < ms_small_alone < 0.5 * ms_big_alone, (ms_small_alone, ms_big_alone)
    # both in flight, the big one queued last: the figure is the big one's (each set has its own pair of events)
    dev.render_tiles_device(small, prm, [(0, 0, 16, 16)], b2.data_ptr(), None, s2.cuda_stream)
    dev.render_tiles_device(big, prm, frame.plan_tiles(600, 800), b1.data_ptr(), None, s1.cuda_stream)
    ms_last = dev.last_kernel_ms()
    assert ms_last > 0.5 * ms_big_alone, (ms_last, ms_big_alone)
    # the other way round the small launch is the last one: it may have had to wait for the big one's workgroups to leave
    # the chip (the grid is persistent), but its figure is its own pair's, bounded by both
    dev.render_tiles_device(big, prm, frame.plan_tiles(600, 800), b1.data_ptr(), None, s1.cuda_stream)
    dev.render_tiles_device(small, prm, [(0, 0, 16, 16)], b2.data_ptr(), None, s2.cuda_stream)
    ms_last = dev.last_kernel_ms()
    torch.cuda.synchronize()
    assert 0 < ms_last < 2.5 * ms_big_alone, (ms_last, ms_big_alone)
    dev.close()
