#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (whole node) and frame time on the Sponza-class scene at 4K,
path depth 5, tiles interleaved over N GPUs and the frame assembled on rank 0 with one gather.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one full frame: every rank renders its share of the 64x64 tiles with ONE launch of the
render kernel (inputs resident in HBM), then the packed shares are gathered to rank 0 and
un-permuted into the frame.  Rays are counted on the device (one ray = one nearest-hit query =
one radiance() invocation of the reference).  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

# Algorithmic traffic per unit of work, FP64 layout of the reference's records actually needed:
B_NODE = 72   # box 48 + split 8 + axis/children/list range 16, per node whose records are fetched (a visit that the
              # parent's step decides from the child's box fetches nothing and is priced at nothing)
B_TRI = 72    # three FP64 vertices, per Polygon::intersect
B_SHADE = 76  # nc, col, lig, type, per shaded triangle hit
B_SPH = 32    # centre + radius, per Sphere::intersect executed
B_CULL = 64   # base and step 24 + 24 plane bytes of four boxes + four links 16, per culling record visited (long own lists)
# FP64 operation counts per unit of work (adds/multiplies/compares; divides and sqrt counted once)
F_NODE, F_TRI, F_SPH, F_SHADE, F_CULL = 40, 45, 20, 100, 96
HBM_PEAK_GBS = 8000.0
FP64_VALU_PEAK_TFLOPS = 78.6


def build_workload(name, sizex, sizey):
    from raytrace_amd import scenes
    if name == "sponza":
        return scenes.sponza_class(262267, sizex or 2160, sizey or 3840), dict(rr_depth=5, rr_mode=0)
    if name == "bunny":
        return scenes.bunny_class(69451, sizex or 1080, sizey or 1920), dict(rr_depth=2, rr_mode=0)
    if name == "spheres":
        return scenes.random_spheres(10000, sizex or 1080, sizey or 1920), dict(rr_depth=8, rr_mode=0)
    if name == "lucy":
        return scenes.lucy_class(28055742, sizex or 2160, sizey or 3840), dict(rr_depth=2, rr_mode=0)
    if name == "cornell":
        return scenes.cornell_demo(sizex or 600, sizey or 800), dict(rr_depth=5, rr_mode=1)
    if name == "sponza_small":
        return scenes.sponza_class(40000, sizex or 270, sizey or 480, detail=0.3), dict(rr_depth=5, rr_mode=0)
    raise SystemExit(f"unknown workload {name}")


def _cpu_leg(orc, cam, prm, sizex, sizey, threads, budget_s):
    """Blocks of 8x8 pixels on nested lattices (3x4, 6x8, 12x16, ... positions) over the frame until the
    time budget is used.  Returns (rays, seconds, blocks)."""
    rays, t_used, blocks, level = 0, 0.0, 0, 0
    seen = set()
    while t_used < budget_s and level < 6:
        ni, nj = 3 << level, 4 << level
        for i in range(ni):
            for j in range(nj):
                x0, y0 = int((i + 0.5) * sizex / ni) // 8 * 8, int((j + 0.5) * sizey / nj) // 8 * 8
                if (x0, y0) in seen or t_used >= budget_s:
                    continue
                seen.add((x0, y0))
                t0 = time.perf_counter()
                _, _, st = orc.render_tile(cam, prm, x0, y0, min(8, sizex - x0), min(8, sizey - y0), nthreads=threads)
                t_used += time.perf_counter() - t0
                rays += st["rays"]
                blocks += 1
        level += 1
    return rays, t_used, blocks


def cpu_baseline(scene, cam_od, sizex, sizey, pkw, budget_s=15.0):
    """The oracle (CPU restatement of the reference path) on this machine's host cores, on a bounded
    sample of the same workload.  Three legs: all the cores this job may use (the headline figure),
    ONE thread (how a rank of the reference runs: its OpenMP pragma is commented out,
    raytrace.cpp:185), and one thread of a build with the reference's own flags (-Ofast
    -march=native, CMakeLists.txt:30; not IEEE-reproducible, for information)."""
    import pyoracle as po
    po.build()
    orc = po.OracleScene(scene)
    cam = po.make_camera(cam_od[0], cam_od[1], sizex, sizey)
    prm = po.make_params(**pkw)
    visible = os.cpu_count() or 1
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = visible
    # a one-GPU job's share of the host is 16 CPUs even where the machine shows many more
    cores = max(1, min(allowed, int(os.environ.get("OMP_NUM_THREADS", "16") or 16)))
    rays, t_used, blocks = _cpu_leg(orc, cam, prm, sizex, sizey, cores, budget_s)
    r1, t1, b1 = _cpu_leg(orc, cam, prm, sizex, sizey, 1, budget_s / 3)
    out = {"value": rays / t_used / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
           "host_cpus_visible": visible, "host_cpus_allowed": allowed,
           "sample": f"{blocks} blocks of 8x8 pixels on nested lattices over the same {sizex}x{sizey} frame, same scene/params, "
                     f"{rays} rays in {t_used:.1f} s, pixel-parallel pthreads",
           "single_thread": {"value": r1 / t1 / 1e6, "unit": "Mrays/s", "cores": 1,
                             "sample": f"{b1} blocks of 8x8 pixels, {r1} rays in {t1:.1f} s"}}
    try:
        fast = po.OracleScene(scene, library=po.build_fast())
        r2, t2, b2 = _cpu_leg(fast, cam, prm, sizex, sizey, 1, budget_s / 3)
        out["single_thread_ofast_native"] = {"value": r2 / t2 / 1e6, "unit": "Mrays/s", "cores": 1,
                                             "flags": "-Ofast -march=native (the reference's CMakeLists.txt:30)",
                                             "sample": f"{b2} blocks of 8x8 pixels, {r2} rays in {t2:.1f} s"}
    except Exception as e:  # no compiler on this host: the leg is informational
        out["single_thread_ofast_native"] = {"value": None, "error": str(e)[:200]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0)  # 0: 5 frames on one GPU, 5 per rank on several (a rank's share of a frame is short)
    ap.add_argument("--warmup", type=int, default=3)  # the engine clock needs about half a second of load to settle
    ap.add_argument("--workload", default="sponza")
    ap.add_argument("--sizex", type=int, default=0)
    ap.add_argument("--sizey", type=int, default=0)
    ap.add_argument("--samps", type=int, default=1)
    ap.add_argument("--tile", type=int, default=64)
    # frames in flight per rank (0: one on one GPU, where the kernel's own duration is what the roofline is made of;
    # two on several, where a rank's share of the frame is short and its tail and the gather overlap the next frame)
    ap.add_argument("--in-flight", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    args = ap.parse_args()

    # stdout carries exactly one line, the result: whatever libraries print there (RCCL's version banner at the
    # first collective, for one) goes to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import raytrace_amd as ra
    from raytrace_amd import frame

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.steps <= 0:
        args.steps = 5 * world
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    pg_mode = os.environ.get("RT_BENCH_PG", "early")  # experiment knob: early | late | none (one rank only)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)
    elif pg_mode == "early":
        # one GPU runs the same program as N: a one-rank RCCL group (rendezvous through a file, no network) for the
        # barrier and the reductions; its frame needs no exchange (tests/test_gpu_rccl.py drives the gather itself
        # through a one-rank group)
        import tempfile
        store = os.path.join(tempfile.mkdtemp(prefix="rt_bench_pg_"), "store")
        dist.init_process_group("nccl", init_method="file://" + store, rank=0, world_size=1, device_id=device)

    scene, mode = build_workload(args.workload, args.sizex, args.sizey)
    sizex, sizey = scene["sizex"], scene["sizey"]
    host = ra.HostScene(scene)
    dev = ra.DeviceScene(host, local_rank)
    cam = ra.capi.make_camera(scene["camera"]["o"], scene["camera"]["d"], sizex, sizey)
    pkw = dict(samps=args.samps, rr_depth=mode["rr_depth"], rr_mode=mode["rr_mode"], split_depth=2, hard_depth=64, jitter=1, seed=1)
    prm = ra.capi.make_params(**pkw)
    in_flight = args.in_flight if args.in_flight > 0 else (1 if world == 1 else 2)
    pipe = frame.FramePipeline(dev, cam, prm, in_flight, tile_rows=args.tile, tile_cols=args.tile, rank=rank, world=world, device=device)
    if world == 1 and pg_mode == "late":
        import tempfile
        store = os.path.join(tempfile.mkdtemp(prefix="rt_bench_pg_"), "store")
        dist.init_process_group("nccl", init_method="file://" + store, rank=0, world_size=1, device_id=device)
    have_pg = dist.is_initialized()

    def sync_all():
        torch.cuda.synchronize(device)
        if have_pg:
            dist.barrier()
            torch.cuda.synchronize(device)

    def step(i, ev=None):
        pipe.submit(1 + i, ev)  # a fresh random stream every frame

    for i in range(args.warmup):
        step(i)
    sync_all()
    dev.stats()  # clear the counters: only timed steps are counted below
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, events[i])
    sync_all()
    elapsed = time.perf_counter() - t0
    st = dev.stats()
    kern_ms = [a.elapsed_time(b) for a, b in events]  # HIP events on the stream the kernel runs on

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    cnt = torch.tensor([st["rays"], st["node_visits"], st["tri_tests"], st["sph_tests"], st["tri_shades"]], dtype=torch.float64, device=device)
    if have_pg:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    rays_total = float(cnt[0].item())

    if rank == 0:
        # roofline of the dominant kernel (k_render) from THIS rank's launches
        launches = max(1, args.steps)
        per_launch = {k: st[k] / launches for k in st}
        # bytes/flops of the work the kernel EXECUTES (long own lists are culled); the work the
        # reference would do for the same rays (every own-list triangle tested) is reported beside it
        alg_bytes = (per_launch["node_fetches"] * B_NODE + per_launch["tri_tests_done"] * B_TRI + per_launch["cull_tests"] * B_CULL
                     + per_launch["tri_shades"] * B_SHADE + per_launch["sph_tests_done"] * B_SPH)
        alg_flops = (per_launch["node_fetches"] * F_NODE + per_launch["tri_tests_done"] * F_TRI + per_launch["cull_tests"] * F_CULL
                     + per_launch["sph_tests_done"] * F_SPH + per_launch["rays"] * F_SHADE)
        ref_bytes = (per_launch["node_visits"] * B_NODE + per_launch["tri_tests"] * B_TRI + per_launch["tri_shades"] * B_SHADE
                     + per_launch["sph_tests"] * B_SPH)
        avg_ms = float(np.mean(kern_ms))
        # the render kernel instance the library picks at upload: sphere hierarchy or not; 140 slots with an
        # 8-level stack ring for trees of 28 levels and more, else 152 with a 4-level ring; one wavefront per workgroup;
        # up to three triangles behind one culling box for the deep trees, two otherwise
        deep = int(host.desc.contents.tree_depth) >= int(os.environ.get("RT_HIP_DEEP_TREE", "28"))
        kernel_name = "k_main<false, %s, %d, %s>" % ("true" if host.num_spheres >= 16 else "false", 1 if host.desc.contents.n_textures else 0,
                                                    "140, 8, 1, 3" if deep else "152, 4, 1, 2")
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        # HBM bytes per launch from hardware counters cannot be collected inside this process; the figure of the
        # last profiled full-size launch of the same workload is kept in bench_counters.json (how: its "source")
        # (likewise the vector-instruction counters that show what binds the kernel)
        traffic, traffic_source, valu = None, None, None
        tf = os.path.join(ROOT, "bench_counters.json")
        if os.path.exists(tf):
            with open(tf) as fh:
                tj = json.load(fh)
            if (tj["workload"], tj["sizex_rows"], tj["sizey_cols"], tj["samps_per_subpixel"]) == (args.workload, sizex, sizey, args.samps) and world == 1:
                traffic, traffic_source = tj["traffic_bytes_per_launch"], tj["source"]
            if tj["workload"] == args.workload and "valu" in tj:
                valu = dict(tj["valu"], measured_in_this_run=False)
        out = {
            "metric": "Mrays/s (node), Sponza-class 4K depth-5" if (args.workload, sizex, sizey) == ("sponza", 2160, 3840) else f"Mrays/s (node), {args.workload} {sizex}x{sizey}",
            "value": rays_total / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": scene["name"], "triangles": host.num_tris, "spheres": host.num_spheres, "sizex_rows": sizex,
                       "sizey_cols": sizey, "samps_per_subpixel": args.samps, "paths_per_pixel": 4 * args.samps,
                       "max_depth": mode["rr_depth"], "russian_roulette": bool(mode["rr_mode"]), "tile": args.tile,
                       "parallelism": f"tiles%{world} + gather", "frames_in_flight": in_flight, "rays_per_frame": rays_total / args.steps},
            "frame_time_ms": elapsed / args.steps * 1e3,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_measured_in_this_run": False if traffic is not None else None,
                         "traffic_source": traffic_source, "kernel": kernel_name, "kernel_ms_avg": avg_ms,
                         # with two frames in flight a launch shares the chip with its neighbours for part of its run
                         "kernels_overlap": in_flight > 1,
                         # what the counters say limits this kernel: vector-instruction issue, not bytes
                         "limiter": "valu_issue" if valu else None, "valu_issue": valu,
                         "per_ray": {"node_visits": per_launch["node_visits"] / max(1, per_launch["rays"]),
                                     "node_fetches": per_launch["node_fetches"] / max(1, per_launch["rays"]),
                                     "sph_tests_done": per_launch["sph_tests_done"] / max(1, per_launch["rays"]),
                                     "tri_tests_reference": per_launch["tri_tests"] / max(1, per_launch["rays"]),
                                     "tri_tests_done": per_launch["tri_tests_done"] / max(1, per_launch["rays"]),
                                     "cull_tests": per_launch["cull_tests"] / max(1, per_launch["rays"]),
                                     "sph_tests_reference": per_launch["sph_tests"] / max(1, per_launch["rays"]),
                                     "sched_steps_x64": 64 * per_launch["sched_steps"] / max(1, per_launch["rays"]),
                                     "alg_bytes": alg_bytes / max(1, per_launch["rays"]),
                                     "reference_bytes": ref_bytes / max(1, per_launch["rays"])},
                         "reference_work_GBps": ref_bytes / (avg_ms * 1e-3) / 1e9,
                         "active_lane_fraction_per_step": per_launch["sched_lanes"] / max(1, 64 * per_launch["sched_steps"]),
                         "fp64_valu": {"achieved_tflops": alg_flops / (avg_ms * 1e-3) / 1e12, "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                                       "frac": alg_flops / (avg_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, (scene["camera"]["o"], scene["camera"]["d"]), sizex, sizey, pkw, args.cpu_budget)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
    if have_pg:
        dist.destroy_process_group()
    dev.close()


if __name__ == "__main__":
    main()
