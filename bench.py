#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (whole node) and frame time on the Sponza-class scene at 4K,
path depth 5, tiles interleaved over N GPUs and the frame assembled on rank 0 with one gather.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms give the same line.  Called without a launcher (no WORLD_SIZE in the environment) with N > 1 -- or with
--spawn for any N -- this process touches no GPU: it starts the second form as a child process, relays rank 0's
one JSON line and exits with the child's status.

A step = one full frame: every rank renders its share of the 64x64 tiles with ONE launch of the
render kernel (inputs resident in HBM), then the packed shares are gathered to rank 0 and
un-permuted into the frame.  Rays are counted on the device (one ray = one nearest-hit query =
one radiance() invocation of the reference).  Rank 0 prints one JSON line.

Two timed regions of K frames each, the same at every N: the headline one keeps ONE frame in flight, so
`ms_per_step` = `frame_time_ms` is the time of a frame from launch to assembled frame (what the metric's "frame
time ms" means) and `value` the rate at that latency; the second keeps two frames in flight (`pipelined`: a
frame's last paths and its gather overlap the next frame's kernel), the rate of a stream of frames.
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
    if name == "sponza_tex":  # the same geometry with images on floor, ceiling and walls: the textured kernel instance
        return scenes.sponza_class(262267, sizex or 2160, sizey or 3840, textured=True), dict(rr_depth=5, rr_mode=0)
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


def _cpu_leg(orc, cam, prm, sizex, sizey, threads, budget_s, keep=None):
    """Blocks of 8x8 pixels on nested lattices (3x4, 6x8, 12x16, ... positions) over the frame until the
    time budget is used.  Returns (rays, seconds, blocks); `keep` collects (x0, y0, rgb, ids) of every block."""
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
                rgb, ids, st = orc.render_tile(cam, prm, x0, y0, min(8, sizex - x0), min(8, sizey - y0), nthreads=threads)
                t_used += time.perf_counter() - t0
                rays += st["rays"]
                blocks += 1
                if keep is not None:
                    keep.append((x0, y0, rgb, ids))
        level += 1
    return rays, t_used, blocks


def host_cores():
    visible = os.cpu_count() or 1
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = visible
    # a one-GPU job's share of the host is 16 CPUs even where the machine shows many more
    return max(1, min(allowed, int(os.environ.get("OMP_NUM_THREADS", "16") or 16))), visible, allowed


def cpu_baseline(scene, cam_od, sizex, sizey, pkw, budget_s=15.0, keep=None):
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
    cores, visible, allowed = host_cores()
    rays, t_used, blocks = _cpu_leg(orc, cam, prm, sizex, sizey, cores, budget_s, keep)
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


def oracle_sample(scene, cam_od, sizex, sizey, pkw, budget_s):
    """A short sample of oracle blocks for the parity check alone (runs with several ranks, where the CPU baseline is not taken)."""
    import pyoracle as po
    po.build()
    orc = po.OracleScene(scene)
    keep = []
    _cpu_leg(orc, po.make_camera(cam_od[0], cam_od[1], sizex, sizey), po.make_params(**pkw), sizex, sizey, host_cores()[0], budget_s, keep)
    return keep


def parity_check(frame, ids, blocks):
    """The frame the timed kernel configuration renders (same scene, size, controls; the seed of the oracle's sample),
    assembled on rank 0, against the oracle's blocks: primary-hit ids bit for bit, linear colour within 1e-9, 8-bit
    output (drawPixel, raytrace.cpp:33-39) within one step."""
    import pyoracle as po
    n = ids_eq = close = close8 = 0
    worst = 0.0
    for (x0, y0, rgb, pid) in blocks:
        r, c = pid.shape
        g = frame[x0:x0 + r, y0:y0 + c]
        gi = ids[x0:x0 + r, y0:y0 + c]
        n += r * c
        ids_eq += int((gi == pid).sum())
        d = np.abs(g - rgb).max(-1)
        worst = max(worst, float(d.max()))
        close += int((d <= 1e-9).sum())
        d8 = np.abs(po.draw_pixels(g).astype(np.int32) - po.draw_pixels(rgb).astype(np.int32)).max(-1)
        close8 += int((d8 <= 1).sum())
    return {"pixels": n, "blocks": len(blocks), "ids_equal": ids_eq == n, "ids_equal_frac": ids_eq / max(1, n), "frac_within_1e-9": close / max(1, n),
            "frac_8bit_within_1": close8 / max(1, n), "max_abs_diff": worst,
            "against": "oracle (CPU restatement of the reference path) on the same frame, seed 1"}


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_children(args, argv):
    """No GPU call has been made in this process: start `args.gpus` ranks as a child process tree (torch.distributed.run,
    one rank per GPU), relay rank 0's line, exit with the children's status."""
    import subprocess
    worker = os.environ.get("RT_BENCH_WORKER") or os.path.abspath(__file__)  # (RT_BENCH_WORKER: the launcher's own test runs a stub)
    passed = [a for a in argv if a != "--spawn"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), worker] + passed
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("RT_BENCH_WORKER", None)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in proc.stdout.decode(errors="replace").splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)  # whatever else the ranks printed
    if proc.returncode == 0 and len(lines) != 1:
        print(f"bench.py: the ranks exited cleanly but printed {len(lines)} result lines", file=sys.stderr)
        return 1
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    return proc.returncode


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
    # frames in flight per rank in the headline region: one at every N, so that a step is a frame's latency
    ap.add_argument("--in-flight", type=int, default=1)
    # ... and in the second region (0: no second region)
    ap.add_argument("--pipelined", type=int, default=2)
    ap.add_argument("--spawn", action="store_true", help="run the ranks as child processes even for one GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--light-mode", type=int, default=0)  # 1: explicit light sampling (rt_params.light_mode)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.spawn or args.gpus > 1):
        raise SystemExit(launch_children(args, sys.argv[1:]))

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
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or "MASTER_ADDR" in os.environ:
        dist.init_process_group("nccl", device_id=device)
    else:
        # one GPU runs the same program as N: a one-rank RCCL group (rendezvous through a file, no network) for the
        # barrier and the reductions; its frame needs no exchange (tests/test_gpu_rccl.py drives the gather itself
        # through a one-rank group)
        import tempfile
        store = os.path.join(tempfile.mkdtemp(prefix="rt_bench_pg_"), "store")
        dist.init_process_group("nccl", init_method="file://" + store, rank=0, world_size=1, device_id=device)
    pg_ranks = dist.get_world_size()

    scene, mode = build_workload(args.workload, args.sizex, args.sizey)
    sizex, sizey = scene["sizex"], scene["sizey"]
    host = ra.HostScene(scene)
    dev = ra.DeviceScene(host, local_rank)
    cam = ra.capi.make_camera(scene["camera"]["o"], scene["camera"]["d"], sizex, sizey)
    pkw = dict(samps=args.samps, rr_depth=mode["rr_depth"], rr_mode=mode["rr_mode"], split_depth=2, hard_depth=64, jitter=1, seed=1,
               light_mode=args.light_mode)
    prm = ra.capi.make_params(**pkw)
    fargs = dict(tile_rows=args.tile, tile_cols=args.tile, rank=rank, world=world, device=device)

    def sync_all():
        torch.cuda.synchronize(device)
        dist.barrier()
        torch.cuda.synchronize(device)

    def timed_region(depth, warmup, steps, seed0):
        """`warmup` untimed frames, then exactly `steps` frames between barrier + synchronize on both sides."""
        pipe = frame.FramePipeline(dev, cam, prm, depth, **fargs)
        for i in range(warmup):
            pipe.submit(seed0 + i)
        sync_all()
        dev.stats()  # clear the counters: only timed steps are counted below
        mk = lambda: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))  # noqa: E731
        kev, aev = [mk() for _ in range(steps)], [mk() for _ in range(steps)]
        t0 = time.perf_counter()
        for i in range(steps):
            pipe.submit(seed0 + warmup + i, kev[i], aev[i])  # a fresh random stream every frame
        sync_all()
        elapsed = time.perf_counter() - t0
        st = dev.stats()
        kern_ms = [a.elapsed_time(b) for a, b in kev]  # HIP events on the stream the kernel runs on
        asm_ms = [a.elapsed_time(b) for a, b in aev]   # gather + un-permute (rank 0), on the same stream
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        cnt = torch.tensor([float(st[k]) for k in sorted(st)], dtype=torch.float64, device=device)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        tot = dict(zip(sorted(st), [float(x) for x in cnt.tolist()]))
        per_rank = torch.zeros(world, dtype=torch.float64, device=device)
        per_rank[rank] = float(np.mean(kern_ms))
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
        del pipe
        return dict(elapsed=float(t.item()), st=st, total=tot, kern_ms=kern_ms, asm_ms=asm_ms, per_rank_kernel_ms=[float(x) for x in per_rank.tolist()])

    head = timed_region(max(1, args.in_flight), args.warmup, args.steps, 1)
    pipelined = timed_region(args.pipelined, 2, args.steps, 1000) if args.pipelined > 0 and args.pipelined != args.in_flight else None

    # ---- one untimed frame at the oracle sample's seed, ids included, through the same tiles / gather
    check_frame = check_ids = None
    if not args.no_parity_check:
        fr = frame.FrameRenderer(dev, cam, ra.capi.make_params(**pkw), want_ids=True, **fargs)
        fr.render()
        sync_all()
        dev.stats()
        if rank == 0:
            check_frame, check_ids = fr.frame.cpu().numpy(), fr.ids.cpu().numpy()
        del fr

    elapsed, st = head["elapsed"], head["st"]
    rays_total = head["total"]["rays"]
    if rank == 0:
        # roofline of the dominant kernel from THIS rank's launches of the headline region
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
        avg_ms = float(np.mean(head["kern_ms"]))
        rays_l = max(1, per_launch["rays"])
        # the render kernel instance the library picks at upload: sphere hierarchy or not; textures / light sampling; 140 slots
        # with an 8-level stack ring for trees of 28 levels and more, else 152 with a 4-level ring; one wavefront per
        # workgroup; up to three triangles behind one culling box for the deep trees, two otherwise
        deep = int(host.desc.contents.tree_depth) >= int(os.environ.get("RT_HIP_DEEP_TREE", "28"))
        feat = 2 if args.light_mode else 1 if host.desc.contents.n_textures else 0
        kernel_name = "k_main<false, %s, %d, %s>" % ("true" if host.num_spheres >= 16 else "false", feat, "140, 8, 1, 3" if deep else "152, 4, 1, 2")
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        # HBM bytes per launch from hardware counters cannot be collected inside this process; the figures of the
        # last profiled full-size launch of the same workload are kept in bench_counters.json (how: its "source"),
        # likewise the vector-instruction counters that show what binds the kernel
        traffic, hbm, valu = None, None, None
        tf = os.path.join(ROOT, "bench_counters.json")
        if os.path.exists(tf):
            with open(tf) as fh:
                tj = json.load(fh)
            if (tj["workload"], tj["sizex_rows"], tj["sizey_cols"], tj["samps_per_subpixel"]) == (args.workload, sizex, sizey, args.samps) and world == 1:
                traffic = tj["traffic_bytes_per_launch"]
                sec = avg_ms * 1e-3
                lo = tj["hbm_read_bytes_per_launch_lower"] + tj["hbm_write_bytes_per_launch"]
                hbm = {"read_bytes_lower": tj["hbm_read_bytes_per_launch_lower"], "read_bytes_upper": tj["hbm_read_bytes_per_launch"],
                       "write_bytes": tj["hbm_write_bytes_per_launch"], "GBps_lower": lo / sec / 1e9, "GBps_upper": traffic / sec / 1e9,
                       "frac_of_peak_lower": lo / sec / 1e9 / HBM_PEAK_GBS, "frac_of_peak_upper": traffic / sec / 1e9 / HBM_PEAK_GBS,
                       "measured_in_this_run": False, "source": tj["source"]}
            if tj["workload"] == args.workload and "valu" in tj:
                valu = dict(tj["valu"], measured_in_this_run=False)
        out = {
            "metric": "Mrays/s (node), Sponza-class 4K depth-5" if (args.workload, sizex, sizey) == ("sponza", 2160, 3840) else f"Mrays/s (node), {args.workload} {sizex}x{sizey}",
            "value": rays_total / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": scene["name"], "triangles": host.num_tris, "spheres": host.num_spheres, "sizex_rows": sizex,
                       "sizey_cols": sizey, "samps_per_subpixel": args.samps, "paths_per_pixel": 4 * args.samps,
                       "max_depth": mode["rr_depth"], "russian_roulette": bool(mode["rr_mode"]), "light_mode": args.light_mode, "tile": args.tile,
                       "parallelism": f"tiles%{world} + gather", "frames_in_flight": max(1, args.in_flight), "rays_per_frame": rays_total / args.steps,
                       "process_group_ranks": pg_ranks, "launched_by": "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "direct"},
            # one frame in flight: a step is a frame from launch to assembled frame on rank 0
            "frame_time_ms": elapsed / args.steps * 1e3, "frame_latency_ms": elapsed / args.steps * 1e3,
            "pipelined": None if pipelined is None else {
                "frames_in_flight": args.pipelined, "value": pipelined["total"]["rays"] / pipelined["elapsed"] / 1e6, "unit": "Mrays/s",
                "ms_per_frame": pipelined["elapsed"] / args.steps * 1e3, "steps": args.steps,
                "note": "a launch shares the chip with its neighbour for part of its run; ms_per_frame is the inverse frame rate, not a latency"},
            "per_rank": {"kernel_ms_min": min(head["per_rank_kernel_ms"]), "kernel_ms_max": max(head["per_rank_kernel_ms"]),
                         "kernel_ms": head["per_rank_kernel_ms"], "gather_unpermute_ms_rank0": float(np.mean(head["asm_ms"]))},
            # `achieved` is what the contract prices: the algorithmic bytes of the records the kernel fetches (cache hits
            # included) over the kernel's duration -- NOT HBM traffic, which `hbm_measured` gives, and not what binds the
            # kernel, which `limiter` names
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "achieved_is": "algorithmic record bytes per second, cache hits included",
                         "traffic": traffic, "hbm_measured": hbm, "kernel": kernel_name, "kernel_ms_avg": avg_ms,
                         "kernels_overlap": max(1, args.in_flight) > 1,
                         # what the counters say limits this kernel: vector-instruction issue, not bytes
                         "limiter": "valu_issue" if valu else None, "valu_issue": valu,
                         "valu_issue_frac": valu["valu_busy_frac"] if valu else None,
                         "per_ray": {"node_visits": per_launch["node_visits"] / rays_l,
                                     "node_fetches": per_launch["node_fetches"] / rays_l,
                                     "sph_tests_done": per_launch["sph_tests_done"] / rays_l,
                                     "tri_tests_reference": per_launch["tri_tests"] / rays_l,
                                     "tri_tests_done": per_launch["tri_tests_done"] / rays_l,
                                     "cull_tests": per_launch["cull_tests"] / rays_l,
                                     "sph_tests_reference": per_launch["sph_tests"] / rays_l,
                                     "sched_steps_x64": 64 * per_launch["sched_steps"] / rays_l,
                                     "alg_bytes": alg_bytes / rays_l,
                                     "reference_bytes": ref_bytes / rays_l},
                         "reference_work_GBps": ref_bytes / (avg_ms * 1e-3) / 1e9,
                         "active_lane_fraction_per_step": per_launch["sched_lanes"] / max(1, 64 * per_launch["sched_steps"]),
                         "fp64_valu": {"achieved_tflops": alg_flops / (avg_ms * 1e-3) / 1e12, "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                                       "frac": alg_flops / (avg_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS}},
        }
        cam_od = (scene["camera"]["o"], scene["camera"]["d"])
        blocks = []
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, cam_od, sizex, sizey, pkw, args.cpu_budget, blocks)
        elif not args.no_parity_check:
            blocks = oracle_sample(scene, cam_od, sizex, sizey, pkw, min(4.0, args.cpu_budget))
        if check_frame is not None and blocks:
            out["parity_check"] = parity_check(check_frame, check_ids, blocks)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()
    dev.close()


if __name__ == "__main__":
    main()
