#!/usr/bin/env python3
"""Benchmark of the render path: Mrays/s and ms/frame.

  python bench.py --gpus N --steps K --warmup W [--workload all|bunny|cornell|bmw|envmap|glass]

A step is one frame: Renderer::render(spp) over the whole image (src/renderer.cpp:27-52), i.e. one pass of
the path kernel, with the scene already resident in HBM.  A ray is one closest-hit query
(Scene::find_intersection, src/scene.cpp:19): a path segment or a shadow ray that is actually traced; the
count comes from the kernel's own counters.

The top-level numbers are the `bunny` workload (Cornell box + 81,920-triangle mesh, 1024 x 1024, 1024 spp:
the configuration the metric names for one GPU).  With the default `--workload all` on one GPU the Cornell box
and the 1.3 M-triangle scene (the two scenes the metric is quoted on) are measured in the same run and
reported under "extra", each with its own roofline and CPU baseline.

N > 1: one process per GPU.  `python bench.py --gpus N` starts the N ranks itself (children under
torch.distributed.run, before this process touches a GPU); under torch.distributed.run it is a rank.  The accumulation
buffers are merged onto rank 0 with ONE RCCL collective per frame, inside the timed region; its bytes and time are also
reported by themselves ("merge").  With the default `--workload all` the N > 1 run times the three multi-GPU
configurations the metric names, each sharing ONE fixed frame (strong scaling: the frame is the same whatever N):
  top level  `bmw`    1.31 M triangles, 3840 x 2160, 1024 spp, interleaved 16-row bands per GPU + a gather
  extra      `glass`  dielectric scene, 500 x 500, 8 bounces, 8192 spp shared by sample index + a sum-reduce
  extra      `envmap` textured scene + environment map + depth of field, 1920 x 1080, 1024 spp shared by sample index
A named workload (`--workload bunny --gpus N`) is shared by --split (default spp) under --scaling (default weak for
spp: every GPU renders `spp` samples, the frame gets N x spp; strong: the frame's `spp` samples are shared).

Roofline: nothing here is a dense contraction.  Two fractions are computed and the larger is reported as "frac":
the fp32 vector rate (algorithmic flops per ray: 21 per box test, 60 per primitive test, 250 per shaded segment;
tests per ray counted by an instrumented launch on the same scene; x rays per launch / the kernel's launch time from
HIP events on its own stream, against 157.3 TFLOP/s), and the measured HBM-side traffic per launch from the hardware
counters (profiles/traffic.json) over the launch time against 8 TB/s ("hbm_frac").  The same accounting in bytes
(32 per box test, 36 per primitive test, 48 per ray) is "algorithmic_gbs", and "bytes_frac_survey" is that figure over
the HBM peak: it exceeds 1 because those bytes are what the lanes ask the CACHES for (L1 / L2 / Infinity Cache serve
most of them), so it measures cache-level demand, not HBM use, and is not the reported fraction.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    "bunny": "Cornell box + 81,920-triangle smooth mesh (bunny-class stand-in: diffuse + emissive area light)",
    "cornell": "Cornell box OBJ (12 triangles, textured floor, depth of field)",
    "bmw": "Cornell box + 1,310,720-triangle mesh (BMW-class stand-in)",
    "envmap": "Cornell box (textured floor) + 81,920-triangle mesh, 3584x1792 environment map, depth of field",
    "glass": "Cornell box + dielectric 20,480-triangle mesh + glass and mirror spheres (Russian roulette active)",
}
# per-workload defaults for (width, height, spp, bounces)
WORKLOAD_DEFAULTS = {"envmap": (1920, 1080, 1024, 3), "glass": (500, 500, 2048, 8)}
PEAK_FP32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def build_scene(name, width, height):
    from pathtracer_amd import scenes
    if name == "bunny":
        return scenes.cornell_with_mesh(6, width, height)
    if name == "bmw":
        return scenes.cornell_with_mesh(8, width, height)
    if name == "cornell":
        return scenes.cornell_box(width, height)
    if name == "envmap":
        return scenes.cornell_with_mesh(6, width, height, envmap=scenes.procedural_envmap())
    if name == "glass":
        return scenes.glass_scene(5, width, height)
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(sd, spp_max, bounces, budget_s):
    """The CPU oracle (a restatement of the reference renderer) timed on this box's host cores on a bounded
    sample of the same workload: a band of rows in the middle of the frame, sized from a one-sample probe so
    that the timed run takes about budget_s seconds.  Also its single-thread rate on a smaller band, to set
    beside the reference's own single-thread rate (0.908 M trace_ray bodies per second on the Cornell box)."""
    from oracle import orc
    o = orc.OracleScene(sd)
    h = sd.camera.height
    threads = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or min(16, os.cpu_count() or 1)
    band = probe_band = max(8, min(h, 64))
    y0 = (h - band) // 2
    t = time.perf_counter()
    o.render(1, bounces, seed=99, rows=(y0, y0 + band), threads=threads)
    t1 = max(time.perf_counter() - t, 1e-4)  # one sample per pixel of the probe band
    spp = int(max(1, min(spp_max, budget_s / t1)))
    if spp == spp_max:  # the band at full spp is still short: widen it
        band = int(max(band, min(h, band * budget_s / (t1 * spp))))
        y0 = (h - band) // 2
    t = time.perf_counter()
    _, c = o.render(spp, bounces, seed=100, rows=(y0, y0 + band), threads=threads)
    dt = time.perf_counter() - t
    # one thread on eight rows, about a seventh of the budget (t1 was `threads` threads on probe_band rows)
    spp1 = int(max(1, min(spp_max, budget_s * 0.15 / (t1 * threads * 8.0 / probe_band))))
    y1 = (h - 8) // 2
    t = time.perf_counter()
    _, c1 = o.render(spp1, bounces, seed=101, rows=(y1, y1 + 8), threads=1)
    dt1 = max(time.perf_counter() - t, 1e-6)
    return {"value": c["queries"] / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"rows {y0}..{y0 + band} of the frame, {spp} spp, {bounces} bounces, "
                      f"{c['queries']} rays in {dt:.2f} s",
            "segments_per_s": c["segments"] / dt, "prim_tests_per_ray": c["prim_tests"] / max(1, c["queries"]),
            "box_tests_per_ray": c["box_tests"] / max(1, c["queries"]),
            "single_thread": {"segments_per_s": c1["segments"] / dt1, "rays_per_s": c1["queries"] / dt1,
                              "sample": f"rows {y1}..{y1 + 8}, {spp1} spp, {dt1:.2f} s",
                              "reference_single_thread_segments_per_s_cornell": 0.908e6,
                              "explanation": "this restatement runs faster per thread than the reference binary (0.908 M segments/s on the Cornell box, "
                                             "8-core Xeon VM of the survey) because it has no process-global state: the reference draws every random "
                                             "number from one shared std::mt19937 (src/util.h:72-78, two engine draws per double) and bumps two global "
                                             "atomics per primitive test and per segment (src/geometry.cpp:162, src/renderer.cpp:63); here the generator "
                                             "is a counter hash and the counters are per thread.  The algorithm (tree, tests per ray, estimator) is the "
                                             "reference's; the host CPU also differs from the survey container's"}}


def traffic_entry(key):
    """HBM-side bytes per launch of exactly this configuration from the hardware-counter passes whose summary is
    committed under profiles/ (tools/profile.sh + tools/make_traffic.py); None when it was not profiled."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(path)).get(key)
    except Exception:
        return None


def run_workload(name, args, ctx, torch, dist, dev, rank, world, with_cpu, split, scaling):
    """Times `args.steps` frames of one workload; returns the result dictionary (rank 0) or None.
    split: how the ranks share a frame; scaling "strong": the frame has `spp` samples whatever the number of ranks,
    "weak" (sample split only): every rank renders `spp` samples."""
    from pathtracer_amd import parallel
    width, height, spp, bounces = args.sizes[name]
    if world > 1 and split == "spp" and scaling == "strong":
        mine = (spp - rank + world - 1) // world  # samples rank, rank + world, ... below spp
    else:
        mine = spp
    frame_spp = spp * world if (world > 1 and split == "spp" and scaling == "weak") else spp
    sd = build_scene(name, width, height)
    ctx.upload_scene(sd, args.builder)
    ctx.set_camera(sd.camera)
    info = ctx.scene_info()
    accum = torch.zeros((height, width, 4), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx.bind_accum(accum.data_ptr())
    plan = parallel.frame_plan(split, rank, world, mine, height, batch=args.batch)
    merge = parallel.FrameMerge(split, rank, world, accum)
    merge_s = [0.0]

    def frame(step):
        accum.zero_()
        torch.cuda.synchronize()  # the kernel runs on the library's own stream
        for c in plan:
            ctx.render(c.samples, bounces, seed=args.seed + step, sample_first=c.sample_first,
                       sample_stride=c.sample_stride, row_begin=c.row_begin, row_end=c.row_end,
                       band_rows=c.band_rows, band_first=c.band_first, band_stride=c.band_stride)
        ctx.synchronize()
        if world > 1:
            dist.barrier()  # the merge is timed from the moment the slowest rank has rendered
            torch.cuda.synchronize()
            t_merge = time.perf_counter()
        merge.run()  # RCCL over xGMI: the frame's one exchange step
        if world > 1:
            torch.cuda.synchronize()
            merge_s[0] += time.perf_counter() - t_merge

    for w in range(args.warmup):
        frame(-1 - w)
    merge_s[0] = 0.0
    ctx.reset_stats()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        frame(s)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    st = ctx.stats()

    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    counts = torch.tensor([float(st["queries"]), float(st["segments"]), float(st["camera_rays"])], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    elapsed = float(tt.item())
    queries, segments, camera_rays = [float(x) for x in counts.tolist()]

    # instrumented pass on the same scene: box / primitive tests per ray for the algorithmic-work figures
    ctx.reset_stats()
    accum.zero_()
    torch.cuda.synchronize()
    ctx.render(max(1, min(args.stats_spp, spp)), bounces, seed=7, collect_stats=True, row_end=min(height, 1024))
    si = ctx.stats()
    ctx.bind_accum(None)
    box_per_ray = si["box_tests"] / max(1, si["queries"])
    prim_per_ray = si["prim_tests"] / max(1, si["queries"])
    seg_per_ray = si["segments"] / max(1, si["queries"])
    if rank != 0:
        return None

    cpu = cpu_baseline(sd, spp, bounces, args.cpu_seconds) if with_cpu else None
    # Work is counted per ray as the smaller of what this tree and the reference's tree need: a test that the
    # reference would not have made is not algorithmic work (only the 5-node Cornell box does more box tests here,
    # because a node visit tests both children; the reference's counts come from the CPU leg, when it ran).
    box_counted, prim_counted = box_per_ray, prim_per_ray
    if cpu:
        box_counted = min(box_per_ray, cpu["box_tests_per_ray"])
        prim_counted = min(prim_per_ray, cpu["prim_tests_per_ray"])
    bytes_per_ray = box_counted * 32.0 + prim_counted * 36.0 + 48.0
    flops_per_ray = box_counted * 21.0 + prim_counted * 60.0 + 250.0 * seg_per_ray
    rays_per_launch = st["queries"] / max(1, st["launches"])
    kernel_s = (st["kernel_ms_total"] / max(1, st["launches"])) / 1e3
    achieved_tf = rays_per_launch * flops_per_ray / kernel_s / 1e12
    algorithmic_gbs = rays_per_launch * bytes_per_ray / kernel_s / 1e9
    tr = traffic_entry(f"{name}_{width}x{height}_{spp}spp")
    traffic = tr.get("hbm_bytes_per_launch") if tr else None
    valu_frac = achieved_tf / PEAK_FP32_TFLOPS
    hbm_frac = (traffic / kernel_s / 1e9 / PEAK_HBM_GBS) if traffic else None
    roofline = {"bound": "fp32_valu", "achieved": achieved_tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": valu_frac,
                "traffic": traffic, "hbm_frac": hbm_frac, "kernel": "render_kernel", "kernel_ms_avg": kernel_s * 1e3,
                "flops_per_ray": flops_per_ray, "bytes_per_ray": bytes_per_ray, "box_tests_per_ray": box_per_ray,
                "prim_tests_per_ray": prim_per_ray, "box_tests_counted": box_counted, "prim_tests_counted": prim_counted,
                "rays_per_launch": rays_per_launch,
                "algorithmic_gbs": algorithmic_gbs, "algorithmic_bytes_per_launch": rays_per_launch * bytes_per_ray,
                "bytes_frac_survey": algorithmic_gbs / PEAK_HBM_GBS,
                "bytes_frac_survey_note": "algorithmic bytes (32 per box test, 36 per primitive test, 48 per ray) over the HBM peak: "
                                          "what the lanes ask the caches for, most of it served by L1 / L2 / Infinity Cache; above 1 by "
                                          "construction on these scenes and NOT the reported fraction",
                "note": "the scene fits the L2s (32 MB) and the Infinity Cache: what leaves the L2s ('traffic': re-reads and register "
                        "spills, 'hbm_frac' of the HBM peak if all of it reached HBM) is not the limit, so the fp32 vector rate is the "
                        "bound reported; 'units' holds the hardware counters of the same launch from profiles/ (occupancy of "
                        "the vector pipe, busy fractions of the L1's address and data units)"}
    if tr:
        roofline["units"] = {k: tr.get(k) for k in ("l1_hit_rate", "l2_hit_rate", "valu_lane_utilisation", "valu_fast_share",
                                                    "valu_pipe_occupancy_lower", "valu_pipe_occupancy_upper", "ta_busy", "td_busy",
                                                    "l1_latency_cycles", "wait_fraction") if tr.get(k) is not None}
    if traffic is None:
        roofline["note"] = ("no hardware-counter pass is committed for this configuration (profiles/traffic.json): only the fp32 "
                            "vector-rate fraction is known")
    # Which fraction is reported: the memory one only for a scene that does not fit the eight 4 MB L2s together - for a
    # smaller scene what leaves the L2s is re-reads that the 256 MB Infinity Cache serves and register spills, not HBM
    # work, and the vector pipe is the unit that is full ('units').  Both fractions are in the line either way.
    scene_bytes = info["node_bytes"] + info["prim_bytes"] + info["shading_bytes"]
    roofline["scene_bytes"] = scene_bytes
    if hbm_frac is not None and hbm_frac > valu_frac and scene_bytes > 8 * (4 << 20):
        roofline.update({"bound": "hbm", "achieved": traffic / kernel_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm_frac,
                         "note": "the scene (%.0f MB of nodes, primitive and shading records) does not fit the 4 MB L2 of an XCD: the "
                                 "measured bytes that leave the L2 per launch (FETCH_SIZE + WRITE_SIZE, profiles/traffic.json: an upper "
                                 "bound on HBM bytes, the 256 MB Infinity Cache serves part of them) over the launch time are the larger "
                                 "fraction; 'valu_frac' is the fp32 vector-rate fraction of the same launch"
                                 % (scene_bytes / 1e6),
                         "valu_frac": valu_frac})
    out = {
        "value": queries / elapsed / 1e6, "unit": "Mrays/s", "ms_per_step": elapsed / args.steps * 1e3,
        "config": {"workload": f"{name}: {WORKLOADS[name]}; {width}x{height}, {frame_spp} spp per frame"
                               + (f" ({mine} on this GPU)" if world > 1 and split == "spp" else "") + f", {bounces} bounces",
                   "triangles": int(sd.n_triangles), "spheres": int(sd.n_spheres), "bvh_nodes": info["n_nodes"], "bvh_depth": info["max_depth"],
                   "traversal_kind": info["traversal_kind"],
                   "bvh_builder": "device" if info["builder"] else "host",
                   "scene_upload_ms": round(info["upload_ms"], 2), "device_build_ms": round(info["device_build_ms"], 3),
                   "split": (f"{split} split, {merge.describe()}" if world > 1 else "none"),
                   "launches_per_frame": st["launches"] // max(1, args.steps)},
        "ms_per_frame": elapsed / args.steps * 1e3,
        "segments_per_s": segments / elapsed, "paths_per_s": camera_rays / elapsed,
        "rays_per_segment": queries / max(1.0, segments),
        "roofline": roofline,
    }
    if world > 1:
        out["scaling"] = scaling
        out["merge"] = {"bytes_per_rank": merge.bytes_per_rank(), "ms_per_frame": merge_s[0] / args.steps * 1e3,
                        "collective": "gather" if split == "tile" else "reduce",
                        "note": "inside the timed region; measured between a barrier after the slowest rank's render and the end of the collective"}
    if cpu:
        out["cpu_baseline"] = cpu
    return out


def main(args):
    import torch
    import torch.distributed as dist
    from pathtracer_amd import native

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N > 1 path on a box with one GPU: every rank on GPU 0, collectives through gloo (RCCL refuses two
    # ranks on one device).  Not a measurement.
    rehearsal = os.environ.get("BENCH_REHEARSAL_ONE_GPU") == "1"
    if rehearsal:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the render path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        print(f"[bench] rank {rank}/{world} on {torch.cuda.get_device_name(local_rank)} (cuda:{local_rank})", file=sys.stderr, flush=True)

    # (workload, split, scaling) per timed configuration
    if args.workload != "all":
        split = args.split or "spp"
        runs = [(args.workload, split, args.scaling or ("weak" if split == "spp" else "strong"))]
    elif world > 1:
        # the multi-GPU configurations the metric names, each one fixed frame shared by the GPUs
        args.sizes["bmw"] = (args.width or 3840, args.height or 2160, args.spp or 1024, args.sizes["bmw"][3])
        args.sizes["glass"] = (args.sizes["glass"][0], args.sizes["glass"][1], args.spp or 8192, args.sizes["glass"][3])
        runs = [("bmw", "tile", "strong"), ("glass", "spp", "strong"), ("envmap", "spp", "strong")]
    else:
        runs = [("bunny", "spp", "weak"), ("cornell", "spp", "weak"), ("bmw", "spp", "weak")]
    names = [r[0] for r in runs]
    if world > 1:
        # every rank builds the same tree on the host: share the cores instead of 16 builder threads per rank
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 16
        os.environ.setdefault("PTK_BUILD_THREADS", str(max(2, min(16, cores // world))))
    ctx = native.Context(local_rank)
    results = {}
    for name, split, scaling in runs:
        results[name] = run_workload(name, args, ctx, torch, dist, dev, rank, world, with_cpu=(world == 1 and not args.no_cpu),
                                     split=split, scaling=scaling)
    if world == 1 and args.workload == "all" and not args.no_4k:
        # the frame that the N > 1 default shares among the GPUs (bmw at 3840x2160), on this one GPU and with few steps: the
        # single-GPU figure that its top-level line compares with
        import copy
        a4 = copy.copy(args)
        a4.sizes = dict(args.sizes)
        a4.sizes["bmw"] = (3840, 2160, args.sizes["bmw"][2], args.sizes["bmw"][3])
        a4.steps, a4.warmup = min(args.steps, 3), min(args.warmup, 1)
        results["bmw_4k"] = run_workload("bmw", a4, ctx, torch, dist, dev, rank, world, with_cpu=False, split="spp", scaling="weak")
        results["bmw_4k"]["steps"] = a4.steps
        names.append("bmw_4k")
    if rank == 0:
        head = results[names[0]]
        out = {"metric": "Mrays/s", "value": head["value"], "unit": "Mrays/s",
               "n_gpus": dist.get_world_size() if world > 1 else 1, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": head["ms_per_step"], "higher_is_better": True,
               "scaling": runs[0][2], "vs_baseline": None, "dtype": "f32", "data": "synthetic"}
        out.update({k: v for k, v in head.items() if k not in ("value", "unit", "ms_per_step")})
        if len(names) > 1:
            out["extra"] = {n: results[n] for n in names[1:]}
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def launch_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start N ranks as children of a fresh launcher
    process (nothing in this process has touched a GPU), pass their output through, return their exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="all", choices=["all"] + sorted(WORKLOADS),
                    help="all = bunny (the top-level numbers) + cornell + bmw under 'extra'")
    ap.add_argument("--width", type=int, default=None, help="default 1024 (workloads envmap and glass have their own)")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel per GPU, default 1024")
    ap.add_argument("--bounces", type=int, default=None, help="default 3")
    ap.add_argument("--batch", type=int, default=0, help="samples per launch (0 = the whole frame in one launch)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--stats-spp", type=int, default=4, help="samples per pixel of the instrumented (test-counting) pass")
    ap.add_argument("--split", default=None, choices=["spp", "tile"],
                    help="how N > 1 GPUs share a frame of a named workload: interleaved sample indices (default) or interleaved row bands")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="named workload, N > 1, sample split: weak = every GPU renders --spp samples (default), strong = they share --spp")
    ap.add_argument("--builder", default=None, choices=["host", "device"],
                    help="tree builder (default: host surface-area-heuristic build; device: linear BVH built on the GPU)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-4k", action="store_true", help="skip the 3840x2160 frame of the 1.3 M-triangle scene that the default run adds under extra")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU time budget of each workload's baseline leg")
    args = ap.parse_args(argv)
    args.sizes = {}
    for name in WORKLOADS:
        w, h, s, b = WORKLOAD_DEFAULTS.get(name, (1024, 1024, 1024, 3))
        args.sizes[name] = (args.width or w, args.height or h, args.spp or s, args.bounces if args.bounces is not None else b)
    return args


if __name__ == "__main__":
    ARGS = parse_args()
    if ARGS.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(ARGS))
    main(ARGS)
