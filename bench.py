#!/usr/bin/env python3
"""Benchmark of the render path: Mrays/s and ms/frame.

  python bench.py --gpus N --steps K --warmup W

A step is one frame: Renderer::render(spp) over the whole image (src/renderer.cpp:27-52), i.e. one pass of
the path kernel, with the scene already resident in HBM.  A ray is one closest-hit query
(Scene::find_intersection, src/scene.cpp:19): a path segment or a shadow ray that is actually traced; the
count comes from the kernel's own counters.  With N > 1 (one process per GPU, launched by
torch.distributed.run) every GPU renders `spp` samples of the frame with interleaved global sample indices
(weak scaling: the frame gets N x spp samples; --split tile shares one frame's rows instead) and the
accumulation buffers are summed onto rank 0 with one RCCL reduce per frame, inside the timed region.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (description, builder kwargs)
    "bunny": "Cornell box + 81,920-triangle smooth mesh (bunny-class stand-in: diffuse + emissive area light)",
    "cornell": "Cornell box OBJ (12 triangles, textured floor, depth of field)",
    "bmw": "Cornell box + 1,310,720-triangle mesh (BMW-class stand-in)",
    "envmap": "Cornell box (textured floor) + 81,920-triangle mesh, 3584x1792 environment map, depth of field",
    "glass": "Cornell box + dielectric 20,480-triangle mesh + glass and mirror spheres (Russian roulette active)",
}
# per-workload defaults for (width, height, spp, bounces) where they differ from the command line's
WORKLOAD_DEFAULTS = {"envmap": (1920, 1080, 1024, 3), "glass": (500, 500, 2048, 8)}


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


def cpu_baseline(sd, budget_s=15.0):
    """The CPU oracle (a restatement of the reference renderer) timed on this box's host cores on a bounded
    sample of the same workload: a band of rows in the middle of the frame, sized from a one-sample probe so
    that the timed run takes about budget_s seconds."""
    from oracle import orc
    o = orc.OracleScene(sd)
    h = sd.camera.height
    threads = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or min(16, os.cpu_count() or 1)
    band = max(8, min(h, 64))
    y0 = (h - band) // 2
    t = time.perf_counter()
    o.render(1, ARGS.bounces, seed=99, rows=(y0, y0 + band), threads=threads)
    t1 = max(time.perf_counter() - t, 1e-4)  # one sample per pixel of the probe band
    spp = int(max(1, min(ARGS.spp, budget_s / t1)))
    if spp == ARGS.spp:  # the band at full spp is still short: widen it
        band = int(max(band, min(h, band * budget_s / (t1 * spp))))
        y0 = (h - band) // 2
    t = time.perf_counter()
    _, c = o.render(spp, ARGS.bounces, seed=100, rows=(y0, y0 + band), threads=threads)
    dt = time.perf_counter() - t
    return {"value": c["queries"] / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"rows {y0}..{y0 + band} of the frame, {spp} spp, {ARGS.bounces} bounces, "
                      f"{c['queries']} rays in {dt:.2f} s",
            "segments_per_s": c["segments"] / dt, "prim_tests_per_ray": c["prim_tests"] / max(1, c["queries"]),
            "box_tests_per_ray": c["box_tests"] / max(1, c["queries"])}


def main():
    import torch
    import torch.distributed as dist
    from pathtracer_amd import native, parallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != ARGS.gpus:
        if world == 1 and ARGS.gpus > 1:
            raise SystemExit("--gpus N > 1 needs one process per GPU: launch with python -m torch.distributed.run "
                             "--nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the render path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    sd = build_scene(ARGS.workload, ARGS.width, ARGS.height)
    ctx = native.Context(local_rank)
    ctx.upload_scene(sd, ARGS.builder)
    ctx.set_camera(sd.camera)
    info = ctx.scene_info()
    accum = torch.zeros((ARGS.height, ARGS.width, 4), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx.bind_accum(accum.data_ptr())

    plan = parallel.frame_plan(ARGS.split, rank, world, ARGS.spp, ARGS.height, batch=ARGS.batch)

    def frame(step):
        accum.zero_()
        torch.cuda.synchronize()  # the kernel runs on the library's own stream
        for c in plan:
            ctx.render(c.samples, ARGS.bounces, seed=ARGS.seed + step, sample_first=c.sample_first,
                       sample_stride=c.sample_stride, row_begin=c.row_begin, row_end=c.row_end,
                       band_rows=c.band_rows, band_first=c.band_first, band_stride=c.band_stride)
        ctx.synchronize()
        parallel.reduce_frame(accum, world)  # RCCL over xGMI: the frame's one exchange step

    for w in range(ARGS.warmup):
        frame(-1 - w)
    ctx.reset_stats()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(ARGS.steps):
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

    # instrumented pass on the same scene: box / primitive tests per ray for the algorithmic-work figure
    ctx.reset_stats()
    accum.zero_()
    torch.cuda.synchronize()
    ctx.render(max(1, min(ARGS.stats_spp, ARGS.spp)), ARGS.bounces, seed=7, collect_stats=True)
    si = ctx.stats()
    box_per_ray = si["box_tests"] / max(1, si["queries"])
    prim_per_ray = si["prim_tests"] / max(1, si["queries"])
    seg_per_ray = si["segments"] / max(1, si["queries"])

    if rank == 0:
        # algorithmic bytes per ray: 32 B per child box tested (6 fp32 bounds + 2 links), 36 B per primitive
        # tested (3 fp32 positions x 3), 32 B ray in, 16 B hit out
        bytes_per_ray = box_per_ray * 32.0 + prim_per_ray * 36.0 + 48.0
        flops_per_ray = box_per_ray * 21.0 + prim_per_ray * 60.0 + 250.0 * seg_per_ray
        rays_per_launch = st["queries"] / max(1, st["launches"])
        kernel_s = (st["kernel_ms_total"] / max(1, st["launches"])) / 1e3
        achieved_gbs = rays_per_launch * bytes_per_ray / kernel_s / 1e9
        achieved_tf = rays_per_launch * flops_per_ray / kernel_s / 1e12
        # memory-side bytes per launch of this exact configuration, from the hardware-counter passes whose
        # summary is committed under profiles/ (tools/profile.sh + tools/make_traffic.py); null for others
        traffic, l2_hit = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = f"{ARGS.workload}_{ARGS.width}x{ARGS.height}_{ARGS.spp}spp"
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
                l2_hit = tj.get(key, {}).get("l2_hit_rate")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s", "value": queries / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world,
            "steps": ARGS.steps, "warmup": ARGS.warmup, "ms_per_step": elapsed / ARGS.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if ARGS.split == "spp" else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{ARGS.workload}: {WORKLOADS[ARGS.workload]}; {ARGS.width}x{ARGS.height}, "
                                   f"{ARGS.spp} spp per GPU, {ARGS.bounces} bounces",
                       "triangles": int(sd.n_triangles), "spheres": int(sd.n_spheres), "bvh_nodes": info["n_nodes"], "bvh_builder": "device" if info["builder"] else "host",
                       "scene_upload_ms": round(info["upload_ms"], 2), "device_build_ms": round(info["device_build_ms"], 3),
                       "split": (f"{ARGS.split} split, one RCCL reduce per frame" if world > 1 else "none"),
                       "launches_per_frame": st["launches"] // max(1, ARGS.steps)},
            "ms_per_frame": elapsed / ARGS.steps * 1e3,
            "segments_per_s": segments / elapsed, "paths_per_s": camera_rays / elapsed,
            "rays_per_segment": queries / max(1.0, segments),
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved_gbs / 8000.0, "traffic": traffic,
                         "algorithmic_bytes_per_launch": rays_per_launch * bytes_per_ray, "l2_hit_rate": l2_hit,
                         "note": "achieved counts every box and primitive test as bytes fetched; the scene is "
                                 "cache resident (traffic << algorithmic bytes), so frac can exceed 1; the "
                                 "kernel is bound by load latency and L1 accesses per ray, not by HBM: see fp32_valu and profiles/",
                         "kernel": "render_kernel", "kernel_ms_avg": kernel_s * 1e3,
                         "bytes_per_ray": bytes_per_ray, "box_tests_per_ray": box_per_ray,
                         "prim_tests_per_ray": prim_per_ray, "rays_per_launch": rays_per_launch,
                         "fp32_valu": {"achieved": achieved_tf, "peak": 157.3, "unit": "TFLOP/s",
                                       "frac": achieved_tf / 157.3, "flops_per_ray": flops_per_ray}},
        }
        if world == 1 and not ARGS.no_cpu:
            out["cpu_baseline"] = cpu_baseline(sd)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="bunny", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=None, help="default 1024 (workloads envmap and glass have their own)")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel per GPU, default 1024")
    ap.add_argument("--bounces", type=int, default=None, help="default 3")
    ap.add_argument("--batch", type=int, default=0, help="samples per launch (0 = the whole frame in one launch)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--stats-spp", type=int, default=4, help="samples per pixel of the instrumented (test-counting) pass")
    ap.add_argument("--split", default="spp", choices=["spp", "tile"],
                    help="how N > 1 GPUs share a frame: interleaved sample indices or interleaved row bands")
    ap.add_argument("--builder", default=None, choices=["host", "device"],
                    help="tree builder (default: host surface-area-heuristic build; device: linear BVH built on the GPU)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ARGS = ap.parse_args()
    _w, _h, _s, _b = WORKLOAD_DEFAULTS.get(ARGS.workload, (1024, 1024, 1024, 3))
    ARGS.width = ARGS.width or _w
    ARGS.height = ARGS.height or _h
    ARGS.spp = ARGS.spp or _s
    ARGS.bounces = ARGS.bounces if ARGS.bounces is not None else _b
    main()
