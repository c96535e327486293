#!/usr/bin/env python3
"""Scene upload times on the GPU box: host tree build against the on-device build (and the tree cache), for
the 81 920- and 1 310 720-triangle scenes.  Each variant is run twice; the second run has every kernel loaded.

    python tools/bench_build.py [--levels 6 8]
"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import native, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", type=int, nargs="+", default=[6, 8])
    args = ap.parse_args()
    ctx = native.Context(0)
    for level in args.levels:
        sd = scenes.cornell_with_mesh(level, 64, 64)
        desc_t = time.time()
        for builder in ("device", "host"):
            for run in range(2):
                t = time.time()
                ctx.upload_scene(sd, builder)
                wall = (time.time() - t) * 1e3
                i = ctx.scene_info()
                print(f"level {level} ({i['n_primitives']} primitives) {builder:6s} run {run}: call {i['upload_ms']:8.1f} ms "
                      f"(python wall {wall:8.1f} ms), device tree {i['device_build_ms']:6.2f} ms, {i['n_nodes']} nodes, "
                      f"depth {i['max_depth']}", flush=True)
        with tempfile.TemporaryDirectory() as d:
            os.environ["PTK_BVH_CACHE"] = d
            for run in range(2):
                ctx.upload_scene(sd, "host")
                i = ctx.scene_info()
                print(f"level {level} host + cache run {run}: call {i['upload_ms']:8.1f} ms, from cache {i['tree_from_cache']}", flush=True)
            del os.environ["PTK_BVH_CACHE"]
        del desc_t
    ctx.close()


if __name__ == "__main__":
    main()
