#!/usr/bin/env python3
"""Box and primitive tests per ray of the host builder's tree on a benchmark scene, counted by the host-compiled twin
of the render kernel (tests/hostcheck) - no GPU needed.  Builder experiments are switched by their environment
variables.    python tools/tree_quality.py [bunny|bmw|envmap|glass] [width] [spp]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from hostcheck import hc  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "bunny"
width = int(sys.argv[2]) if len(sys.argv) > 2 else 128
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 4
sd = bench.build_scene(name, width, width)
t = time.time()
h = hc.HostCheck(sd)
build_s = time.time() - t
info = h.info()
_, c = h.render(spp, 3)
print(f"{name}: {info['n_nodes']} nodes, depth {info['max_depth']}, build {build_s:.2f} s; "
      f"box tests per ray {c['box_tests'] / c['queries']:.3f}, primitive tests per ray {c['prim_tests'] / c['queries']:.3f}")
