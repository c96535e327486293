#!/usr/bin/env python3
"""Throughput of the ray-batch entry point (ptk_intersect_batch_device) on incoherent rays:
origins uniform in the Cornell box, directions uniform on the sphere.  Prints Mrays/s."""
import argparse, os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pathtracer_amd import native, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--level", type=int, default=6)
ap.add_argument("--rays", type=int, default=1 << 24)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--coherent", action="store_true", help="camera rays through a pixel grid, row-major, instead of random rays")
a = ap.parse_args()
dev = torch.device("cuda", 0)
sd = scenes.cornell_with_mesh(a.level, 64, 64)
ctx = native.Context(0)
ctx.upload_scene(sd)
g = torch.Generator(device=dev); g.manual_seed(1)
n = a.rays
o = torch.rand((n, 3), device=dev, generator=g) * torch.tensor([9.0, 9.0, 9.0], device=dev) + torch.tensor([-4.5, 0.5, -4.5], device=dev)
z = 1 - 2 * torch.rand(n, device=dev, generator=g); phi = 6.2831853 * torch.rand(n, device=dev, generator=g)
s = torch.sqrt(torch.clamp(1 - z * z, min=0))
d = torch.stack([s * torch.cos(phi), s * torch.sin(phi), z], 1)
if a.coherent:
    side = int(n ** 0.5)
    n = side * side
    ys, xs = torch.meshgrid(torch.arange(side, device=dev), torch.arange(side, device=dev), indexing="ij")
    u = (xs.flatten().float() + 0.5) / side * 2 - 1
    v = (ys.flatten().float() + 0.5) / side * 2 - 1
    d = torch.stack([u * 0.25, -v * 0.25, -torch.ones_like(u)], 1)
    d = d / d.norm(dim=1, keepdim=True)
    o = torch.tensor([0.0, 5.0, 24.0], device=dev).expand(n, 3)
rays = torch.cat([o, d], 1).contiguous().float()
hits = torch.zeros((n, 13), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
L = ctx._L
for r in range(a.reps + 1):
    if r == 1:
        ctx.reset_stats()
    rc = L.ptk_intersect_batch_device(ctx.h, C.c_void_p(rays.data_ptr()), n, C.c_void_p(hits.data_ptr()))
    assert rc == 0
ctx.synchronize()
st = ctx.stats()
ms = st["kernel_ms_total"] / a.reps
print(f"{n} {'coherent' if a.coherent else 'incoherent'} rays, level {a.level}: {ms:.2f} ms per batch, {n / ms / 1e3:.1f} Mrays/s, hit fraction {float((hits[:,0]!=0).float().mean()):.3f}")
