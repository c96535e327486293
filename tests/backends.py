"""Two ways to run the kernels' logic in tests: the real HIP kernels through the C ABI ("gpu"), and the same
device functions compiled for the CPU ("host", tests/hostcheck).  Both are compared against the oracle."""
from __future__ import annotations

import functools

import numpy as np
import pytest

from pathtracer_amd import native, scenes

BACKENDS = [pytest.param("host", id="hostcheck"), pytest.param("gpu", marks=pytest.mark.gpu, id="gpu")]


class GpuBackend:
    def __init__(self, ctx, sd, builder=None):
        self.ctx = ctx
        self.sd = sd
        ctx.upload_scene(sd, builder)
        ctx.set_camera(sd.camera)
        ctx.set_flags(native.PtkFlags.defaults())
        ctx.bind_accum(None)

    def info(self):
        i = self.ctx.scene_info()
        return dict(n_nodes=i["n_nodes"], max_depth=i["max_depth"], max_leaf=i["max_leaf_size"],
                    n_lights=i["n_lights"], n_prims=i["n_primitives"], builder=i["builder"], tree_from_cache=i["tree_from_cache"],
                    upload_ms=i["upload_ms"], device_build_ms=i["device_build_ms"], form=i["traversal_kind"])

    def intersect(self, rays):
        self.ctx.reset_stats()
        hits = self.ctx.intersect(rays)
        s = self.ctx.stats()
        return hits, dict(box_tests=s["box_tests"], prim_tests=s["prim_tests"])

    def render(self, samples, max_bounce, seed=1, sample_first=0, sample_stride=1, row_begin=0, row_end=0, flags=None,
               reset=True, collect_stats=True, band_rows=0, band_first=0, band_stride=1):
        self.ctx.set_flags(flags if flags is not None else native.PtkFlags.defaults())
        if reset:
            self.ctx.reset_accum()
        self.ctx.reset_stats()
        self.ctx.render(samples, max_bounce, seed, sample_first, sample_stride, row_begin, row_end, collect_stats,
                        band_rows, band_first, band_stride)
        img, total = self.ctx.read_accum()
        s = self.ctx.stats()
        return img.astype(np.float64), {k: s[k] for k in ("camera_rays", "segments", "queries", "box_tests", "prim_tests")}


class HostBackend:
    def __init__(self, sd):
        from tests.hostcheck import hc
        self.k = hc.HostCheck(sd)
        self.sd = sd
        self.accum = None

    def info(self):
        return self.k.info()

    def intersect(self, rays):
        return self.k.intersect(rays)

    def render(self, samples, max_bounce, seed=1, sample_first=0, sample_stride=1, row_begin=0, row_end=0, flags=None,
               reset=True, collect_stats=True, band_rows=0, band_first=0, band_stride=1):
        if reset or self.accum is None:
            self.accum = None
        self.accum, cnt = self.k.render(samples, max_bounce, seed, sample_first, sample_stride, row_begin, row_end,
                                        flags, self.accum, band_rows, band_first, band_stride)
        w = self.accum[..., 3:4]
        img = np.where(w > 0, self.accum[..., :3] / np.maximum(w, 1e-30), 0.0)
        return img.astype(np.float64), cnt


def make_backend(kind, sd, request):
    if kind == "gpu":
        return GpuBackend(request.getfixturevalue("gpu_ctx"), sd)
    return HostBackend(sd)


# ---- scenes shared by the parity tests (small enough for the oracle to finish in seconds) ----

@functools.lru_cache(maxsize=None)
def scene(name: str):
    if name == "cornell":
        return scenes.cornell_box(64, 64)
    if name == "cornell_pinhole":
        return scenes.cornell_box(64, 64, aperture=0.0, focus_distance=0.0)
    if name == "cornell_plain_floor":
        return scenes.cornell_box(64, 64, textured_floor=False)
    if name == "mesh":  # 5 120-triangle mesh + box + both spheres
        s = scenes.cornell_with_mesh(4, 64, 64)
        scenes.add_json_spheres(s)
        return s
    if name == "glass":  # dielectric mesh + glass sphere + mirror sphere
        return scenes.glass_scene(3, 48, 48)
    if name == "envmap":  # textured floor, environment map, depth of field, mesh
        return scenes.cornell_with_mesh(3, 64, 36, envmap=scenes.procedural_envmap(512, 256))
    if name == "tiled":  # texture coordinates outside [0,1]: invalid taps and negative blend fractions
        s = scenes.cornell_box(48, 48, aperture=0.0, focus_distance=0.0)
        s.tri_uv = s.tri_uv * 3.0 - 1.0
        return s
    if name == "sky":  # no background colour -> sky gradient; fuzzy specular + microfacet + sphere light
        s = scenes.cornell_box(48, 48, aperture=0.0, focus_distance=0.0)
        s.background_color = (-1.0, -1.0, -1.0)
        fuzzy = s.add_material(scenes.Material(type=scenes.SPECULAR, albedo=(0.9, 0.6, 0.3), roughness=0.3))
        s.add_sphere((-2.5, 1.5, 0.0), 1.5, fuzzy)
        micro = s.add_material(scenes.Material(type=scenes.MICROFACET, albedo=(0.8, 0.8, 0.9), roughness=0.4, metallic=0.8))
        s.add_sphere((2.5, 1.5, 1.0), 1.5, micro)
        lamp = s.add_material(scenes.Material(albedo=(0.8, 0.8, 0.8), emission=(4.0, 3.0, 2.0)))
        s.add_sphere((0.0, 6.0, -2.0), 0.7, lamp)
        return s
    if name == "mirror_tex":  # textured sphere (Sphere::texcoord), MIRROR-type sphere and wall, textured floor
        s = scenes.cornell_box(48, 48, aperture=0.0, focus_distance=0.0)
        tex = s.add_texture(scenes.procedural_texture(256, 6, seed=3))
        globe = s.add_material(scenes.Material(albedo=(1.0, 1.0, 1.0), texture=tex))
        s.add_sphere((2.2, 1.8, 1.0), 1.8, globe)
        # Material::MIRROR is perfectly specular (src/material.h:30-35) but BxDF::sample / eval have no case for
        # it (src/material.cpp:110-136): it scatters like a diffuse surface and only switches light sampling off
        mirror = s.add_material(scenes.Material(type=scenes.MIRROR, albedo=(0.9, 0.9, 0.9), roughness=0.0))
        s.add_sphere((-2.4, 1.5, -1.0), 1.5, mirror)
        s.tri_mat = s.tri_mat.copy()
        s.tri_mat[6:8] = mirror  # the right wall
        return s
    if name == "envmap_full":  # environment map at the size of the reference's asset (3584 x 1792, 19 MB of texels)
        return scenes.cornell_with_mesh(3, 64, 36, envmap=scenes.procedural_envmap(3584, 1792))
    raise KeyError(name)
