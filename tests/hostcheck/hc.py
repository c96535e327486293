"""Loader for the CPU-compiled copy of the kernels' device functions (TEST HARNESS ONLY, see hostcheck.cpp)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from pathtracer_amd import native

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_LIB = os.path.join(_HERE, "libhostcheck.so")
_SRCS = [os.path.join(_HERE, "hostcheck.cpp"), os.path.join(_ROOT, "pathtracer_amd", "csrc", "ptk_build.cpp")]
_DEPS = _SRCS + [os.path.join(_ROOT, "pathtracer_amd", "csrc", h)
                 for h in ("ptk_math.h", "ptk_layout.h", "ptk_traverse.h", "ptk_shade.h", "ptk_build.h")]


def build():
    newest = max(os.path.getmtime(p) for p in _DEPS)
    if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < newest:
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=fast",
                               "-o", _LIB] + _SRCS)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        vp = C.c_void_p
        L.hc_create.argtypes = [C.POINTER(native.PtkSceneDesc)]
        L.hc_create.restype = vp
        L.hc_error.argtypes = [vp]
        L.hc_error.restype = C.c_char_p
        L.hc_destroy.argtypes = [vp]
        L.hc_info.argtypes = [vp, vp]
        L.hc_tree_check.argtypes = [vp, vp]
        L.hc_intersect.argtypes = [vp, vp, C.c_int, vp, vp]
        L.hc_render.argtypes = [vp, C.POINTER(native.PtkCamera), C.POINTER(native.PtkFlags),
                                C.POINTER(native.PtkRenderParams), vp, vp]
        _lib = L
    return _lib


class HostCheck:
    def __init__(self, sd):
        L = lib()
        desc, keep = native.make_scene_desc(sd)
        self.h = L.hc_create(C.byref(desc))
        err = L.hc_error(self.h).decode()
        if err:
            raise RuntimeError(err)
        self.sd = sd
        self.cam = native.make_camera(sd.camera)

    def __del__(self):
        try:
            lib().hc_destroy(self.h)
        except Exception:
            pass

    def info(self):
        out = np.zeros(6, np.int32)
        lib().hc_info(self.h, out.ctypes.data)
        return dict(n_nodes=int(out[0]), max_depth=int(out[1]), max_leaf=int(out[2]), n_lights=int(out[3]),
                    n_prims=int(out[4]), tree_from_cache=int(out[5]))

    def tree_check(self):
        out = np.zeros(3, np.int64)
        lib().hc_tree_check(self.h, out.ctypes.data)
        return dict(violations=int(out[0]), reached=int(out[1]), deepest=int(out[2]))

    def intersect(self, rays):
        rays = np.asarray(rays, np.float64).reshape(-1, 6)
        r = np.zeros(len(rays), native.RAY_DTYPE)
        r["origin"] = rays[:, :3]
        r["direction"] = rays[:, 3:]
        hits = np.zeros(len(r), native.HIT_DTYPE)
        cnt = np.zeros(2, np.uint64)
        high = lib().hc_intersect(self.h, r.ctypes.data, len(r), hits.ctypes.data, cnt.ctypes.data)
        return hits, dict(box_tests=int(cnt[0]), prim_tests=int(cnt[1]), stack_high_water=int(high))

    def render(self, samples, max_bounce, seed=1, sample_first=0, sample_stride=1, row_begin=0, row_end=0,
               flags=None, accum=None, band_rows=0, band_first=0, band_stride=1):
        w, h = self.cam.width, self.cam.height
        if accum is None:
            accum = np.zeros((h, w, 4), np.float32)
        p = native.PtkRenderParams(samples, max_bounce, seed, sample_first, sample_stride, row_begin, row_end, 1,
                                   band_rows, band_first, band_stride)
        fl = flags if flags is not None else native.PtkFlags.defaults()
        cnt = np.zeros(5, np.uint64)
        lib().hc_render(self.h, C.byref(self.cam), C.byref(fl), C.byref(p), accum.ctypes.data, cnt.ctypes.data)
        names = ["camera_rays", "segments", "queries", "box_tests", "prim_tests"]
        return accum, {k: int(v) for k, v in zip(names, cnt)}
