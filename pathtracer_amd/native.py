"""ctypes binding of the C ABI in include/ptk.h (libptk.so).

There is no CPU fallback: if the HIP library is missing or no GPU is present,
loading / creating a context raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PTK_LIB selects another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("PTK_LIB") or os.path.join(_HERE, "libptk.so")


class PtkError(RuntimeError):
    pass


class PtkMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("texture", C.c_int32), ("albedo", C.c_double * 3), ("emission", C.c_double * 3),
                ("refraction_index", C.c_double), ("roughness", C.c_double), ("metallic", C.c_double)]


class PtkTexture(C.Structure):
    _fields_ = [("rgb", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


class PtkSceneDesc(C.Structure):
    _fields_ = [("n_triangles", C.c_int32), ("tri_pos", C.c_void_p), ("tri_nrm", C.c_void_p), ("tri_uv", C.c_void_p),
                ("tri_material", C.c_void_p), ("n_spheres", C.c_int32), ("sphere_center", C.c_void_p),
                ("sphere_radius", C.c_void_p), ("sphere_material", C.c_void_p), ("n_materials", C.c_int32),
                ("materials", C.POINTER(PtkMaterial)), ("n_textures", C.c_int32), ("textures", C.POINTER(PtkTexture)),
                ("background_texture", C.c_int32), ("background_color", C.c_double * 3)]


class PtkCamera(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("position", C.c_double * 3), ("forward", C.c_double * 3),
                ("right", C.c_double * 3), ("up", C.c_double * 3), ("fov", C.c_double), ("aperture", C.c_double),
                ("focus_distance", C.c_double)]


class PtkFlags(C.Structure):
    _fields_ = [("debug_normal", C.c_int32), ("russian_roulette", C.c_int32), ("direct_light_sampling", C.c_int32),
                ("indirect_light_sampling", C.c_int32), ("importance_sample", C.c_int32),
                ("check_hemisphere", C.c_int32)]

    @classmethod
    def defaults(cls):  # src/config.h:3-8
        return cls(0, 1, 1, 1, 1, 1)


class PtkRenderParams(C.Structure):
    _fields_ = [("samples", C.c_int32), ("max_bounce", C.c_int32), ("seed", C.c_uint32), ("sample_first", C.c_int32),
                ("sample_stride", C.c_int32), ("row_begin", C.c_int32), ("row_end", C.c_int32),
                ("collect_stats", C.c_int32), ("band_rows", C.c_int32), ("band_first", C.c_int32),
                ("band_stride", C.c_int32)]


class PtkStats(C.Structure):
    _fields_ = [("camera_rays", C.c_uint64), ("segments", C.c_uint64), ("queries", C.c_uint64),
                ("box_tests", C.c_uint64), ("prim_tests", C.c_uint64), ("kernel_ms_last", C.c_double),
                ("kernel_ms_total", C.c_double), ("launches", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class PtkSceneInfo(C.Structure):
    _fields_ = [("n_primitives", C.c_int32), ("n_lights", C.c_int32), ("n_nodes", C.c_int32), ("max_depth", C.c_int32),
                ("max_leaf_size", C.c_int32), ("traversal_kind", C.c_int32), ("node_bytes", C.c_uint64),
                ("prim_bytes", C.c_uint64), ("shading_bytes", C.c_uint64), ("texture_bytes", C.c_uint64),
                ("builder", C.c_int32), ("tree_from_cache", C.c_int32), ("upload_ms", C.c_double),
                ("device_build_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


BUILDERS = {"host": 0, "device": 1}  # PTK_BUILDER_HOST_SAH, PTK_BUILDER_DEVICE_LBVH

RAY_DTYPE = np.dtype([("origin", np.float64, 3), ("direction", np.float64, 3)])
HIT_DTYPE = np.dtype([("hit", np.int32), ("id", np.uint32), ("t", np.float32), ("point", np.float32, 3),
                      ("normal", np.float32, 3), ("uv", np.float32, 2), ("material", np.int32), ("inside", np.int32)])
assert HIT_DTYPE.itemsize == 52

# every symbol include/ptk.h declares
EXPORTS = ["ptk_create", "ptk_destroy", "ptk_last_error", "ptk_upload_scene", "ptk_upload_scene_ex", "ptk_scene_info_get", "ptk_set_camera",
           "ptk_set_flags", "ptk_bind_accum", "ptk_reset_accum", "ptk_render", "ptk_render_static", "ptk_read_accum", "ptk_read_display_u8", "ptk_synchronize",
           "ptk_intersect_batch", "ptk_intersect_batch_device", "ptk_get_stats", "ptk_reset_stats", "ptk_stream",
           "ptk_group_create", "ptk_group_destroy", "ptk_group_last_error", "ptk_group_size", "ptk_group_context",
           "ptk_group_upload_scene", "ptk_group_set_camera", "ptk_group_set_flags", "ptk_group_reset_accum", "ptk_group_render",
           "ptk_group_synchronize", "ptk_group_get_stats", "ptk_group_reset_stats", "ptk_group_read_accum", "ptk_group_read_display_u8"]
SPLITS = {"spp": 0, "tile": 1}  # PTK_SPLIT_SAMPLES, PTK_SPLIT_BANDS


def make_scene_desc(sd):
    """ptk_scene_desc for a scenes.SceneData-like object.  Returns (desc, keepalive)."""
    keep = []

    def arr(a, dt):
        a = np.ascontiguousarray(a, dt)
        keep.append(a)
        return a.ctypes.data if a.size else None

    d = PtkSceneDesc()
    d.n_triangles = sd.n_triangles
    d.tri_pos = arr(sd.tri_pos, np.float64)
    d.tri_nrm = arr(sd.tri_nrm, np.float64)
    d.tri_uv = arr(sd.tri_uv, np.float64)
    d.tri_material = arr(sd.tri_mat, np.int32)
    d.n_spheres = sd.n_spheres
    d.sphere_center = arr(sd.sphere_center, np.float64)
    d.sphere_radius = arr(sd.sphere_radius, np.float64)
    d.sphere_material = arr(sd.sphere_mat, np.int32)
    mats = (PtkMaterial * max(1, len(sd.materials)))()
    for i, m in enumerate(sd.materials):
        mats[i].type = int(m.type)
        mats[i].texture = int(m.texture)
        mats[i].albedo = (C.c_double * 3)(*[float(x) for x in m.albedo])
        mats[i].emission = (C.c_double * 3)(*[float(x) for x in m.emission])
        mats[i].refraction_index = float(m.refraction_index)
        mats[i].roughness = float(m.roughness)
        mats[i].metallic = float(m.metallic)
    keep.append(mats)
    d.n_materials = len(sd.materials)
    d.materials = mats
    texs = (PtkTexture * max(1, len(sd.textures)))()
    for i, t in enumerate(sd.textures):
        t = np.ascontiguousarray(t, np.uint8)
        keep.append(t)
        texs[i].rgb = t.ctypes.data
        texs[i].width = t.shape[1]
        texs[i].height = t.shape[0]
    keep.append(texs)
    d.n_textures = len(sd.textures)
    d.textures = texs
    d.background_texture = int(sd.background_texture)
    d.background_color = (C.c_double * 3)(*[float(x) for x in sd.background_color])
    return d, keep


def make_camera(cd) -> PtkCamera:
    """Camera::Camera + Camera::look_at + Camera::compute, src/camera.cpp:6-17,78-89 (double precision)."""
    pos = np.asarray(cd.position, np.float64)
    tgt = np.asarray(cd.target, np.float64)
    fwd = tgt - pos
    fwd = fwd * (1.0 / np.sqrt(np.dot(fwd, fwd)))
    world_up = np.array([0.0, 1.0, 0.0])
    right = np.cross(fwd, world_up)
    up = np.cross(right, fwd)
    c = PtkCamera()
    c.width, c.height = int(cd.width), int(cd.height)
    c.position = (C.c_double * 3)(*pos)
    c.forward = (C.c_double * 3)(*fwd)
    c.right = (C.c_double * 3)(*right)
    c.up = (C.c_double * 3)(*up)
    c.fov = float(cd.fov) * 0.017453292519943295  # glm::radians
    c.aperture = float(cd.aperture)
    c.focus_distance = float(cd.focus_distance)
    return c


_lib = None


def load_library():
    """Load libptk.so; raises PtkError when it has not been built (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PtkError(f"{LIB_PATH} is missing: build it with `make -C pathtracer_amd/csrc` "
                       "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.ptk_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.ptk_destroy.argtypes = [vp]
    L.ptk_last_error.argtypes = [vp]
    L.ptk_last_error.restype = C.c_char_p
    L.ptk_upload_scene.argtypes = [vp, C.POINTER(PtkSceneDesc)]
    L.ptk_upload_scene_ex.argtypes = [vp, C.POINTER(PtkSceneDesc), C.c_int32]
    L.ptk_scene_info_get.argtypes = [vp, C.POINTER(PtkSceneInfo)]
    L.ptk_set_camera.argtypes = [vp, C.POINTER(PtkCamera)]
    L.ptk_set_flags.argtypes = [vp, C.POINTER(PtkFlags)]
    L.ptk_bind_accum.argtypes = [vp, vp]
    L.ptk_reset_accum.argtypes = [vp]
    L.ptk_render.argtypes = [vp, C.POINTER(PtkRenderParams)]
    if os.environ.get("PTK_LIB") and not hasattr(L, "ptk_group_create"):
        _lib = L  # an older build of the library selected for an A/B run: only the entry points it has
        return L
    L.ptk_render_static.argtypes = [vp, C.POINTER(PtkRenderParams)]
    L.ptk_read_accum.argtypes = [vp, vp, C.POINTER(C.c_int32)]
    L.ptk_read_display_u8.argtypes = [vp, vp, C.POINTER(C.c_int32)]
    L.ptk_synchronize.argtypes = [vp]
    L.ptk_intersect_batch.argtypes = [vp, vp, C.c_int32, vp]
    L.ptk_intersect_batch_device.argtypes = [vp, vp, C.c_int32, vp]
    L.ptk_get_stats.argtypes = [vp, C.POINTER(PtkStats)]
    L.ptk_reset_stats.argtypes = [vp]
    L.ptk_stream.argtypes = [vp]
    L.ptk_stream.restype = vp
    L.ptk_group_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.ptk_group_destroy.argtypes = [vp]
    L.ptk_group_last_error.argtypes = [vp]
    L.ptk_group_last_error.restype = C.c_char_p
    L.ptk_group_size.argtypes = [vp]
    L.ptk_group_context.argtypes = [vp, C.c_int]
    L.ptk_group_context.restype = vp
    L.ptk_group_upload_scene.argtypes = [vp, C.POINTER(PtkSceneDesc), C.c_int32]
    L.ptk_group_set_camera.argtypes = [vp, C.POINTER(PtkCamera)]
    L.ptk_group_set_flags.argtypes = [vp, C.POINTER(PtkFlags)]
    L.ptk_group_reset_accum.argtypes = [vp]
    L.ptk_group_render.argtypes = [vp, C.POINTER(PtkRenderParams), C.c_int32]
    L.ptk_group_synchronize.argtypes = [vp]
    L.ptk_group_get_stats.argtypes = [vp, C.POINTER(PtkStats)]
    L.ptk_group_reset_stats.argtypes = [vp]
    L.ptk_group_read_accum.argtypes = [vp, vp, C.POINTER(C.c_int32)]
    L.ptk_group_read_display_u8.argtypes = [vp, vp, C.POINTER(C.c_int32)]
    _lib = L
    return L


class Context:
    """One ptk_ctx (one GPU)."""

    def __init__(self, device: int = 0):
        self._L = load_library()
        h = C.c_void_p()
        rc = self._L.ptk_create(int(device), C.byref(h))
        if rc != 0:
            raise PtkError(f"ptk_create failed ({rc}): {self._L.ptk_last_error(None).decode()}")
        self.h = h
        self.width = self.height = 0

    def _check(self, rc, what):
        if rc != 0:
            raise PtkError(f"{what} failed ({rc}): {self._L.ptk_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self._L.ptk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_scene(self, sd, builder=None):
        """builder: None (ptk_upload_scene: host tree unless PTK_BUILDER says otherwise), "host" or "device"."""
        desc, keep = make_scene_desc(sd)
        if builder is None:
            self._check(self._L.ptk_upload_scene(self.h, C.byref(desc)), "ptk_upload_scene")
        else:
            self._check(self._L.ptk_upload_scene_ex(self.h, C.byref(desc), BUILDERS[builder]), "ptk_upload_scene_ex")
        del keep

    def scene_info(self):
        info = PtkSceneInfo()
        self._check(self._L.ptk_scene_info_get(self.h, C.byref(info)), "ptk_scene_info_get")
        return info.as_dict()

    def set_camera(self, cd):
        cam = cd if isinstance(cd, PtkCamera) else make_camera(cd)
        self._check(self._L.ptk_set_camera(self.h, C.byref(cam)), "ptk_set_camera")
        self.width, self.height = cam.width, cam.height

    def set_flags(self, flags: PtkFlags):
        self._check(self._L.ptk_set_flags(self.h, C.byref(flags)), "ptk_set_flags")

    def bind_accum(self, device_ptr):
        self._check(self._L.ptk_bind_accum(self.h, C.c_void_p(device_ptr) if device_ptr else None), "ptk_bind_accum")

    def reset_accum(self):
        self._check(self._L.ptk_reset_accum(self.h), "ptk_reset_accum")

    def render(self, samples, max_bounce, seed=1, sample_first=0, sample_stride=1, row_begin=0, row_end=0,
               collect_stats=False, band_rows=0, band_first=0, band_stride=1, static=False):
        """static=True: ptk_render_static, the variant that allocates no device memory."""
        p = PtkRenderParams(int(samples), int(max_bounce), int(seed), int(sample_first), int(sample_stride),
                            int(row_begin), int(row_end), 1 if collect_stats else 0, int(band_rows), int(band_first),
                            int(band_stride))
        if static:
            self._check(self._L.ptk_render_static(self.h, C.byref(p)), "ptk_render_static")
        else:
            self._check(self._L.ptk_render(self.h, C.byref(p)), "ptk_render")

    def synchronize(self):
        self._check(self._L.ptk_synchronize(self.h), "ptk_synchronize")

    def read_accum(self):
        out = np.zeros((self.height, self.width, 3), np.float32)
        total = C.c_int32(0)
        self._check(self._L.ptk_read_accum(self.h, out.ctypes.data, C.byref(total)), "ptk_read_accum")
        return out, int(total.value)

    def read_display_u8(self):
        """Renderer::save_image's tone-mapped 8-bit image (src/renderer.cpp:180-192), computed on the device."""
        out = np.zeros((self.height, self.width, 3), np.uint8)
        total = C.c_int32(0)
        self._check(self._L.ptk_read_display_u8(self.h, out.ctypes.data, C.byref(total)), "ptk_read_display_u8")
        return out, int(total.value)

    def intersect(self, rays):
        r = np.zeros(len(rays), RAY_DTYPE)
        rays = np.asarray(rays, np.float64).reshape(-1, 6)
        r["origin"] = rays[:, 0:3]
        r["direction"] = rays[:, 3:6]
        hits = np.zeros(len(r), HIT_DTYPE)
        self._check(self._L.ptk_intersect_batch(self.h, r.ctypes.data, len(r), hits.ctypes.data), "ptk_intersect_batch")
        return hits

    def intersect_device(self, rays_ptr, n, hits_ptr):
        self._check(self._L.ptk_intersect_batch_device(self.h, C.c_void_p(rays_ptr), int(n), C.c_void_p(hits_ptr)),
                    "ptk_intersect_batch_device")

    def stats(self):
        s = PtkStats()
        self._check(self._L.ptk_get_stats(self.h, C.byref(s)), "ptk_get_stats")
        return s.as_dict()

    def reset_stats(self):
        self._check(self._L.ptk_reset_stats(self.h), "ptk_reset_stats")


class Group:
    """One ptk_group: several GPUs behind one handle (one host thread, one RCCL reduce per frame read)."""

    def __init__(self, devices):
        self._L = load_library()
        devices = [int(d) for d in devices]
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = self._L.ptk_group_create(arr, len(devices), C.byref(h))
        if rc != 0:
            raise PtkError(f"ptk_group_create failed ({rc}): {self._L.ptk_group_last_error(None).decode()}")
        self.h = h
        self.devices = devices
        self.width = self.height = 0

    def _check(self, rc, what):
        if rc != 0:
            raise PtkError(f"{what} failed ({rc}): {self._L.ptk_group_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self._L.ptk_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return int(self._L.ptk_group_size(self.h))

    def context_stats(self, index):
        """ptk_get_stats of the index-th device's context."""
        s = PtkStats()
        c = C.c_void_p(self._L.ptk_group_context(self.h, int(index)))
        rc = self._L.ptk_get_stats(c, C.byref(s))
        if rc != 0:
            raise PtkError(f"ptk_get_stats failed ({rc})")
        return s.as_dict()

    def stats(self):
        """ptk_group_get_stats: the counters of all devices added up."""
        s = PtkStats()
        self._check(self._L.ptk_group_get_stats(self.h, C.byref(s)), "ptk_group_get_stats")
        return s.as_dict()

    def reset_stats(self):
        self._check(self._L.ptk_group_reset_stats(self.h), "ptk_group_reset_stats")

    def upload_scene(self, sd, builder="host"):
        desc, keep = make_scene_desc(sd)
        self._check(self._L.ptk_group_upload_scene(self.h, C.byref(desc), BUILDERS[builder]), "ptk_group_upload_scene")
        del keep

    def set_camera(self, cd):
        cam = cd if isinstance(cd, PtkCamera) else make_camera(cd)
        self._check(self._L.ptk_group_set_camera(self.h, C.byref(cam)), "ptk_group_set_camera")
        self.width, self.height = cam.width, cam.height

    def set_flags(self, flags: PtkFlags):
        self._check(self._L.ptk_group_set_flags(self.h, C.byref(flags)), "ptk_group_set_flags")

    def reset_accum(self):
        self._check(self._L.ptk_group_reset_accum(self.h), "ptk_group_reset_accum")

    def render(self, samples, max_bounce, seed=1, sample_first=0, sample_stride=1, split="spp"):
        p = PtkRenderParams(int(samples), int(max_bounce), int(seed), int(sample_first), int(sample_stride), 0, 0, 0, 0, 0, 1)
        self._check(self._L.ptk_group_render(self.h, C.byref(p), SPLITS[split]), "ptk_group_render")

    def synchronize(self):
        self._check(self._L.ptk_group_synchronize(self.h), "ptk_group_synchronize")

    def read_accum(self):
        out = np.zeros((self.height, self.width, 3), np.float32)
        total = C.c_int32(0)
        self._check(self._L.ptk_group_read_accum(self.h, out.ctypes.data, C.byref(total)), "ptk_group_read_accum")
        return out, int(total.value)

    def read_display_u8(self):
        out = np.zeros((self.height, self.width, 3), np.uint8)
        total = C.c_int32(0)
        self._check(self._L.ptk_group_read_display_u8(self.h, out.ctypes.data, C.byref(total)), "ptk_group_read_display_u8")
        return out, int(total.value)
