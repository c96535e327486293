"""Procedural scene descriptions (host side, numpy only).

The reference takes its geometry from OBJ files and JSON sphere lists
(src/scene.cpp:93-245, src/main.cpp:113-170).  None of its asset files travel
with this package, so the scenes used by tests and benchmarks are generated
here with the same values the reference would end up with after loading:

* vertex data passes through float32 and is widened to double
  (tinyobj real_t = float, src/tiny_obj_loader.h:150; src/scene.cpp:184-201);
* quads are split as [0,1,3],[1,2,3] unless diagonal 0-2 is strictly shorter
  (src/tiny_obj_loader.h:1587-1605);
* primitive ids follow insertion order: models first, then spheres
  (src/scene.cpp:56-62, src/main.cpp:117-145);
* textures are 8-bit RGB with rows stored bottom-up
  (stbi_set_flip_vertically_on_load, src/image.cpp:29).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

# Material::Type, src/material.h:12-18
DIFFUSE, SPECULAR, DIELECTRIC, MIRROR, MICROFACET = 1, 2, 4, 8, 16


@dataclass
class Material:
    """Material, src/material.h:11-36 (same defaults)."""
    type: int = DIFFUSE
    albedo: tuple = (0.8, 0.8, 0.8)
    emission: tuple = (0.0, 0.0, 0.0)
    refraction_index: float = 1.52
    roughness: float = 0.5
    metallic: float = 0.5
    texture: int = -1  # index into SceneData.textures


@dataclass
class CameraDesc:
    """Camera constructor + look_at arguments, src/camera.h:9-15."""
    width: int = 640
    height: int = 480
    fov: float = 45.0          # degrees
    aperture: float = 0.0      # defocus angle in degrees
    focus_distance: float = 0.0
    position: tuple = (0.0, 0.0, 10.0)
    target: tuple = (0.0, 0.0, 0.0)


@dataclass
class SceneData:
    tri_pos: np.ndarray = field(default_factory=lambda: np.zeros((0, 3, 3)))
    tri_nrm: np.ndarray = field(default_factory=lambda: np.zeros((0, 3, 3)))
    tri_uv: np.ndarray = field(default_factory=lambda: np.zeros((0, 3, 2)))
    tri_mat: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int32))
    sphere_center: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    sphere_radius: np.ndarray = field(default_factory=lambda: np.zeros((0,)))
    sphere_mat: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int32))
    materials: List[Material] = field(default_factory=list)
    textures: List[np.ndarray] = field(default_factory=list)   # (H, W, 3) uint8, row 0 = bottom
    background_texture: int = -1
    background_color: tuple = (-1.0, -1.0, -1.0)               # Scene::Scene, src/scene.cpp:15
    camera: CameraDesc = field(default_factory=CameraDesc)

    @property
    def n_triangles(self) -> int:
        return int(self.tri_pos.shape[0])

    @property
    def n_spheres(self) -> int:
        return int(self.sphere_center.shape[0])

    @property
    def n_primitives(self) -> int:
        return self.n_triangles + self.n_spheres

    def add_material(self, m: Material) -> int:
        self.materials.append(m)
        return len(self.materials) - 1

    def add_texture(self, rgb_bottom_up: np.ndarray) -> int:
        t = np.ascontiguousarray(rgb_bottom_up, dtype=np.uint8)
        assert t.ndim == 3 and t.shape[2] == 3
        self.textures.append(t)
        return len(self.textures) - 1

    def add_triangles(self, pos, nrm, uv, mat) -> None:
        pos = np.asarray(pos, np.float64).reshape(-1, 3, 3)
        n = pos.shape[0]
        nrm = np.asarray(nrm, np.float64).reshape(n, 3, 3)
        uv = np.asarray(uv, np.float64).reshape(n, 3, 2)
        mat = np.broadcast_to(np.asarray(mat, np.int32), (n,))
        self.tri_pos = np.concatenate([self.tri_pos, pos])
        self.tri_nrm = np.concatenate([self.tri_nrm, nrm])
        self.tri_uv = np.concatenate([self.tri_uv, uv])
        self.tri_mat = np.concatenate([self.tri_mat, mat]).astype(np.int32)

    def add_sphere(self, center, radius, mat) -> None:
        self.sphere_center = np.concatenate([self.sphere_center, np.asarray(center, np.float64).reshape(1, 3)])
        self.sphere_radius = np.concatenate([self.sphere_radius, [float(radius)]])
        self.sphere_mat = np.concatenate([self.sphere_mat, [int(mat)]]).astype(np.int32)

    def light_ids(self) -> np.ndarray:
        """Primitive ids with any emission component > 0, in id order (src/geometry.cpp:225, src/scene.cpp:60)."""
        em = np.array([max(m.emission) > 0.0 for m in self.materials], bool)
        mats = np.concatenate([self.tri_mat, self.sphere_mat])
        return np.nonzero(em[mats])[0].astype(np.uint32)


def _f32(a):
    """Round through float32 like the OBJ parser does, widen back to double."""
    return np.asarray(a, np.float64).astype(np.float32).astype(np.float64)


def _quad(scene: SceneData, p, uv, n, mat: int) -> None:
    """One OBJ quad face -> two triangles, split like src/tiny_obj_loader.h:1587-1605."""
    p = _f32(p); uv = _f32(uv); n = _f32(n)
    pf = p.astype(np.float32)
    e02 = pf[2] - pf[0]; e13 = pf[3] - pf[1]
    if float(np.dot(e02, e02)) < float(np.dot(e13, e13)):
        order = [(0, 1, 2), (0, 2, 3)]
    else:
        order = [(0, 1, 3), (1, 2, 3)]
    for tri in order:
        scene.add_triangles(p[list(tri)], np.broadcast_to(n, (3, 3)), uv[list(tri)], mat)


def procedural_texture(size: int = 1024, cells: int = 8, seed: int = 7) -> np.ndarray:
    """A UV-test-like colour grid with a fine pattern (synthetic stand-in for a map_Kd image). Row 0 = bottom."""
    rng = np.random.RandomState(seed)
    palette = rng.randint(40, 256, size=(cells, cells, 3)).astype(np.uint8)
    ys, xs = np.mgrid[0:size, 0:size]
    cy = ys * cells // size
    cx = xs * cells // size
    img = palette[cy, cx].astype(np.int32)
    stripe = (((xs // 4) + (ys // 4)) % 2) * 24
    img = np.clip(img - stripe[..., None], 0, 255).astype(np.uint8)
    return img


def procedural_envmap(width: int = 3584, height: int = 1792, seed: int = 11) -> np.ndarray:
    """Synthetic equirectangular LDR environment (sky gradient, sun disc, ground, noise). Row 0 = bottom."""
    rng = np.random.RandomState(seed)
    v = (np.arange(height) + 0.5) / height            # 0 = bottom (down), 1 = top (up)
    u = (np.arange(width) + 0.5) / width
    vv, uu = np.meshgrid(v, u, indexing="ij")
    sky = np.stack([0.45 + 0.35 * (1 - vv), 0.6 + 0.25 * (1 - vv), 0.95 * np.ones_like(vv)], -1)
    ground = np.stack([0.35 + 0.1 * np.sin(40 * uu), 0.3 + 0.1 * np.cos(31 * uu), 0.22 * np.ones_like(vv)], -1)
    img = np.where((vv > 0.5)[..., None], sky, ground)
    sun = np.exp(-(((uu - 0.3) * 2) ** 2 + (vv - 0.8) ** 2) / 0.0015)
    img = np.clip(img + sun[..., None] * np.array([1.0, 0.95, 0.8]), 0, 1)
    img = img + rng.uniform(-0.02, 0.02, img.shape)
    return np.clip(img * 255.0, 0, 255).astype(np.uint8)


def cornell_box(width: int = 512, height: int = 512, *, texture: Optional[np.ndarray] = None,
                textured_floor: bool = True, aperture: float = 0.5, focus_distance: float = 24.0) -> SceneData:
    """The Cornell box of assets/cornell_box/cornell_box.{obj,mtl}: 12 triangles, ids 0-11,
    lights = ids 8 and 9, floor (ids 10, 11) textured; camera of assets/cornell_box/cornell_box.json:2-16."""
    s = SceneData()
    # material order = cornell_box.mtl order; load_obj appends one default material (src/scene.cpp:215)
    green = s.add_material(Material(albedo=(0.0, 1.0, 0.0)))
    light = s.add_material(Material(albedo=(0.8, 0.8, 0.8), emission=(10.0, 10.0, 10.0)))
    red = s.add_material(Material(albedo=(1.0, 0.0, 0.0)))
    tex_id = -1
    if textured_floor:
        tex_id = s.add_texture(texture if texture is not None else procedural_texture())
    floor = s.add_material(Material(albedo=(1.0, 1.0, 1.0), texture=tex_id))
    white = s.add_material(Material(albedo=(0.8, 0.8, 0.8)))
    s.add_material(Material(albedo=(0.5, 0.5, 0.5)))
    for m in s.materials:  # tinyobj defaults: Pr (roughness) and Pm (metallic) are 0 when absent
        m.roughness = 0.0
        m.metallic = 0.0

    L = 0.947293
    Y = 9.862494
    # roof
    _quad(s, [(5, 10, 5), (-5, 10, 5), (-5, 10, -5), (5, 10, -5)],
          [(0.625, 0.5), (0.625, 0.75), (0.875, 0.75), (0.875, 0.5)], (0, -1, 0), white)
    # back wall
    _quad(s, [(-5, 0, -5), (5, 0, -5), (5, 10, -5), (-5, 10, -5)],
          [(0.375, 0.0), (0.375, 0.25), (0.625, 0.25), (0.625, 0.0)], (0, 0, 1), white)
    # left wall (red)
    _quad(s, [(-5, 0, 5), (-5, 0, -5), (-5, 10, -5), (-5, 10, 5)],
          [(0.375, 0.75), (0.375, 1.0), (0.625, 1.0), (0.625, 0.75)], (1, 0, 0), red)
    # right wall (green)
    _quad(s, [(5, 0, -5), (5, 0, 5), (5, 10, 5), (5, 10, -5)],
          [(0.375, 0.25), (0.375, 0.5), (0.625, 0.5), (0.625, 0.25)], (-1, 0, 0), green)
    # light
    _quad(s, [(-L, Y, -L), (L, Y, -L), (L, Y, L), (-L, Y, L)],
          [(0, 0), (1, 0), (1, 1), (0, 1)], (0, -1, 0), light)
    # floor
    _quad(s, [(5, 0, -5), (-5, 0, -5), (-5, 0, 5), (5, 0, 5)],
          [(0.9999, 0.0001), (0.9999, 0.9999), (0.0001, 0.9999), (0.0001, 0.0001)], (0, 1, 0), floor)

    s.background_color = (0.0, 0.0, 0.0)
    s.camera = CameraDesc(width=width, height=height, fov=14.5, aperture=aperture, focus_distance=focus_distance,
                          position=(0.0, 5.0, 24.0), target=(0.0, 5.0, 0.0))
    return s


def icosphere(level: int):
    """Unit icosphere: golden-ratio icosahedron, midpoint subdivision, vertices renormalised.
    level 6 -> 40 962 vertices / 81 920 faces; level 8 -> 655 362 / 1 310 720."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
                  (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)], np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
                  (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5),
                  (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)], np.int64)
    for _ in range(level):
        nv = v.shape[0]
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        e.sort(axis=1)
        key = e[:, 0] * nv + e[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        a = uniq // nv; b = uniq % nv
        mid = v[a] + v[b]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        v = np.concatenate([v, mid])
        nf = f.shape[0]
        m01 = nv + inv[0:nf]; m12 = nv + inv[nf:2 * nf]; m20 = nv + inv[2 * nf:3 * nf]
        f = np.concatenate([np.stack([f[:, 0], m01, m20], 1), np.stack([f[:, 1], m12, m01], 1),
                            np.stack([f[:, 2], m20, m12], 1), np.stack([m01, m12, m20], 1)])
    return v, f


def blob_mesh(level: int = 6, scale: float = 2.0, offset=(0.0, 2.3, 0.0)):
    """Bumpy sphere with smooth vertex normals: r = 1 + 0.15 sin(5x) sin(7y) sin(3z) on the unit vertex,
    position = scale*r*v + offset, normal = undisplaced unit vertex.  Values are rounded the way an
    OBJ text round trip would round them ('v %.6f', 'vn %.4f') and then through float32."""
    v, f = icosphere(level)
    r = 1.0 + 0.15 * np.sin(5 * v[:, 0]) * np.sin(7 * v[:, 1]) * np.sin(3 * v[:, 2])
    pos = scale * r[:, None] * v + np.asarray(offset, np.float64)
    pos = _f32(np.round(pos, 6))
    nrm = _f32(np.round(v, 4))
    return pos, nrm, f


def add_mesh(scene: SceneData, pos, nrm, faces, mat: int) -> None:
    faces = np.asarray(faces)
    uv = np.zeros((faces.shape[0], 3, 2))
    scene.add_triangles(pos[faces], nrm[faces], uv, mat)


def cornell_with_mesh(level: int = 6, width: int = 1024, height: int = 1024, *, mesh_material: Optional[Material] = None,
                      textured_floor: bool = True, envmap: Optional[np.ndarray] = None,
                      aperture: float = 0.5, focus_distance: float = 24.0) -> SceneData:
    """Mesh first, Cornell box second (the model order of assets/cornell_box/cornell_box.json:17).
    level 6 is the bunny-class stand-in (81 920 triangles), level 8 the ~1.3 M-triangle one."""
    s = SceneData()
    mm = mesh_material if mesh_material is not None else Material(albedo=(0.8, 0.8, 0.8), roughness=0.0, metallic=0.0)
    mesh_mat = s.add_material(mm)
    s.add_material(Material(albedo=(0.5, 0.5, 0.5)))  # load_obj's appended default
    pos, nrm, f = blob_mesh(level)
    add_mesh(s, pos, nrm, f, mesh_mat)
    box = cornell_box(width, height, textured_floor=textured_floor, aperture=aperture, focus_distance=focus_distance)
    mat_off = len(s.materials)
    for t in box.textures:
        s.add_texture(t)
    for m in box.materials:
        s.add_material(m)
    s.add_triangles(box.tri_pos, box.tri_nrm, box.tri_uv, box.tri_mat + mat_off)
    s.camera = box.camera
    if envmap is not None:
        s.background_texture = s.add_texture(envmap)
    else:
        s.background_color = (0.0, 0.0, 0.0)
    return s


def add_json_spheres(scene: SceneData) -> None:
    """The two visible spheres of assets/cornell_box/cornell_box.json:23-34 (setup as in src/main.cpp:124-145)."""
    glass = scene.add_material(Material(type=DIELECTRIC, albedo=(1.0, 1.0, 1.0), roughness=0.5, metallic=0.5))
    scene.add_sphere((3.0, 1.5, 2.5), 1.5, glass)
    mirror = scene.add_material(Material(type=SPECULAR, albedo=(1.0, 1.0, 1.0), roughness=0.0, metallic=0.5))
    scene.add_sphere((-3.0, 1.5, -2.5), 1.5, mirror)


def glass_scene(level: int = 5, width: int = 500, height: int = 500) -> SceneData:
    """Dielectric mesh (bunny.mtl: illum 4 -> DIELECTRIC, Kd 1, Pr 0.5, Pm 0.5) + both JSON spheres in the box."""
    s = cornell_with_mesh(level, width, height,
                          mesh_material=Material(type=DIELECTRIC, albedo=(1.0, 1.0, 1.0), roughness=0.5, metallic=0.5))
    add_json_spheres(s)
    return s


# ---------------------------------------------------------------------------------------------
# Asset files for the command-line driver (`pt`): the same Cornell box as cornell_box(), written as
# OBJ / MTL / PNG / JSON text the way the reference's asset directory is laid out.
_CORNELL_QUADS = [
    # name, material, vertices (4), texture coordinates (4), normal
    ("Back_Roof", "White", [(5, 10, 5), (-5, 10, 5), (-5, 10, -5), (5, 10, -5)],
     [(0.625, 0.5), (0.625, 0.75), (0.875, 0.75), (0.875, 0.5)], (0, -1, 0)),
    ("Back_Wall", "White", [(-5, 0, -5), (5, 0, -5), (5, 10, -5), (-5, 10, -5)],
     [(0.375, 0.0), (0.375, 0.25), (0.625, 0.25), (0.625, 0.0)], (0, 0, 1)),
    ("Left_Wall", "Red", [(-5, 0, 5), (-5, 0, -5), (-5, 10, -5), (-5, 10, 5)],
     [(0.375, 0.75), (0.375, 1.0), (0.625, 1.0), (0.625, 0.75)], (1, 0, 0)),
    ("Right_Wall", "Green", [(5, 0, -5), (5, 0, 5), (5, 10, 5), (5, 10, -5)],
     [(0.375, 0.25), (0.375, 0.5), (0.625, 0.5), (0.625, 0.25)], (-1, 0, 0)),
    ("Light_Plane", "Material", [(-0.947293, 9.862494, -0.947293), (0.947293, 9.862494, -0.947293),
                                 (0.947293, 9.862494, 0.947293), (-0.947293, 9.862494, 0.947293)],
     [(0, 0), (1, 0), (1, 1), (0, 1)], (0, -1, 0)),
    ("Floor", "Textured", [(5, 0, -5), (-5, 0, -5), (-5, 0, 5), (5, 0, 5)],
     [(0.9999, 0.0001), (0.9999, 0.9999), (0.0001, 0.9999), (0.0001, 0.0001)], (0, 1, 0)),
]


def write_cornell_assets(directory, width=64, height=64, texture: Optional[np.ndarray] = None, spheres: bool = False,
                         aperture: float = 0.5, focus_distance: float = 24.0, background_texture: Optional[str] = None,
                         textured_sphere: Optional[str] = None) -> str:
    """Writes cornell_box.obj / .mtl / floor texture / config JSON into `directory`; returns the JSON path.
    The geometry, materials and camera equal cornell_box(width, height, texture=...)."""
    import json
    import os
    from PIL import Image as PILImage
    os.makedirs(directory, exist_ok=True)
    tex = texture if texture is not None else procedural_texture()
    PILImage.fromarray(np.flipud(tex)).save(os.path.join(directory, "floor.png"))  # file rows are top-down
    with open(os.path.join(directory, "cornell_box.mtl"), "w") as f:
        def mtl(name, kd, ke=(0, 0, 0), extra=""):
            f.write(f"newmtl {name}\nNs 225.000000\nKa 1.000000 1.000000 1.000000\nKd {kd[0]:.6f} {kd[1]:.6f} {kd[2]:.6f}\n"
                    f"Ks 0.500000 0.500000 0.500000\nKe {ke[0]:.6f} {ke[1]:.6f} {ke[2]:.6f}\nNi 1.450000\nd 1.000000\nillum 2\n{extra}\n")
        mtl("Green", (0, 1, 0))
        mtl("Material", (0.8, 0.8, 0.8), (10, 10, 10))
        mtl("Red", (1, 0, 0))
        mtl("Textured", (1, 1, 1), extra="map_Kd floor.png\n")
        mtl("White", (0.8, 0.8, 0.8))
    with open(os.path.join(directory, "cornell_box.obj"), "w") as f:
        f.write("# generated\nmtllib cornell_box.mtl\n")
        nv = nt = nn = 0
        for name, mat, verts, uvs, normal in _CORNELL_QUADS:
            f.write(f"o {name}\n")
            for v in verts:
                f.write(f"v {v[0]:.6f} {v[1]:.6f} {v[2]:.6f}\n")
            for t in uvs:
                f.write(f"vt {t[0]:.6f} {t[1]:.6f}\n")
            f.write(f"vn {normal[0]:.4f} {normal[1]:.4f} {normal[2]:.4f}\nusemtl {mat}\ns off\n")
            f.write("f " + " ".join(f"{nv + k + 1}/{nt + k + 1}/{nn + 1}" for k in range(4)) + "\n")
            nv += 4; nt += 4; nn += 1
    cfg = {
        "image_width": width, "image_height": height, "camera_aperture": aperture,
        "camera_focus_distance": focus_distance, "camera_position": {"x": 0, "y": 5, "z": 24},
        "camera_target": {"x": 0, "y": 5, "z": 0}, "camera_fov": 14.5,
        "models": [os.path.join(directory, "cornell_box.obj")],
        "background_color": {"x": 0, "y": 0, "z": 0},
    }
    if spheres:
        cfg["spheres"] = [
            {"center": {"x": 3, "y": 1.5, "z": 2.5}, "radius": 1.5, "type": "TRANSMISSIVE"},
            {"center": {"x": -3, "y": 1.5, "z": -2.5}, "radius": 1.5, "type": "SPECULAR", "roughness": 0.0},
            {"hidden": True, "center": {"x": 0, "y": 8, "z": 0}, "radius": 1.5, "emissive": {"x": 2, "y": 2, "z": 2}},
        ]
    if background_texture:  # wins over background_color (src/main.cpp:147-158)
        cfg["background_texture"] = background_texture
    if textured_sphere:
        cfg.setdefault("spheres", []).append({"center": {"x": 0.5, "y": 1.2, "z": 1.5}, "radius": 1.2, "texture": textured_sphere})
    path = os.path.join(directory, "cornell_box.json")
    with open(path, "w") as f:
        json.dump(cfg, f, indent=1)
    return path
