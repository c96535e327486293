// Device-memory layout of a scene: what the host flattens the reference's
// pointer-rich structures (264-byte Primitive, src/geometry.h:54-71; pointer
// based BVH::Node, src/bvh.h:14-21; Material, src/material.h:11-36) into.
// Every record is a multiple of 16 bytes so that a lane fetches it with
// 128-bit loads.
#pragma once

#include "ptk_math.h"

namespace ptk {

constexpr uint32_t kNoPrim = 0xFFFFFFFFu;
constexpr int kMaxLeafSize = 4;   // primitives per BVH leaf
// Per-lane traversal stack entries: a sentinel, one per tree level, one of scratch.  27 entries x 64 lanes
// x 4 bytes + the tile's pixel sums = 7680 bytes of LDS per wave, which lets 20 waves (5 per SIMD) share a
// CU's 160 KB; 28 entries no longer do (measured: 5 waves +8 %, and back to the 4-wave rate at 28).
#ifndef PTK_STACK_DEPTH
#define PTK_STACK_DEPTH 27
#endif
constexpr int kStackDepth = PTK_STACK_DEPTH;
constexpr int kMaxTreeDepth = kStackDepth - 2;  // the builder keeps trees this shallow (median splits if need be)

// Inner node with the boxes of BOTH children (one 64-byte fetch decides both).
// child >= 0: index of an inner node.  child < 0: leaf, ~child = (first_prim << 3) | (count - 1).
// kEmptyChild marks a child that holds nothing (its box is inverted so it is never entered).
struct alignas(16) NodePair {
  f4 xy0;  // child 0: min.x, max.x, min.y, max.y
  f4 xy1;  // child 1: min.x, max.x, min.y, max.y
  f4 zz;   // child 0: min.z, max.z; child 1: min.z, max.z
  int32_t child0, child1, pad0, pad1;
};
static_assert(sizeof(NodePair) == 64, "NodePair must be one 64-byte line");

enum PrimKind : uint32_t { kTriangle = 0, kSphere = 1 };

// Intersection record, stored in BVH leaf order.
//   triangle: a = (v0.x, v0.y, v0.z, e1.x)  b = (e1.y, e1.z, e2.x, e2.y)  c = (e2.z, id, kind, 0)
//   sphere:   a = (c.x, c.y, c.z, radius)   b = unused                    c = (0,    id, kind, 0)
// id is the reference's primitive id (insertion order, src/scene.cpp:59).
struct alignas(16) PrimRec {
  f4 a, b, c;
};
static_assert(sizeof(PrimRec) == 48, "PrimRec is 48 bytes");

// Shading attributes, indexed by primitive id; read once per closest hit.
//   triangle: n0,n1,n2 vertex normals, t0,t1,t2 texture coordinates
//   sphere:   n0 = centre, n1.x = radius
struct alignas(16) ShadeRec {
  float n0[3], n1[3], n2[3];
  float t0[2], t1[2], t2[2];
  uint32_t material;
};
static_assert(sizeof(ShadeRec) == 64, "ShadeRec is 64 bytes");

enum MaterialFlags : uint32_t {
  kMatLight = 1u,     // any emission component > 0          (Primitive::is_light, src/geometry.cpp:225)
  kMatSpecular = 2u,  // Material::is_perfectly_specular,     src/material.h:30-35
};

struct alignas(16) MaterialRec {
  uint32_t type;   // Material::Type bits
  int32_t texture; // -1 = none
  uint32_t flags;
  float ior;
  float albedo[3];
  float roughness;
  float emission[3];
  float metallic;
};
static_assert(sizeof(MaterialRec) == 48, "MaterialRec is 48 bytes");

// One emissive primitive, self-contained for next-event estimation
// (Primitive::sample_point / sample_area, src/geometry.cpp:228-254).
struct alignas(16) LightRec {
  uint32_t kind, id;
  float area;      // triangle: |cross|/2; sphere: 2*pi*r^2
  float radius;
  float p0[3], p1[3], p2[3];  // triangle vertices; sphere: p0 = centre
  float emission[3];
  float n0[3], n1[3], n2[3];  // triangle vertex normals
  float pad[3];
};
static_assert(sizeof(LightRec) == 112, "LightRec is 112 bytes");

struct alignas(16) TextureRec {
  uint64_t offset;  // byte offset of the first texel in the texel pool
  int32_t width, height;
};

// Everything a kernel needs to know about the scene; passed by value as a kernel argument.
struct SceneView {
  const NodePair* nodes;
  const PrimRec* prims;
  const ShadeRec* shade;
  const MaterialRec* materials;
  const LightRec* lights;
  const TextureRec* textures;
  const uint8_t* texels;
  int32_t n_prims, n_lights, n_nodes, n_materials;
  int32_t n_triangles;  // primitive ids 0 .. n_triangles-1 are triangles, the rest spheres (the C ABI's order)
  int32_t background_texture;  // -1 = none
  int32_t background_is_color; // all components >= 0 (src/scene.cpp:39)
  float background_color[3];
};

struct CameraView {  // src/camera.h:21-26 after compute()
  int32_t width, height;
  float position[3], forward[3], right[3], up[3];
  float plane_w, plane_h;   // 2 * tan(fov/2) * aspect and 2 * tan(fov/2)  (src/camera.cpp:28-33)
  float defocus_radius;     // focus * tan(radians(aperture / 2)), 0 when depth of field is off (src/camera.cpp:44-50)
  float focus_distance;
};

struct FlagView {  // src/config.h:3-8
  int32_t debug_normal, russian_roulette, direct_light_sampling, indirect_light_sampling, importance_sample,
      check_hemisphere;
};

struct Counters {
  uint32_t camera_rays, segments, queries, box_tests, prim_tests;
};

}  // namespace ptk
