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
// Per-lane traversal stack entries: a sentinel, one per tree level, one of scratch.  kStackDepth is the most a stack
// can hold, and so what bounds the depth of a tree (kMaxTreeDepth); how many of the entries live in LDS depends on
// the kernel form, the rest are overflow words in the lane's private memory (LdsStack in ptk_kernels.hip).  LDS is
// handed out in granules of 1280 bytes and a one-wave workgroup also keeps its tile's pixel sums (768 bytes) there,
// so the LDS part decides how many waves a CU's 160 KB can hold:
//   deep    27 entries x 64 lanes x 4 bytes + sums = 7680 bytes = 6 granules: 20 waves per CU, 5 per SIMD
//           (28 entries no longer fit: measured, back to the 4-wave rate); the older form for trees of more than 20
//           levels, kept for comparison
//   shallow 22 entries + sums = 6400 bytes = 5 granules: 24 waves per CU, 6 per SIMD (the kernel is then built
//           for 80 vector registers instead of 96); trees up to 20 levels without overflow words.  The sixth wave is
//           worth 5-11 % on the 82k-triangle scenes; capping the 1.3M-triangle scene's tree at 20 levels to get it
//           costs 12 %.
// The kernels exist in these forms and the scene's tree depth picks one.  The depth limit is well above what a
// balanced tree of the largest scenes needs (1.3M triangles: 21 levels): both builders fall back to median splits
// where it binds, and with a limit of 25 that was where more than half of that scene's leaves ended up.
#ifndef PTK_STACK_DEPTH
#define PTK_STACK_DEPTH 34
#endif
constexpr int kStackDepth = PTK_STACK_DEPTH;   // the largest stack, LDS part and overflow words together
#ifndef PTK_STACK_DEPTH_DEEP
#define PTK_STACK_DEPTH_DEEP 27
#endif
constexpr int kStackDepthDeep = PTK_STACK_DEPTH_DEEP < PTK_STACK_DEPTH ? PTK_STACK_DEPTH_DEEP : PTK_STACK_DEPTH;  // deep form: entries in LDS
#ifndef PTK_STACK_DEPTH_SHALLOW
#define PTK_STACK_DEPTH_SHALLOW 22
#endif
constexpr int kStackDepthShallow = PTK_STACK_DEPTH_SHALLOW;  // (smaller in test builds, to exercise the overflow entries)
constexpr int kMaxTreeDepth = kStackDepth - 2;  // the builder keeps trees this shallow (median splits if need be)
constexpr int kShallowTreeDepth = kStackDepthShallow - 2;
// With three entries fewer (19: trees up to 17 levels without overflow entries) the same five granules also hold
// the scene's material and light tables, 768 bytes of them: records that every path vertex reads and that all lanes
// share, at three and seven vector loads each from global memory.  Scenes whose tables fit get these forms; the
// entries beyond the nineteenth then live in the overflow words (ptk_kernels.hip, LdsStack).  Measured against the
// 22-entry forms: +2.7 % on the 82k-triangle scene (20 levels: the overflow checks cost 3.9 %, the tables give 6.5 %),
// +4.3 % on the 1.3M-triangle scene.
#ifndef PTK_STACK_DEPTH_TABLES
#define PTK_STACK_DEPTH_TABLES 19
#endif
constexpr int kStackDepthTables = PTK_STACK_DEPTH_TABLES;
constexpr int kTablesTreeDepth = kStackDepthTables - 2;
constexpr int kTableBytes = 768;
#ifndef PTK_PARK
#define PTK_PARK 0  // 1: the waiting light term and the next direction of a path in LDS, 2: the light term only
#endif
constexpr int kParkBytes = (PTK_PARK == 2 ? 3 : 6) * 64 * 4;  // vectors of a lane that wait in LDS while a shadow ray is traced (LdsPark in ptk_kernels.hip)
// Tiny scenes (the Cornell box: 5 nodes, 12 triangles, 6 materials, 2 lights): the render kernel has a third form
// that copies the whole scene except the texels into LDS when a workgroup starts and works from there - no
// vector-memory instruction in the traversal, and none in shading but the texture fetches.  16 nodes (512 B) + 24
// primitive records (1 152) + 24 shading records (1 536) + 8 materials (384) + 4 lights (448) + 6-entry stacks
// (1 536) + sums (768) = 6 336 bytes, inside the five LDS granules of the six-wave forms.
constexpr int kTinyNodes = 16, kTinyPrims = 24, kTinyMaterials = 8, kTinyLights = 4, kStackDepthTiny = 6, kTinyTreeDepth = kStackDepthTiny - 2;

// Inner node with the boxes of BOTH children, 32 bytes.  The render kernel is bound by the number of bytes its
// lanes pull through the L1 (every lane is at its own node), which is why the boxes are not stored as floats:
// box planes are 16-bit positions on a grid over the scene's bounds (BoxGrid), rounded outwards, low plane in
// the low half of a word and the box's extent (high plane - low plane) in the high half (grid_pack).  The words of
// one axis sit next to each other, child 0 first.
// child >= 0: index of an inner node.  child < 0: leaf, see leaf_code().
struct alignas(16) NodePair {
  uint32_t x0, x1;  // x planes of child 0, of child 1
  uint32_t y0, y1;
  uint32_t z0, z1;
  int32_t child0, child1;
};
static_assert(sizeof(NodePair) == 32, "NodePair is 32 bytes");

// plane position = origin + q * scale for q in 0 .. 65535, per axis
struct BoxGrid {
  float origin[3], scale[3];
};

// Grid covering the box [lo, hi] (which has been padded already, so no axis is empty).
PTK_HD BoxGrid make_box_grid(const float lo[3], const float hi[3])
{
  BoxGrid g;
  for (int a = 0; a < 3; a++) {
    g.origin[a] = lo[a];
    g.scale[a] = fmaxf((hi[a] - lo[a]) / 65534.0f, 1e-30f);
  }
  return g;
}

// Largest grid position not above v / smallest not below v, as the traversal evaluates positions
// (fused multiply-add in fp32), so that a quantised box always contains the box it was made from.
PTK_HD uint32_t grid_floor(float v, float origin, float scale)
{
  float q = fminf(fmaxf(floorf((v - origin) / scale), 0.0f), 65535.0f);
  while (q > 0.0f && fmaf(q, scale, origin) > v) q -= 1.0f;
  return (uint32_t)q;
}
PTK_HD uint32_t grid_ceil(float v, float origin, float scale)
{
  float q = fminf(fmaxf(ceilf((v - origin) / scale), 0.0f), 65535.0f);
  while (q < 65535.0f && fmaf(q, scale, origin) < v) q += 1.0f;
  return (uint32_t)q;
}
// One axis of a box as a node stores it: the low plane's grid position in the low half of the word and, in the
// high half, the extent (high plane - low plane, in grid steps) as the top sixteen bits of a float that is not below
// it (a bfloat16, rounded up).  The slab test wants the extent as a float (hit_box), and with this encoding the WHOLE
// word read as a float is one: the low plane's bits sit in the low mantissa bits, where they can only add (less than
// 1/128 of the extent).  The extent then costs the traversal no instruction at all, where an integer extent cost a
// conversion - and conversions issue at half the rate of the multiply-adds around them.  The box grows by at most
// 1.6 % of its size on its high side; it never shrinks.
PTK_HD uint32_t grid_pack(uint32_t q_lo, uint32_t extent_steps)
{
  const uint32_t bits = as_uint((float)extent_steps);  // < 2^16: exact
  return q_lo | (((bits + 0xffffu) >> 16) << 16);
}
// the extent the traversal sees is not below this one (a whole number of grid steps: a bfloat16 of an integer below
// 2^16, rounded up, is an integer)
PTK_HD uint32_t grid_extent_steps(uint32_t word) { return (uint32_t)as_float(word & 0xffff0000u); }
PTK_HD float grid_extent(uint32_t word) { return as_float(word); }  // as hit_box takes it
PTK_HD uint32_t grid_word(float lo, float hi, float origin, float scale)
{
  const uint32_t q_lo = grid_floor(lo, origin, scale), q_hi = grid_ceil(hi, origin, scale);
  return grid_pack(q_lo, q_hi - q_lo);
}
PTK_HD void set_child_box(NodePair& n, int which, const float lo[3], const float hi[3], const BoxGrid& g)
{
  uint32_t x = grid_word(lo[0], hi[0], g.origin[0], g.scale[0]);
  uint32_t y = grid_word(lo[1], hi[1], g.origin[1], g.scale[1]);
  uint32_t z = grid_word(lo[2], hi[2], g.origin[2], g.scale[2]);
  if (which == 0) { n.x0 = x; n.y0 = y; n.z0 = z; }
  else { n.x1 = x; n.y1 = y; n.z1 = z; }
}

// A leaf link: ~((first record << 3) | (spheres ? 4 : 0) | (count - 1)), count <= kMaxLeafSize = 4.  The sphere bit
// says that at least one of the leaf's records is a sphere; only such leaves are scanned by the loop that looks at
// every record's kind, all others by a loop that knows it has triangles (a scene has a handful of spheres, if any,
// and the kind test with its branch was measured at 3.5 % of a frame even in a scene without one).
static_assert(kMaxLeafSize <= 4, "a leaf link keeps count - 1 in two bits");
constexpr uint32_t kLeafSphereBit = 4u;
PTK_HD int32_t leaf_code(uint32_t first, uint32_t count, bool spheres)
{
  return (int32_t) ~((first << 3) | (spheres ? kLeafSphereBit : 0u) | (count - 1u));
}
PTK_HD uint32_t leaf_first(int32_t link) { return (~(uint32_t)link) >> 3; }
PTK_HD uint32_t leaf_count(int32_t link) { return ((~(uint32_t)link) & 3u) + 1u; }
PTK_HD bool leaf_has_spheres(int32_t link) { return ((~(uint32_t)link) & kLeafSphereBit) != 0u; }

enum PrimKind : uint32_t { kTriangle = 0, kSphere = 1 };

// Intersection record, stored in BVH leaf order.
//   triangle: a = (v0.x, v0.y, v0.z, e1.x)  b = (e1.y, e1.z, e2.x, e2.y)  c = (e2.z, id,                  -, -)
//   sphere:   a = (c.x, c.y, c.z, radius)   b = unused                    c = (0,    id | kPrimSphereBit, -, -)
// id is the reference's primitive id (insertion order, src/scene.cpp:59).  The traversal fetches 40 bytes.
constexpr uint32_t kPrimSphereBit = 0x80000000u;
constexpr uint32_t kPrimIdMask = 0x7fffffffu;
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
  float albedo[3];  // texture >= 0: the bits of the texture's byte offset in the texel pool, its width and its height
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
  uint32_t flat;              // triangle: the three normals are equal (the interpolated one is then n0)
  float pad[2];
};
static_assert(sizeof(LightRec) == 112, "LightRec is 112 bytes");

struct alignas(16) TextureRec {
  uint64_t offset;  // byte offset of the first texel in the texel pool
  int32_t width, height;
};

// The texture records are preceded, in the same array, by a table of pow(i / 255, 2.2) for i = 0 .. 255 (what a
// texel's byte becomes before it is used as a colour): 256 floats = kGammaLutRecords records' worth.  SceneView::textures
// points at the first real record, the table lies just below it - no further kernel argument for it.
constexpr int kGammaLutRecords = 256 * (int)sizeof(float) / (int)sizeof(TextureRec);
static_assert(kGammaLutRecords * sizeof(TextureRec) == 256 * sizeof(float), "the table fills whole records");

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
  BoxGrid grid;         // what the nodes' box planes are positions on
  int32_t background_texture;  // -1 = none
  int32_t background_is_color; // all components >= 0 (src/scene.cpp:39)
  float background_color[3];
};

PTK_HD const float* gamma_lut(const SceneView& sc) { return reinterpret_cast<const float*>(sc.textures) - 256; }

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

// What the render kernel reads through a pointer instead of taking by value: the camera and the switches are
// needed only where paths start and where hits are shaded.  As by-value kernel arguments they sit in scalar
// registers for the whole kernel and push the hot loops' lane masks out into spill lanes (the kernel has been
// seen to lose 4-6 % to ONE more scalar argument); behind a pointer they are loaded where they are used.
// (The scene's own pointers and counts stay by-value arguments: moving them here as well measured 1 % slower.)
struct DeviceConsts {
  CameraView cam;
  FlagView fl;
  float camera_shift;  // camera_entry_shift() of the scene and camera in use
};

struct Counters {
  uint32_t camera_rays, segments, queries, box_tests, prim_tests;
#ifdef PTK_PHASE_STATS
  uint32_t leaf_iterations = 0;  // passes of a leaf loop, counted once per wave (by the first active lane)
#endif
};

}  // namespace ptk
