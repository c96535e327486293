// Closest-hit query: what Scene::find_intersection -> BVH::traverse ->
// Node::intersect -> Primitive::intersect compute in the reference
// (src/scene.cpp:19, src/bvh.cpp:10-52, src/geometry.cpp:108-223), restated
// for one GPU lane per ray:
//   * iterative, near-child-first traversal of a flattened BVH with the far
//     child pushed on a short per-lane stack, and pruning against the closest
//     hit so far (the reference visits both children of every node it enters
//     and never prunes, src/bvh.cpp:20-22);
//   * the same acceptance rules: |N.d| < 1e-5 rejects (unnormalised N,
//     src/geometry.cpp:121), t >= 1e-5 (src/geometry.cpp:131), all three edge
//     functions >= 0 (src/geometry.cpp:142-154), spheres take the nearer root
//     inside the open interval (1e-5, 1e9) and otherwise the farther one
//     (src/geometry.cpp:41-45), the strictly closer hit wins (src/geometry.h:22).
//
// fp32 detail: the reference starts secondary rays exactly on the surface
// they leave and relies on double precision to get t ~ 1e-16 < 1e-5 for the
// surface itself.  In fp32 that self-distance can exceed 1e-5, so a ray
// carries the id of the primitive it starts on: a triangle with that id is
// not tested, a sphere with that id only offers its far root.
#pragma once

#include "ptk_layout.h"

namespace ptk {

constexpr float kTMin = 1e-5f;  // Interval(0.00001, 1e9), src/geometry.cpp:165
constexpr float kTMax = 1e9f;

struct Hit {
  float t;
  uint32_t id;    // reference primitive id, kNoPrim = miss (ids below SceneView::n_triangles are triangles)
  float u, v;     // triangle: weights of v1 and v2
};

// Möller-Trumbore form of the reference's plane + three-edge test.
// det = e1 . (d x e2) = -N.d with N = e1 x e2, so |det| is the reference's |N.d|.
PTK_HD bool hit_triangle(const PrimRec& p, v3 o, v3 d, float tbest, float& t, float& u, float& v)
{
  v3 v0 = V3(p.a.x, p.a.y, p.a.z);
  v3 e1 = V3(p.a.w, p.b.x, p.b.y);
  v3 e2 = V3(p.b.z, p.b.w, p.c.x);
  // Straight-line: every quantity is computed and the acceptance tests are combined at the end
  // (a wave would run all of it anyway as soon as one lane passes each test).  A zero det makes
  // inv infinite and the comparisons, all written to fail on NaN, reject.
  v3 pvec = cross(d, e2);
  float det = dot(e1, pvec);
  float inv = frcp(det);
  v3 tvec = o - v0;
  float uu = dot(tvec, pvec) * inv;
  v3 qvec = cross(tvec, e1);
  float vv = dot(d, qvec) * inv;
  float tt = dot(e2, qvec) * inv;
  t = tt; u = uu; v = vv;
  // edge functions >= 0  <=>  u >= 0, v >= 0, 1-u-v >= 0
  return (fabsf(det) >= kTMin) & (uu >= 0.0f) & (vv >= 0.0f) & (uu + vv <= 1.0f) & (tt >= kTMin) & (tt < tbest);
}

// Roots of |o + t d - c|^2 = r^2 as (h -+ sqrt(disc)) / a with h = d.(c-o), evaluated from the
// perpendicular offset l = (c-o) - (h/a) d, which avoids the cancellation of |c-o|^2 - r^2 for
// far-away origins.  own_surface: the ray starts on this sphere, so one root is 0 by construction
// and the other is the chord 2h/a (only ahead of the ray when it points inward).
PTK_HD bool hit_sphere(const PrimRec& p, v3 o, v3 d, float tbest, bool own_surface, float& t)
{
  v3 oc = V3(p.a.x, p.a.y, p.a.z) - o;
  float r = p.a.w;
  float a = dot(d, d);
  float h = dot(d, oc);
  float inv_a = frcp(a);
  float root;
  if (own_surface) {
    root = 2.0f * h * inv_a;
  } else {
    v3 l = oc - d * (h * inv_a);
    float disc = a * (r * r - dot(l, l));
    if (disc < 0.0f) return false;
    float sqrtd = sqrtf(disc);
    root = (h - sqrtd) * inv_a;
    if (!(kTMin < root && root < kTMax)) root = (h + sqrtd) * inv_a;
  }
  if (!(kTMin < root && root < kTMax)) return false;
  if (!(root < tbest)) return false;
  t = root;
  return true;
}

// min of two numbers that are known not to be NaN.  fminf would first quiet a possible signalling NaN
// in an operand it knows nothing about (one more instruction per call; tbest is such an operand).
PTK_HD float min_raw(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return a < b ? a : b;
#endif
}

// Slab test of one child box against [0, tbest]; returns the entry distance in tn.  (lo, e): the box's low grid
// plane and its extent hi - lo per axis, as floats (what the node stores).  The ray carries inv = grid scale / d and oi = (o - grid origin) / d per axis, so
// the distance to grid plane q is fma(q, inv, -oi).  Which of an axis' two planes is the near one depends on
// the sign of d only, so the ray also carries inv split by sign (neg = min(inv, 0), pos = max(inv, 0)) and the
// two distances are
//     near = t(lo) + e * neg,   far = t(lo) + e * pos
// - three fused multiply-adds per axis, all of which issue at the full vector rate on this
// hardware (v_min_f32 / v_max_f32 take twice as long as v_fma_f32: see tools/ubench/valu_rate.hip).
// The caller keeps 1/d finite (axis-parallel rays get +-1e20), so a slab the origin lies in gives -huge..+huge
// and a slab it lies outside gives two huge values of one sign.  The rounding of these products is covered by
// the padding the builder puts on every box.
struct RaySlabs {
  v3 inv, oi, neg, pos;
};
PTK_HD RaySlabs ray_slabs(v3 inv, v3 oi)
{
  RaySlabs r;
  r.inv = inv; r.oi = oi;
  r.neg = V3(fminf(inv.x, 0.0f), fminf(inv.y, 0.0f), fminf(inv.z, 0.0f));
  r.pos = V3(inv.x - r.neg.x, inv.y - r.neg.y, inv.z - r.neg.z);
  return r;
}
PTK_HD bool hit_box(float xlo, float ex, float ylo, float ey, float zlo, float ez, const RaySlabs& r, float tbest, float& tn)
{
  const float tx = fmaf(xlo, r.inv.x, -r.oi.x);
  const float ty = fmaf(ylo, r.inv.y, -r.oi.y);
  const float tz = fmaf(zlo, r.inv.z, -r.oi.z);
  const float nx = fmaf(ex, r.neg.x, tx), fx = fmaf(ex, r.pos.x, tx);
  const float ny = fmaf(ey, r.neg.y, ty), fy = fmaf(ey, r.pos.y, ty);
  const float nz = fmaf(ez, r.neg.z, tz), fz = fmaf(ez, r.pos.z, tz);
  const float tnear = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));
  const float tfar = fminf(fminf(fx, fy), min_raw(fz, tbest));
  tn = tnear;
  return tnear <= tfar;  // no slack needed: the builder pads every box well beyond the rounding of these products
}

// ---------------------------------------------------------------------------------
// Resumable traversal.  A lane's query is a small state (ray, closest hit so far, current
// node) plus its stack; trav_inner() visits one inner node, trav_leaf() intersects one leaf.
// Splitting the work this way lets a wave run "all lanes descend" and "all lanes intersect"
// as separate uniform loops, and lets a lane whose query has finished pick up its next ray
// while its neighbours are still traversing.
//
// Stack: reset(), pop() -> uint32_t, and step(do_push, v, do_pop, top) -> the top entry as it was before the
// call, after which v has been pushed or the top popped as asked (never both); the last argument is what prefetch()
// returned at the start of the node visit that ends in step() (the top entry: a stack in slow memory starts its read there).  reset() leaves
// kTravDone at the bottom, so popping an exhausted stack yields the "finished" node.
constexpr int32_t kTravDone = (int32_t)0x80000000;  // not a valid leaf code for < 2^28 primitives

struct Traversal {
  v3 o, d, inv, oi;  // inv = grid scale / d, oi = (o - grid origin) / d: see hit_box
  Hit hit;
  int32_t node;      // >= 0 inner node, kTravDone finished, other negative values: leaf
  uint32_t skip_id;  // primitive the ray starts on (kNoPrim for none)
  // Visibility queries: the caller only asks whether `target` is the closest hit.  A hit on any other
  // primitive nearer than t_stop answers that (no), so the traversal ends there.  t_stop < 0: a plain
  // closest-hit query.
  uint32_t target;
  float t_stop;
};

PTK_HD bool trav_done(const Traversal& tr) { return tr.node == kTravDone; }
PTK_HD bool trav_at_inner(const Traversal& tr) { return tr.node >= 0; }
PTK_HD bool trav_at_leaf(const Traversal& tr) { return tr.node < 0 && tr.node != kTravDone; }

// A query starts in two parts, so that a kernel whose phases each set up queries for some lanes can run the second,
// arithmetic part once for all of them: trav_set_query() only stores the ray and its limits, trav_start() derives
// what the node visits need from them and empties the stack.  trav_begin() is both.
PTK_HD void trav_set_query(Traversal& tr, const SceneView& sc, v3 o, v3 d, float tmax, uint32_t skip_id,
                           uint32_t target = kNoPrim, float t_stop = -1.0f)
{
  tr.target = target;
  tr.t_stop = t_stop;
  tr.o = o; tr.d = d;
  tr.hit.t = tmax;
  tr.skip_id = skip_id;
  tr.node = sc.n_prims > 0 ? 0 : kTravDone;
}
template <class Stack>
PTK_HD void trav_start(Traversal& tr, Stack& st, const SceneView& sc)
{
  const v3 o = tr.o, d = tr.d;
  // 1/d is kept finite: with an infinite 1/d the fused form below would turn "inf - inf" into NaN for one
  // plane of a slab that contains the origin.  1e20 keeps every product far inside the float range.
  const float big = 1e20f;
  const v3 rd = V3(fmaxf(fminf(frcp(d.x), big), -big), fmaxf(fminf(frcp(d.y), big), -big),
                   fmaxf(fminf(frcp(d.z), big), -big));
  tr.inv = V3(rd.x * sc.grid.scale[0], rd.y * sc.grid.scale[1], rd.z * sc.grid.scale[2]);
  tr.oi = V3((o.x - sc.grid.origin[0]) * rd.x, (o.y - sc.grid.origin[1]) * rd.y, (o.z - sc.grid.origin[2]) * rd.z);
  tr.hit.id = kNoPrim;
  tr.hit.u = tr.hit.v = 0.0f;
  st.reset();
}
template <class Stack, bool DEFER = false>
PTK_HD void trav_begin(Traversal& tr, Stack& st, const SceneView& sc, v3 o, v3 d, float tmax, uint32_t skip_id,
                       uint32_t target = kNoPrim, float t_stop = -1.0f)
{
  trav_set_query(tr, sc, o, d, tmax, skip_id, target, t_stop);
  if (!DEFER) trav_start(tr, st, sc);  // DEFER: the caller runs trav_start() before the query's first node visit
}

// How far a ray that starts far outside the scene can be advanced before it meets the scene's bounds.
// hit_box evaluates plane distances as differences of numbers that grow with the origin's distance from the
// scene, so their rounding grows with that distance while the padding of the boxes does not: from more than about
// thirty scene extents away a thin box could be missed.  Camera rays and the rays of the batch entry point
// (the only ones that can start outside the bounds) are therefore moved to half a chord in front of the bounds
// whenever the origin is farther from them than twice the ray's chord through them; hit distances are then
// measured from the moved origin and this distance is added back.  0 for every other ray.  (The render kernel
// does the same for camera rays with one distance per launch: camera_entry_shift.)
PTK_HD float scene_entry_shift(const SceneView& sc, v3 o, v3 d)
{
  const float big = 1e20f;
  const v3 rd = V3(fmaxf(fminf(frcp(d.x), big), -big), fmaxf(fminf(frcp(d.y), big), -big), fmaxf(fminf(frcp(d.z), big), -big));
  const v3 lo = V3((sc.grid.origin[0] - o.x) * rd.x, (sc.grid.origin[1] - o.y) * rd.y, (sc.grid.origin[2] - o.z) * rd.z);
  const v3 hi = V3((sc.grid.origin[0] + 65535.0f * sc.grid.scale[0] - o.x) * rd.x,
                   (sc.grid.origin[1] + 65535.0f * sc.grid.scale[1] - o.y) * rd.y,
                   (sc.grid.origin[2] + 65535.0f * sc.grid.scale[2] - o.z) * rd.z);
  const float tn = fmaxf(fmaxf(fminf(lo.x, hi.x), fminf(lo.y, hi.y)), fminf(lo.z, hi.z));
  const float tf = fminf(fminf(fmaxf(lo.x, hi.x), fmaxf(lo.y, hi.y)), fmaxf(lo.z, hi.z));
  const float chord = tf - tn;
  if (!(chord > 0.0f) || !(tn > 2.0f * chord) || !(tn < 1e18f)) return 0.0f;  // misses, near origins, NaN
  // stay clear of the bounds: by half a chord, and by enough that the "t >= 1e-5" rule, which is now applied
  // from the moved origin, cannot reject anything inside them (a flat scene's chord is next to nothing)
  const float shift = tn - 0.5f * chord - fmaxf(tn * (1.0f / 1024.0f), 2e-4f);
  return shift > 0.0f ? shift : 0.0f;
}

// Experiment hook (-DPTK_BURN_*): twelve extra vector instructions, to weigh a code region's share of the
// kernel's time by how much slower it gets.
#if defined(__HIP_DEVICE_COMPILE__)
#define PTK_BURN12(x)                                                                                              \
  asm volatile("v_mov_b32 %0, %0\nv_mov_b32 %0, %0\nv_mov_b32 %0, %0\nv_mov_b32 %0, %0\nv_mov_b32 %0, %0\nv_mov_b32 %0, %0\n" \
               "v_mov_b32 %0, %0\nv_mov_b32 %0, %0\nv_mov_b32 %0, %0\nv_mov_b32 %0, %0\nv_mov_b32 %0, %0\nv_mov_b32 %0, %0"    \
               : "+v"(x))
#else
#define PTK_BURN12(x) (void)(x)
#endif

// Where a traversal step fetches nodes and primitive records from: global memory, or (scenes of a few dozen
// primitives) the copy that the workgroup made in LDS when it started.  Offsets are bytes from the array's start.
struct GlobalMem {
  static constexpr bool kPrefetch = true;  // fetch a leaf's next record while the current one is tested
  PTK_HD NodePair node(const SceneView& sc, uint32_t off) const
  {
    return *reinterpret_cast<const NodePair*>(reinterpret_cast<const char*>(sc.nodes) + off);  // loaded whole, by value: two 16-byte fetches
  }
  PTK_HD PrimRec prim(const SceneView& sc, uint32_t off) const
  {
    return *reinterpret_cast<const PrimRec*>(reinterpret_cast<const char*>(sc.prims) + off);
  }
  // shading-side tables: by reference, fields are fetched where they are used
  PTK_HD const ShadeRec& shade(const SceneView& sc, uint32_t id) const { return sc.shade[id]; }
  PTK_HD const MaterialRec& material(const SceneView& sc, uint32_t i) const { return sc.materials[i]; }
  PTK_HD const LightRec& light(const SceneView& sc, int i) const { return sc.lights[i]; }
};
#if defined(__HIPCC__)
#if defined(__HIP_DEVICE_COMPILE__)
#define PTK_LDS_SPACE __attribute__((address_space(3)))  // the loads must be ds_read, not flat loads
#else
#define PTK_LDS_SPACE  // the host pass only has to parse the kernels
#endif
struct LdsMem {
  static constexpr bool kPrefetch = false;
  const PTK_LDS_SPACE char* nodes;
  const PTK_LDS_SPACE char* prims;
  const PTK_LDS_SPACE ShadeRec* shades;
  const PTK_LDS_SPACE MaterialRec* materials;
  const PTK_LDS_SPACE LightRec* lights;
  PTK_HD NodePair node(const SceneView&, uint32_t off) const { return *reinterpret_cast<const PTK_LDS_SPACE NodePair*>(nodes + off); }
  PTK_HD PrimRec prim(const SceneView&, uint32_t off) const { return *reinterpret_cast<const PTK_LDS_SPACE PrimRec*>(prims + off); }
  PTK_HD const PTK_LDS_SPACE ShadeRec& shade(const SceneView&, uint32_t id) const { return shades[id]; }
  PTK_HD const PTK_LDS_SPACE MaterialRec& material(const SceneView&, uint32_t i) const { return materials[i]; }
  PTK_HD const PTK_LDS_SPACE LightRec& light(const SceneView&, int i) const { return lights[i]; }
};
// Nodes, primitive and shading records from global memory; the material and light tables (a few records that
// every path vertex reads) from the workgroup's copy in LDS.
struct TablesMem : GlobalMem {
  const PTK_LDS_SPACE MaterialRec* materials;
  const PTK_LDS_SPACE LightRec* lights;
  PTK_HD const PTK_LDS_SPACE MaterialRec& material(const SceneView&, uint32_t i) const { return materials[i]; }
  PTK_HD const PTK_LDS_SPACE LightRec& light(const SceneView&, int i) const { return lights[i]; }
};
#endif

// Visit one inner node.  Straight-line code: whether a child is pushed, and whether the next node
// comes from the stack, are decided by selects, and the stack is touched the same way every time
// (Stack::step reads the top entry, writes the candidate above it and then moves the stack pointer).
template <class Stack, bool STATS, class Mem = GlobalMem>
PTK_HD void trav_inner(Traversal& tr, Stack& st, const SceneView& sc, Counters& cnt, const Mem& mem = Mem())
{
  // a 32-bit byte offset from the (wave-uniform) array base: one shift instead of 64-bit address arithmetic
  const uint32_t node_off = (uint32_t)tr.node * (uint32_t)sizeof(NodePair);
  const uint32_t stack_top = st.prefetch();  // what step() below hands back, asked for before the node itself
  const NodePair nd = mem.node(sc, node_off);
  struct { float x, y, z, w; } bx, by, bz;
  bx.x = (float)(nd.x0 & 0xffffu); bx.y = grid_extent(nd.x0); bx.z = (float)(nd.x1 & 0xffffu); bx.w = grid_extent(nd.x1);
  by.x = (float)(nd.y0 & 0xffffu); by.y = grid_extent(nd.y0); by.z = (float)(nd.y1 & 0xffffu); by.w = grid_extent(nd.y1);
  bz.x = (float)(nd.z0 & 0xffffu); bz.y = grid_extent(nd.z0); bz.z = (float)(nd.z1 & 0xffffu); bz.w = grid_extent(nd.z1);
  const int32_t child0 = nd.child0, child1 = nd.child1;
#ifdef PTK_EXTRA_LOAD  // experiment: one more 16-byte cache access per node visit, result unused
  {
    // 1: at the lane's own node; 2: lanes 2k and 2k+1 read the two adjacent halves of lane 2k's node
    uint32_t off = node_off;
#if PTK_EXTRA_LOAD == 2 && defined(__HIP_DEVICE_COMPILE__)
    const uint32_t partner = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)off, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
    const bool odd = (threadIdx.x & 1u) != 0;
    off = odd ? partner + 16u : off;
#endif
    const volatile f4* extra = reinterpret_cast<const volatile f4*>(reinterpret_cast<const char*>(sc.nodes) + off);
    float x = extra->x, y = extra->w;
    asm volatile("" ::"v"(x), "v"(y));
  }
#endif
#ifdef PTK_BURN_INNER
  for (int burn = 0; burn < PTK_BURN_INNER; burn++) PTK_BURN12(tr.node);
#endif
#ifdef PTK_BURN_INNER_FMA  // same with dependent fused multiply-adds on a scratch value
  {
    float scratch = tr.hit.t;
    for (int burn = 0; burn < PTK_BURN_INNER_FMA; burn++)
      asm volatile("v_fma_f32 %0, %0, %0, %0\nv_fma_f32 %0, %0, %0, %0\nv_fma_f32 %0, %0, %0, %0\nv_fma_f32 %0, %0, %0, %0\n"
                   "v_fma_f32 %0, %0, %0, %0\nv_fma_f32 %0, %0, %0, %0\nv_fma_f32 %0, %0, %0, %0\nv_fma_f32 %0, %0, %0, %0\n"
                   "v_fma_f32 %0, %0, %0, %0\nv_fma_f32 %0, %0, %0, %0\nv_fma_f32 %0, %0, %0, %0\nv_fma_f32 %0, %0, %0, %0"
                   : "+v"(scratch));
  }
#endif
  float tn0, tn1;
  const RaySlabs rs = ray_slabs(tr.inv, tr.oi);  // does not change between a query's visits
  bool h0 = hit_box(bx.x, bx.y, by.x, by.y, bz.x, bz.y, rs, tr.hit.t, tn0);
  bool h1 = hit_box(bx.z, bx.w, by.z, by.w, bz.z, bz.w, rs, tr.hit.t, tn1);
  if (STATS) cnt.box_tests += 2;
  // logical operators on purpose: the results stay lane masks (scalar and/or) instead of 0/1 integers
  const bool both = h0 && h1, none = !h0 && !h1;
  const bool closer1 = tn1 < tn0;
  const bool first1 = h1 && (!h0 || closer1);  // child 1 is the one to visit first
  const int32_t near = first1 ? child1 : child0, far = first1 ? child0 : child1;
  const uint32_t top = st.step(both, (uint32_t)far, none, stack_top);
  tr.node = none ? (int32_t)top : near;
}

// Test one primitive record against the lane's ray and keep it if it is the closest hit so far.  MAY_BE_SPHERE:
// the record's kind is looked at; otherwise the caller knows it is a triangle (leaf_has_spheres).
template <bool STATS, bool MAY_BE_SPHERE = true>
PTK_HD void trav_test_prim(Traversal& tr, const PrimRec& p, Counters& cnt)
{
  // id and kind share a word (PrimRec), which makes the record's third load 8 bytes instead of 12
  const uint32_t idw = as_uint(p.c.y), id = idw & kPrimIdMask;
  if (STATS) cnt.prim_tests++;
#ifdef PTK_BURN_PRIM
  for (int burn = 0; burn < PTK_BURN_PRIM; burn++) PTK_BURN12(tr.node);
#endif
  float t = 0.0f, u = 0.0f, v = 0.0f;
  bool ok;
  if (!MAY_BE_SPHERE || (idw & kPrimSphereBit) == 0u) {
    // a triangle the ray starts on is not a candidate (see the note at the top of this file)
    ok = hit_triangle(p, tr.o, tr.d, tr.hit.t, t, u, v) & (id != tr.skip_id);
  } else {
    ok = hit_sphere(p, tr.o, tr.d, tr.hit.t, id == tr.skip_id, t);
    u = v = 0.0f;
  }
  if (ok) { tr.hit.t = t; tr.hit.id = id; tr.hit.u = u; tr.hit.v = v; }
}

// The records of one leaf, one after the other.
template <bool STATS, bool MAY_BE_SPHERE, class Mem>
PTK_HD void trav_leaf_records(Traversal& tr, uint32_t first, uint32_t count, const SceneView& sc, Counters& cnt, const Mem& mem)
{
  uint32_t off = first * (uint32_t)sizeof(PrimRec);
  if (!Mem::kPrefetch) {  // records in LDS arrive quickly: fetch, then test (+1 % on the Cornell box; from global memory -8 %)
    for (uint32_t i = 0; i < count; i++, off += (uint32_t)sizeof(PrimRec)) {
      const PrimRec p = mem.prim(sc, off);
      trav_test_prim<STATS, MAY_BE_SPHERE>(tr, p, cnt);
    }
    return;
  }
  // The next record is fetched while this one is tested.  Two records take turns (the loop is unrolled by two)
  // so that "next becomes current" is not ten register moves per primitive.
  PrimRec p0 = mem.prim(sc, off), p1 = p0;
  for (uint32_t i = 0;; i += 2) {
#if defined(PTK_PHASE_STATS) && defined(__HIP_DEVICE_COMPILE__)
    {
      const unsigned long long active = __ballot(true);
      if (__builtin_amdgcn_mbcnt_hi((uint32_t)(active >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)active, 0u)) == 0u) cnt.leaf_iterations++;
    }
#endif
    const bool second = i + 1 < count;
    if (second) p1 = mem.prim(sc, off + (uint32_t)sizeof(PrimRec));
    trav_test_prim<STATS, MAY_BE_SPHERE>(tr, p0, cnt);
    if (!second) break;
    const bool third = i + 2 < count;
    off += 2u * (uint32_t)sizeof(PrimRec);
    if (third) p0 = mem.prim(sc, off);
#if defined(PTK_PHASE_STATS) && defined(__HIP_DEVICE_COMPILE__)
    {
      const unsigned long long active = __ballot(true);
      if (__builtin_amdgcn_mbcnt_hi((uint32_t)(active >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)active, 0u)) == 0u) cnt.leaf_iterations++;
    }
#endif
    trav_test_prim<STATS, MAY_BE_SPHERE>(tr, p1, cnt);
    if (!third) break;
  }
}

// Intersect the primitives of the leaf with code `node` (a negative child link).  Returns true when
// the query is decided and needs no further traversal (see Traversal::t_stop).
template <bool STATS, class Mem = GlobalMem>
PTK_HD bool trav_leaf_prims(Traversal& tr, int32_t node, const SceneView& sc, Counters& cnt, const Mem& mem = Mem())
{
  const uint32_t first = leaf_first(node), count = leaf_count(node);
  if (leaf_has_spheres(node)) trav_leaf_records<STATS, true, Mem>(tr, first, count, sc, cnt, mem);  // (rare: skipped by whole waves)
  else trav_leaf_records<STATS, false, Mem>(tr, first, count, sc, cnt, mem);
  return (tr.hit.t < tr.t_stop) & (tr.hit.id != tr.target);
}

// Intersect the leaf the lane is at, then go on with the node on top of the stack.
template <class Stack, bool STATS, class Mem = GlobalMem>
PTK_HD void trav_leaf(Traversal& tr, Stack& st, const SceneView& sc, Counters& cnt, const Mem& mem = Mem())
{
  const bool decided = trav_leaf_prims<STATS, Mem>(tr, tr.node, sc, cnt, mem);
  const int32_t next_node = (int32_t)st.pop();
  tr.node = decided ? kTravDone : next_node;
}

// One whole query on one lane (used where rays are independent: the ray-batch entry point).
template <class Stack, bool STATS>
PTK_HD bool closest_hit(const SceneView& sc, v3 o, v3 d, float tmax, uint32_t skip_id, Stack& st, Hit& hit,
                        Counters& cnt)
{
  Traversal tr;
  const float shift = scene_entry_shift(sc, o, d);
  // distances are measured from the moved origin until the shift is added back below: a finite tmax moves with it
  trav_begin(tr, st, sc, o + d * shift, d, tmax < kTMax ? tmax - shift : tmax, skip_id);
  while (!trav_done(tr)) {
    if (trav_at_inner(tr)) trav_inner<Stack, STATS>(tr, st, sc, cnt);
    else trav_leaf<Stack, STATS>(tr, st, sc, cnt);
  }
  hit = tr.hit;
  hit.t += shift;
  return hit.id != kNoPrim;
}

// The Intersection record of the reference (src/geometry.h:14-24) for a closest hit,
// filled the way Primitive::intersect does (src/geometry.cpp:166-219).
struct Surface {
  v3 point, normal;   // normal: interpolated vertex normals, NOT normalised (src/geometry.cpp:100-105)
  float uv[2];
  uint32_t id, material;
  bool inside;
};

PTK_HD float atan2_ref(float y, float x) { return atan2f(y, x); }

template <class Mem = GlobalMem>
PTK_HD void make_surface(const SceneView& sc, v3 o, v3 d, const Hit& hit, Surface& s, const Mem& mem = Mem())
{
  const auto& sh = mem.shade(sc, hit.id);
  const auto& m = mem.material(sc, sh.material);
  s.id = hit.id;
  s.material = sh.material;
  s.point = o + d * hit.t;  // Ray::point_at, src/ray.h:8
  s.uv[0] = s.uv[1] = 0.0f;
  v3 normal;
  bool keep_normal = false;
  if (hit.id < (uint32_t)sc.n_triangles) {
    float w0 = 1.0f - hit.u - hit.v;
    normal = V3(sh.n0[0], sh.n0[1], sh.n0[2]) * w0 + V3(sh.n1[0], sh.n1[1], sh.n1[2]) * hit.u +
             V3(sh.n2[0], sh.n2[1], sh.n2[2]) * hit.v;
    keep_normal = (m.flags & kMatLight) != 0;  // lights keep their stored normal, src/geometry.cpp:199-201
    if (m.texture >= 0) {  // Triangle::texcoord, src/geometry.cpp:86-91
      s.uv[0] = w0 * sh.t0[0] + hit.u * sh.t1[0] + hit.v * sh.t2[0];
      s.uv[1] = w0 * sh.t0[1] + hit.u * sh.t1[1] + hit.v * sh.t2[1];
    }
  } else {
    v3 c = V3(sh.n0[0], sh.n0[1], sh.n0[2]);
    normal = (s.point - c) * (1.0f / sh.n1[0]);
    if (m.texture >= 0) {  // Sphere::texcoord, src/geometry.cpp:56-64
      v3 q = normalize(s.point - c);
      const float pi = 3.14159265359f;
      float theta = acosf(-q.y);
      float phi = atan2_ref(-q.z, q.x) + pi;
      s.uv[0] = phi / (2.0f * pi);
      s.uv[1] = theta / pi;
    }
  }
  if (keep_normal) { s.normal = normal; s.inside = true; }
  else if (dot(d, normal) > 0.0f) { s.normal = -normal; s.inside = false; }
  else { s.normal = normal; s.inside = true; }
}

}  // namespace ptk
