// TEST HARNESS ONLY -- never part of libptk.so and never loaded by the package.
//
// Compiles the kernels' per-lane device functions (ptk_traverse.h, ptk_shade.h) and the host-side
// scene flattening (ptk_build.cpp) for the CPU, and runs them lane by lane.  This lets the CPU-only
// test suite check the fp32 kernel logic against the oracle in a container without a GPU; the GPU
// tests then check the real kernels through the C ABI.
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ptk.h"
#include "../../pathtracer_amd/csrc/ptk_build.h"
#include "../../pathtracer_amd/csrc/ptk_shade.h"

using namespace ptk;

namespace {
struct HostStack {
  uint32_t data[kStackDepth];
  int sp = 0;
  int high_water = 0;
  void reset() { data[0] = (uint32_t)kTravDone; sp = 1; }
  void push(uint32_t v)
  {
    if (sp < kStackDepth) data[sp++] = v;
    if (sp > high_water) high_water = sp;
  }
  uint32_t pop() { return data[--sp]; }
  uint32_t prefetch() const { return data[sp - 1]; }
  uint32_t step(bool do_push, uint32_t v, bool do_pop, uint32_t top)
  {
    if (do_push) push(v);
    if (do_pop) sp--;
    return top;
  }
};

struct Check {
  HostScene host;
  std::string err;
};
}  // namespace

extern "C" {

void* hc_create(const ptk_scene_desc* d)
{
  Check* c = new Check();
  c->err = build_host_scene(*d, c->host);
  return c;
}
const char* hc_error(void* h) { return ((Check*)h)->err.c_str(); }
void hc_destroy(void* h) { delete (Check*)h; }

void hc_info(void* h, int32_t* out /* n_nodes, max_depth, max_leaf, n_lights, n_prims, tree_from_cache */)
{
  Check* c = (Check*)h;
  out[0] = (int32_t)c->host.nodes.size(); out[1] = c->host.max_depth; out[2] = c->host.max_leaf;
  out[3] = (int32_t)c->host.lights.size(); out[4] = c->host.n_prims;
  out[5] = c->host.tree_from_cache ? 1 : 0;
}

// Walks the flattened tree and checks, in double precision, that every child's grid box contains all the
// primitives below it and that every primitive is below exactly one leaf.
// out: violations (a primitive sticking out of a box on its path), primitives reached, deepest leaf level.
void hc_tree_check(void* h, int64_t* out)
{
  Check* c = (Check*)h;
  const HostScene& hs = c->host;
  const BoxGrid g = hs.grid;
  struct Todo { int32_t link; int depth; double lo[3], hi[3]; };
  std::vector<Todo> stack;
  int64_t violations = 0, reached = 0, deepest = 0;
  auto decode = [&](uint32_t wx, uint32_t wy, uint32_t wz, Todo& t) {
    const uint32_t w[3] = {wx, wy, wz};
    for (int a = 0; a < 3; a++) {
      t.lo[a] = (double)g.origin[a] + (double)(w[a] & 0xffffu) * (double)g.scale[a];
      t.hi[a] = (double)g.origin[a] + ((double)(w[a] & 0xffffu) + (double)grid_extent_steps(w[a])) * (double)g.scale[a];  // low plane + extent (what the traversal sees is not below it)
    }
  };
  const NodePair& root = hs.nodes[0];
  Todo a, b;
  a.link = root.child0; a.depth = 1; decode(root.x0, root.y0, root.z0, a);
  b.link = root.child1; b.depth = 1; decode(root.x1, root.y1, root.z1, b);
  stack.push_back(a);
  if (!(root.child1 == root.child0 && root.child0 < 0)) stack.push_back(b);  // a one-leaf scene repeats its leaf
  while (!stack.empty()) {
    Todo t = stack.back(); stack.pop_back();
    if (t.depth > deepest) deepest = t.depth;
    if (t.link >= 0) {
      const NodePair& n = hs.nodes[t.link];
      Todo c0, c1;
      c0.link = n.child0; c0.depth = t.depth + 1; decode(n.x0, n.y0, n.z0, c0);
      c1.link = n.child1; c1.depth = t.depth + 1; decode(n.x1, n.y1, n.z1, c1);
      // boxes are rounded independently, so a child's box may stick out of its parent's by a grid step; what
      // the traversal relies on is that EVERY box on the path contains the primitive: carry their intersection
      for (int k = 0; k < 3; k++) {
        c0.lo[k] = std::max(c0.lo[k], t.lo[k]); c0.hi[k] = std::min(c0.hi[k], t.hi[k]);
        c1.lo[k] = std::max(c1.lo[k], t.lo[k]); c1.hi[k] = std::min(c1.hi[k], t.hi[k]);
      }
      stack.push_back(c0); stack.push_back(c1);
      continue;
    }
    const uint32_t first = leaf_first(t.link), count = leaf_count(t.link);
    for (uint32_t i = first; i < first + count && i < hs.prims.size(); i++) {
      const PrimRec& p = hs.prims[i];
      double lo[3], hi[3];
      if ((as_uint(p.c.y) & kPrimSphereBit) == 0u) {
        const double v0[3] = {p.a.x, p.a.y, p.a.z}, e1[3] = {p.a.w, p.b.x, p.b.y}, e2[3] = {p.b.z, p.b.w, p.c.x};
        for (int k = 0; k < 3; k++) {
          lo[k] = std::min(v0[k], std::min(v0[k] + e1[k], v0[k] + e2[k]));
          hi[k] = std::max(v0[k], std::max(v0[k] + e1[k], v0[k] + e2[k]));
        }
      } else {
        const double ctr[3] = {p.a.x, p.a.y, p.a.z};
        for (int k = 0; k < 3; k++) { lo[k] = ctr[k] - std::fabs((double)p.a.w); hi[k] = ctr[k] + std::fabs((double)p.a.w); }
      }
      reached++;
      for (int k = 0; k < 3; k++)
        if (lo[k] < t.lo[k] || hi[k] > t.hi[k]) { violations++; break; }
    }
  }
  out[0] = violations; out[1] = reached; out[2] = deepest;
}

// returns the deepest stack use seen
int hc_intersect(void* h, const ptk_ray* rays, int n, ptk_hit* hits, uint64_t* counters /* box, prim */)
{
  Check* c = (Check*)h;
  SceneView sc = c->host.host_view();
  int high = 0;
  Counters cnt = {0, 0, 0, 0, 0};
  uint64_t box = 0, prim = 0;
  for (int i = 0; i < n; i++) {
    HostStack st;
    v3 o = V3((float)rays[i].origin[0], (float)rays[i].origin[1], (float)rays[i].origin[2]);
    v3 d = V3((float)rays[i].direction[0], (float)rays[i].direction[1], (float)rays[i].direction[2]);
    Hit hit;
    ptk_hit out;
    std::memset(&out, 0, sizeof out);
    cnt.box_tests = cnt.prim_tests = 0;
    if (closest_hit<HostStack, true>(sc, o, d, kTMax, kNoPrim, st, hit, cnt)) {
      Surface s;
      // the hit point is formed from the origin the traversal used (moved towards a far-away scene), as the kernel forms it
      const float shift = scene_entry_shift(sc, o, d);
      Hit moved = hit;
      moved.t = hit.t - shift;
      make_surface(sc, o + d * shift, d, moved, s);
      out.hit = 1; out.id = hit.id; out.t = hit.t;
      out.point[0] = s.point.x; out.point[1] = s.point.y; out.point[2] = s.point.z;
      out.normal[0] = s.normal.x; out.normal[1] = s.normal.y; out.normal[2] = s.normal.z;
      out.uv[0] = s.uv[0]; out.uv[1] = s.uv[1];
      out.material = (int32_t)s.material; out.inside = s.inside ? 1 : 0;
    }
    box += cnt.box_tests; prim += cnt.prim_tests;
    hits[i] = out;
    if (st.high_water > high) high = st.high_water;
  }
  if (counters) { counters[0] = box; counters[1] = prim; }
  return high;
}

// Node visits of a ray batch by tree level (root = level 0): hist[level] += 1 per inner-node visit.
// Returns the number of levels filled.  A measurement aid (how much of the tree's top is worth keeping in
// faster memory: on the benchmark scenes 49 % of all visits go to the top four levels).
int hc_visit_histogram(void* h, const ptk_ray* rays, int n, uint64_t* hist, int max_levels)
{
  Check* c = (Check*)h;
  const HostScene& hs = c->host;
  std::vector<int> depth(hs.nodes.size(), -1);
  {
    std::vector<std::pair<int32_t, int>> todo{{0, 0}};
    while (!todo.empty()) {
      auto [ni, d] = todo.back(); todo.pop_back();
      depth[ni] = d;
      if (hs.nodes[ni].child0 >= 0) todo.push_back({hs.nodes[ni].child0, d + 1});
      if (hs.nodes[ni].child1 >= 0) todo.push_back({hs.nodes[ni].child1, d + 1});
    }
  }
  SceneView sc = hs.host_view();
  Counters cnt = {0, 0, 0, 0, 0};
  int top = 0;
  for (int i = 0; i < n; i++) {
    HostStack st;
    v3 o = V3((float)rays[i].origin[0], (float)rays[i].origin[1], (float)rays[i].origin[2]);
    v3 d = V3((float)rays[i].direction[0], (float)rays[i].direction[1], (float)rays[i].direction[2]);
    Traversal tr;
    trav_begin(tr, st, sc, o, d, kTMax, kNoPrim);
    while (!trav_done(tr)) {
      if (trav_at_inner(tr)) {
        int lv = depth[tr.node];
        if (lv < max_levels) { hist[lv]++; if (lv + 1 > top) top = lv + 1; }
        trav_inner<HostStack, false>(tr, st, sc, cnt);
      } else {
        trav_leaf<HostStack, false>(tr, st, sc, cnt);
      }
    }
  }
  return top;
}

// accum: W*H*4 floats (sum r,g,b, count), same meaning as the device accumulation buffer
void hc_render(void* h, const ptk_camera* camera, const ptk_flags* flags, const ptk_render_params* p, float* accum,
               uint64_t* counters /* camera, segments, queries, box, prim */)
{
  Check* c = (Check*)h;
  SceneView sc = c->host.host_view();
  CameraView cam = make_camera_view(*camera);
  FlagView fl = make_flag_view(flags);
  int row_end = p->row_end <= 0 ? cam.height : p->row_end;
  uint64_t tot[5] = {0, 0, 0, 0, 0};
#pragma omp parallel
  {
    uint64_t loc[5] = {0, 0, 0, 0, 0};
#pragma omp for schedule(dynamic, 1)
    for (int y = p->row_begin; y < row_end; y++) {
      if (p->band_rows > 0) {  // interleaved bands: only those of this call
        int b = y / p->band_rows - p->band_first;
        if (b < 0 || b % p->band_stride != 0) continue;
      }
      for (int x = 0; x < cam.width; x++) {
        HostStack st;
        Counters cnt = {0, 0, 0, 0, 0};
        v3 sum = render_pixel<HostStack, true>(sc, cam, fl, x, y, p->samples, p->max_bounce, p->sample_first,
                                               p->sample_stride, p->seed, st, cnt);
        float* a = accum + 4 * ((size_t)y * cam.width + x);
        a[0] += sum.x; a[1] += sum.y; a[2] += sum.z; a[3] += (float)p->samples;
        loc[0] += cnt.camera_rays; loc[1] += cnt.segments; loc[2] += cnt.queries; loc[3] += cnt.box_tests; loc[4] += cnt.prim_tests;
      }
    }
#pragma omp critical
    for (int k = 0; k < 5; k++) tot[k] += loc[k];
  }
  if (counters) for (int k = 0; k < 5; k++) counters[k] = tot[k];
}

}  // extern "C"
