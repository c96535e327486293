// Host-side scene flattening and BVH construction (see ptk_build.h).
#include "ptk_build.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>
#include <unistd.h>

namespace ptk {

namespace {

struct Box {
  float lo[3], hi[3];
  void reset()
  {
    for (int a = 0; a < 3; a++) { lo[a] = std::numeric_limits<float>::max(); hi[a] = -std::numeric_limits<float>::max(); }
  }
  void grow(const Box& b)
  {
    for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); }
  }
  void grow(const float p[3])
  {
    for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); }
  }
  float half_area() const
  {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx < 0.f || dy < 0.f || dz < 0.f) return 0.f;
    return dx * dy + dy * dz + dz * dx;
  }
};

struct BuildNode {
  Box box;
  int32_t left = -1, right = -1;  // children (build nodes) or -1
  int32_t first = 0, count = 0;   // range in the index array (leaves)
};

// Levels of median splitting needed until every part holds <= kMaxLeafSize primitives.
int median_levels(int n)
{
  int l = 0;
  while (n > kMaxLeafSize) { n = (n + 1) / 2; l++; }
  return l;
}

#ifndef PTK_SAH_BINS
#define PTK_SAH_BINS 16
#endif
constexpr int kBins = PTK_SAH_BINS;
constexpr float kTraversalCost = 1.0f;
#ifndef PTK_SAH_INTERSECT_COST
#define PTK_SAH_INTERSECT_COST 1.2f
#endif
constexpr float kIntersectCost = PTK_SAH_INTERSECT_COST;
// Largest leaf the surface-area heuristic may choose (the depth guard's median splits can still leave kMaxLeafSize).
#ifndef PTK_SAH_LEAF_MAX
#define PTK_SAH_LEAF_MAX kMaxLeafSize
#endif
constexpr int kSahLeafMax = PTK_SAH_LEAF_MAX;

// Binned surface-area-heuristic builder.  Subtrees are independent, so the two sides of a large
// split are built by different threads; which thread builds what does not influence the tree (every
// decision depends only on the primitives of the range), and the flattening below numbers nodes by
// a depth-first walk, so the result is identical for any thread count.
struct Builder {
  // A primitive as the builder moves it around: partitioning reorders these records themselves, so
  // every pass over a range streams through memory instead of chasing indices.
  struct Item {
    Box box;
    float centroid[3];
    int32_t id;
  };
  std::vector<Item> items;
  std::vector<BuildNode> nodes;  // sized for the worst case up front so that threads never reallocate it
  std::atomic<int32_t> n_nodes{0};
  std::atomic<int> max_depth{0}, max_leaf{0};
  std::atomic<bool> out_of_slots{false};
  static constexpr int kParallelMin = 16384;  // smaller sides are not worth a thread
  static constexpr int kMaxThreads = 64;

  // Slot budget: every range that one thread builds alone takes a block of 2 * count slots (its subtree has at
  // most 2 * count - 1 nodes) and those ranges are disjoint, so the blocks sum to at most 2 * n.  Nodes above
  // the blocks (built while the range can still be shared between threads) take one slot each: they form a
  // tree with at most `threads` ends whose root paths are at most kMaxTreeDepth + 1 long.
  Builder(const std::vector<Box>& b, int threads)
  {
    size_t n = b.size();
    items.resize(n);
    for (size_t i = 0; i < n; i++) {
      items[i].box = b[i];
      items[i].id = (int32_t)i;
      for (int a = 0; a < 3; a++) items[i].centroid[a] = 0.5f * (b[i].lo[a] + b[i].hi[a]);
    }
    nodes.resize(2 * n + (size_t)std::max(threads, 1) * (kMaxTreeDepth + 2) + 64);
  }

  static void raise(std::atomic<int>& m, int v)
  {
    int cur = m.load(std::memory_order_relaxed);
    while (cur < v && !m.compare_exchange_weak(cur, v, std::memory_order_relaxed)) {}
  }

  static int bin_of(const Item& it, int axis, float lo, float scale)
  {
    return std::min(kBins - 1, std::max(0, (int)((it.centroid[axis] - lo) * scale)));
  }

  void finish() { nodes.resize(std::min(nodes.size(), (size_t)n_nodes.load())); }  // slots are not dense: walk by links

  // Node slots of a subtree built by one thread: a contiguous block taken from the shared counter once,
  // so that threads do not write to each other's cache lines.
  struct Block {
    int32_t next, end;
  };

  int32_t build(int first, int count, int depth, int threads, Block* block = nullptr)
  {
    Block own;
    // From here down this thread works alone (a split is shared between threads only when both sides hold
    // kParallelMin primitives, which a range below twice that cannot give): at most 2 * count - 1 nodes.
    if (!block && (threads <= 1 || count < 2 * kParallelMin)) {
      own.next = n_nodes.fetch_add(2 * count);
      own.end = own.next + 2 * count;
      block = &own;
    }
    int32_t ni = block ? block->next++ : n_nodes.fetch_add(1);
    if (ni < 0 || (size_t)ni >= nodes.size() || (block && ni >= block->end)) {  // never write outside the array
      out_of_slots.store(true);
      return 0;
    }
    Box box, cbox;
    box.reset(); cbox.reset();
    for (int i = first; i < first + count; i++) {
      box.grow(items[i].box);
      cbox.grow(items[i].centroid);
    }
    nodes[ni].box = box;
    nodes[ni].first = first;
    nodes[ni].count = count;
    raise(max_depth, depth);

    // --- choose a split: one pass bins the range along all three axes -----------
    int best_axis = -1, best_bin = -1;
    float best_cost = std::numeric_limits<float>::max();
    float parent_area = box.half_area();
    if (count > 1 && parent_area > 0.f) {
      Box bin_box[3][kBins];
      int bin_count[3][kBins];
      float lo[3], scale[3];
      bool use[3];
      for (int axis = 0; axis < 3; axis++) {
        float ext = cbox.hi[axis] - cbox.lo[axis];
        lo[axis] = cbox.lo[axis];
        use[axis] = ext > 0.f;
        scale[axis] = use[axis] ? (float)kBins / ext : 0.f;
        for (int b = 0; b < kBins; b++) { bin_box[axis][b].reset(); bin_count[axis][b] = 0; }
      }
      for (int i = first; i < first + count; i++) {
        const Item& it = items[i];
        for (int axis = 0; axis < 3; axis++) {
          if (!use[axis]) continue;
          int b = bin_of(it, axis, lo[axis], scale[axis]);
          bin_box[axis][b].grow(it.box);
          bin_count[axis][b]++;
        }
      }
      for (int axis = 0; axis < 3; axis++) {
        if (!use[axis]) continue;
        float right_area[kBins];
        int right_count[kBins];
        Box acc;
        acc.reset();
        int c = 0;
        for (int b = kBins - 1; b > 0; b--) {
          acc.grow(bin_box[axis][b]); c += bin_count[axis][b];
          right_area[b] = acc.half_area(); right_count[b] = c;
        }
        acc.reset();
        c = 0;
        for (int b = 0; b < kBins - 1; b++) {
          acc.grow(bin_box[axis][b]); c += bin_count[axis][b];
          if (c == 0 || right_count[b + 1] == 0) continue;
          float cost = kTraversalCost +
                       kIntersectCost * (acc.half_area() * (float)c + right_area[b + 1] * (float)right_count[b + 1]) /
                           parent_area;
          if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
        }
      }
    }
    bool must_split = count > kSahLeafMax;
    bool want_split = best_axis >= 0 && best_cost < kIntersectCost * (float)count;
    if (!must_split && (!want_split || depth + 1 > kMaxTreeDepth)) {
      raise(max_leaf, count);
      return ni;
    }

    int mid = -1;
    if (best_axis >= 0) {
      float lo = cbox.lo[best_axis], scale = (float)kBins / (cbox.hi[best_axis] - cbox.lo[best_axis]);
      auto* begin = items.data() + first;
      auto* m = std::partition(begin, begin + count, [&](const Item& it) { return bin_of(it, best_axis, lo, scale) <= best_bin; });
      mid = first + (int)(m - begin);
      int big = std::max(mid - first, first + count - mid);
      // Depth guard: an unbalanced split is only taken if the larger side can still be finished
      // by median splits within the depth budget.
      if (mid == first || mid == first + count || depth + 1 + median_levels(big) > kMaxTreeDepth) mid = -1;
    }
    if (mid < 0) {  // object median along the widest centroid axis
      int axis = 0;
      float e0 = cbox.hi[0] - cbox.lo[0], e1 = cbox.hi[1] - cbox.lo[1], e2 = cbox.hi[2] - cbox.lo[2];
      if (e1 > e0 && e1 >= e2) axis = 1;
      else if (e2 > e0 && e2 > e1) axis = 2;
      mid = first + (count + 1) / 2;
      auto* begin = items.data() + first;
      std::nth_element(begin, begin + (mid - first), begin + count, [&](const Item& a, const Item& b) {
        float ca = a.centroid[axis], cb = b.centroid[axis];
        return ca < cb || (ca == cb && a.id < b.id);
      });
    }
    const int nl = mid - first, nr = first + count - mid;
    int32_t l, r;
    if (threads > 1 && std::min(nl, nr) >= kParallelMin) {
      int tl = (int)((int64_t)threads * nl / count);
      tl = std::min(std::max(tl, 1), threads - 1);
      std::thread left([&, tl] { l = build(first, nl, depth + 1, tl); });
      r = build(mid, nr, depth + 1, threads - tl);
      left.join();
    } else {
      // a lopsided split keeps the whole thread budget on its large side
      l = build(first, nl, depth + 1, nl >= nr ? threads : 1, block);
      r = build(mid, nr, depth + 1, nr > nl ? threads : 1, block);
    }
    nodes[ni].left = l;
    nodes[ni].right = r;
    return ni;
  }
};

// Tree rotations after the build (bottom-up passes, until one changes nothing): a child of a node trades places with a
// grandchild on the other side when the inner child left behind gets a smaller box - its surface area is the chance
// that a ray has to visit it.  The trade pushes one subtree a level down: only where the tree's depth allows it.  A
// few hundred trades in a scene of 82 000 triangles (the wall-sized triangles of the box and the subtrees of the mesh
// that the top-down splits had put beside them), worth 6 % of the frame rate.  PTK_TREE_ROTATIONS=n in the environment
// sets the number of passes (0: none).
constexpr int kRotationPasses = 8;
struct Rotations {
  std::vector<BuildNode>& n;
  std::vector<int8_t> height;
  int limit;     // deepest level an inner node may have + 1
  long applied = 0;
  static Box both(const Box& a, const Box& b) { Box u = a; u.grow(b); return u; }
  int pass(int32_t i, int depth)
  {
    BuildNode& nd = n[i];
    if (nd.left < 0) return height[i] = 0;
    pass(nd.left, depth + 1);
    pass(nd.right, depth + 1);
    float best = 0.f;
    int best_side = -1, best_g = -1;
    for (int side = 0; side < 2; side++) {
      const int32_t c = side ? nd.right : nd.left, o = side ? nd.left : nd.right;  // c keeps one grandchild and takes o
      if (n[c].left < 0) continue;
      if (depth + 2 + height[o] > limit) continue;
      for (int g = 0; g < 2; g++) {  // g: the grandchild that moves up
        const int32_t stays = g ? n[c].left : n[c].right;
        const float gain = n[c].box.half_area() - both(n[o].box, n[stays].box).half_area();
        if (gain > best) { best = gain; best_side = side; best_g = g; }
      }
    }
    if (best_side >= 0) {
      const int32_t c = best_side ? nd.right : nd.left, o = best_side ? nd.left : nd.right;
      const int32_t up = best_g ? n[c].right : n[c].left, stays = best_g ? n[c].left : n[c].right;
      n[c].left = o; n[c].right = stays;
      n[c].box = both(n[o].box, n[stays].box);
      height[c] = (int8_t)(1 + std::max(height[o], height[stays]));
      if (best_side) nd.left = up; else nd.right = up;
      applied++;
    }
    return height[i] = (int8_t)(1 + std::max(height[nd.left], height[nd.right]));
  }
};

// The best top of the tree.  The frontier of the kTopOrderSubtrees subtrees with the largest boxes is taken (start at
// the root, always open the node with the largest surface area) and EVERY binary tree over them is priced by the
// surface area of its inner nodes - dynamic programming over the subsets, 3^12 steps; the cheapest replaces what the
// top-down splits and the rotations left.  On the mesh scenes that is the same top (the rotations had found it); on
// the scene of glass and mirror spheres in front of a mesh it saves 3.5 % of the box tests (+1.3 % frame rate), and
// it takes out the dependence of the result on the top-down build's bin count.  PTK_TREE_TOP_ORDER=0 skips it.  (On the
// twelve-triangle Cornell box, where the frontier is the whole tree, the same surface area came out as a tree that
// renders 5 % slower: hence kTopOrderMin.)
constexpr int kTopOrderSubtrees = 12;
constexpr int kTopOrderMin = 256;  // primitives: below that the frontier is most of the tree, whose leaves this pass knows nothing about
int tree_levels(const std::vector<BuildNode>& n, int32_t i)  // levels of inner nodes from i down
{
  return n[i].left < 0 ? 0 : 1 + std::max(tree_levels(n, n[i].left), tree_levels(n, n[i].right));
}
struct TopOrder {
  std::vector<BuildNode>& n;
  std::vector<int32_t> subs, inner;  // the frontier and the inner nodes above it (the root first)
  std::vector<Box> box;              // per subset of the frontier
  std::vector<float> cost;
  std::vector<uint32_t> split;
  size_t next_inner = 0;
  int32_t emit(uint32_t mask)
  {
    if ((mask & (mask - 1)) == 0) return subs[__builtin_ctz(mask)];
    const int32_t i = inner[next_inner++];
    const uint32_t a = split[mask];
    n[i].box = box[mask];
    const int32_t l = emit(a), r = emit(mask ^ a);
    n[i].left = l; n[i].right = r;
    return i;
  }
  // returns false (tree unchanged) when there is nothing to order or the result would be deeper than max_levels
  bool run(int max_levels)
  {
    subs.push_back(0);
    while ((int)subs.size() < kTopOrderSubtrees) {
      int open = -1;
      float largest = -1.f;
      for (int j = 0; j < (int)subs.size(); j++)
        if (n[subs[j]].left >= 0 && n[subs[j]].box.half_area() > largest) { largest = n[subs[j]].box.half_area(); open = j; }
      if (open < 0) break;
      const int32_t i = subs[open];
      inner.push_back(i);  // the root is the first one opened
      subs[open] = n[i].left;
      subs.push_back(n[i].right);
    }
    const int k = (int)subs.size();
    if (k < 3) return false;
    const uint32_t full = (1u << k) - 1u;
    box.resize(full + 1); cost.assign(full + 1, 0.f); split.assign(full + 1, 0u);
    for (uint32_t m = 1; m <= full; m++) {
      const int low = __builtin_ctz(m);
      if ((m & (m - 1)) == 0) { box[m] = n[subs[low]].box; continue; }
      box[m] = box[m & (m - 1)];
      box[m].grow(n[subs[low]].box);
      float best = std::numeric_limits<float>::max();
      uint32_t best_part = 0;
      const uint32_t rest = m & (m - 1);  // every split of m into a part with its lowest member and a part without
      for (uint32_t a = rest; a != 0u; a = (a - 1) & rest) {
        const float c = cost[m ^ a] + cost[a];
        if (c < best) { best = c; best_part = m ^ a; }
      }
      cost[m] = best + box[m].half_area();
      split[m] = best_part;
    }
    std::vector<BuildNode> saved;
    for (int32_t i : inner) saved.push_back(n[i]);
    next_inner = 0;
    emit(full);
    if (tree_levels(n, 0) <= max_levels) return true;
    for (size_t j = 0; j < inner.size(); j++) n[inner[j]] = saved[j];
    return false;
  }
};

inline float f_down(double x)
{
  float f = (float)x;
  return ((double)f > x) ? std::nextafter(f, -std::numeric_limits<float>::infinity()) : f;
}
inline float f_up(double x)
{
  float f = (float)x;
  return ((double)f < x) ? std::nextafter(f, std::numeric_limits<float>::infinity()) : f;
}


void set_child_box(NodePair& n, int which, const Box& b, const BoxGrid& g) { set_child_box(n, which, b.lo, b.hi, g); }

// ---- on-disk tree cache ---------------------------------------------------------------------------
struct CacheKey {
  uint64_t a, b;
};

// Two independent multiply-xorshift hashes over the boxes (the only input of the tree) and the limits
// that shape it.
CacheKey hash_boxes(const std::vector<Box>& boxes, int32_t n_triangles)
{
  uint64_t h0 = 0x243F6A8885A308D3ull ^ (uint64_t)boxes.size(), h1 = 0x13198A2E03707344ull + (uint64_t)boxes.size();
  auto mix = [&](uint64_t w) {
    h0 = (h0 ^ w) * 0x9E3779B97F4A7C15ull; h0 ^= h0 >> 29;
    h1 = (h1 + w) * 0xC2B2AE3D27D4EB4Full; h1 ^= h1 >> 31;
  };
  mix(((uint64_t)kMaxLeafSize << 32) | ((uint64_t)kSahLeafMax << 16) | (uint64_t)kMaxTreeDepth);
  mix(((uint64_t)kBins << 32) | ((uint64_t)kTopOrderSubtrees << 16) | ((uint64_t)kRotationPasses << 8) | 6u /* builder revision: change when the builder's decisions change */);
  mix((uint64_t)(uint32_t)n_triangles);  // which primitives are spheres: leaf links say whether a leaf holds one
  uint32_t ic, tc;
  std::memcpy(&ic, &kIntersectCost, 4); std::memcpy(&tc, &kTraversalCost, 4);
  mix(((uint64_t)ic << 32) | tc);
  static_assert(sizeof(Box) == 24, "Box is six floats");
  const uint64_t* w = reinterpret_cast<const uint64_t*>(boxes.data());
  for (size_t i = 0, n = boxes.size() * 3; i < n; i++) mix(w[i]);
  return CacheKey{h0, h1};
}

struct CacheHeader {
  char magic[8];  // "PTKBVH6\0"
  uint64_t key_a, key_b;
  int32_t n_prims, n_nodes, max_depth, max_leaf;
};

std::string cache_path(const char* dir, CacheKey k)
{
  char name[64];
  snprintf(name, sizeof name, "/ptkbvh-%016llx%016llx.bin", (unsigned long long)k.a, (unsigned long long)k.b);
  return std::string(dir) + name;
}

// Everything the traversal relies on is checked: a damaged file must not send a kernel out of bounds.
bool tree_is_valid(const std::vector<NodePair>& nodes, const std::vector<uint32_t>& order, int32_t n_prims, int32_t max_depth,
                   int32_t max_leaf)
{
  if (nodes.empty() || (int32_t)order.size() != n_prims || max_depth > kMaxTreeDepth || max_leaf > kMaxLeafSize) return false;
  std::vector<uint8_t> seen((size_t)n_prims, 0);
  for (uint32_t id : order) {
    if (id >= (uint32_t)n_prims || seen[id]) return false;
    seen[id] = 1;
  }
  std::vector<uint8_t> visited(nodes.size(), 0);
  std::vector<uint8_t> covered((size_t)n_prims, 0);
  struct Todo { int32_t node; int32_t depth; };
  std::vector<Todo> stack{{0, 0}};
  while (!stack.empty()) {
    Todo t = stack.back(); stack.pop_back();
    if (t.node < 0 || (size_t)t.node >= nodes.size() || visited[t.node] || t.depth + 1 > max_depth) return false;
    visited[t.node] = 1;
    for (int32_t link : {nodes[t.node].child0, nodes[t.node].child1}) {
      if (link >= 0) { stack.push_back({link, t.depth + 1}); continue; }
      uint32_t first = leaf_first(link), count = leaf_count(link);
      if ((int32_t)count > max_leaf || (uint64_t)first + count > (uint64_t)n_prims) return false;
      for (uint32_t i = first; i < first + count; i++) {
        if (covered[i]) return false;
        covered[i] = 1;
      }
    }
  }
  for (uint8_t c : covered) if (!c) return false;
  return true;
}

bool cache_load(const std::string& path, CacheKey key, int32_t n_prims, std::vector<NodePair>& nodes, std::vector<uint32_t>& order,
                int32_t& max_depth, int32_t& max_leaf)
{
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  CacheHeader h;
  bool ok = std::fread(&h, sizeof h, 1, f) == 1 && std::memcmp(h.magic, "PTKBVH6", 8) == 0 && h.key_a == key.a &&
            h.key_b == key.b && h.n_prims == n_prims && h.n_nodes >= 1 && h.n_nodes <= n_prims + 1;
  if (ok) {
    nodes.resize((size_t)h.n_nodes);
    order.resize((size_t)n_prims);
    ok = std::fread(nodes.data(), sizeof(NodePair), nodes.size(), f) == nodes.size() &&
         std::fread(order.data(), sizeof(uint32_t), order.size(), f) == order.size() && std::fgetc(f) == EOF;
  }
  std::fclose(f);
  if (ok) ok = tree_is_valid(nodes, order, n_prims, h.max_depth, h.max_leaf);
  if (ok) { max_depth = h.max_depth; max_leaf = h.max_leaf; }
  return ok;
}

void cache_store(const std::string& path, CacheKey key, const std::vector<NodePair>& nodes, const std::vector<uint32_t>& order,
                 int32_t max_depth, int32_t max_leaf)
{
  // written beside the final name and renamed, so that a reader never sees half a file
  std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return;
  CacheHeader h;
  std::memset(&h, 0, sizeof h);
  std::memcpy(h.magic, "PTKBVH6", 8);
  h.key_a = key.a; h.key_b = key.b;
  h.n_prims = (int32_t)order.size(); h.n_nodes = (int32_t)nodes.size();
  h.max_depth = max_depth; h.max_leaf = max_leaf;
  bool ok = std::fwrite(&h, sizeof h, 1, f) == 1 && std::fwrite(nodes.data(), sizeof(NodePair), nodes.size(), f) == nodes.size() &&
            std::fwrite(order.data(), sizeof(uint32_t), order.size(), f) == order.size();
  ok = (std::fclose(f) == 0) && ok;
  if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) std::remove(tmp.c_str());
}

}  // namespace

SceneView HostScene::host_view() const
{
  SceneView v;
  std::memset(&v, 0, sizeof v);
  v.nodes = nodes.data();
  v.prims = prims.data();
  v.shade = shade.data();
  v.materials = materials.data();
  v.lights = lights.data();
  v.textures = textures.data() + kGammaLutRecords;
  v.texels = texels.data();
  v.n_prims = n_prims;
  v.n_triangles = n_triangles;
  v.grid = grid;
  v.n_lights = (int32_t)lights.size();
  v.n_nodes = (int32_t)nodes.size();
  v.n_materials = (int32_t)materials.size();
  v.background_texture = background_texture;
  v.background_is_color = background_is_color;
  for (int i = 0; i < 3; i++) v.background_color[i] = background_color[i];
  return v;
}

std::string build_host_scene(const ptk_scene_desc& d, HostScene& out, bool with_tree)
{
  out = HostScene();
  if (d.n_triangles < 0 || d.n_spheres < 0 || d.n_materials < 0 || d.n_textures < 0) return "negative count";
  const int64_t n_total = (int64_t)d.n_triangles + (int64_t)d.n_spheres;
  // a tree of depth kMaxTreeDepth with kMaxLeafSize primitives per leaf holds this many at most
  if (n_total > ((int64_t)kMaxLeafSize << kMaxTreeDepth) || n_total >= (1 << 28))
    return "too many primitives (limit " + std::to_string(std::min<int64_t>((int64_t)kMaxLeafSize << kMaxTreeDepth, (1 << 28) - 1)) + ")";
  if (d.n_triangles > 0 && (!d.tri_pos || !d.tri_nrm || !d.tri_uv || !d.tri_material)) return "null triangle array";
  if (d.n_spheres > 0 && (!d.sphere_center || !d.sphere_radius || !d.sphere_material)) return "null sphere array";
  if (d.n_materials > 0 && !d.materials) return "null material array";
  if (d.n_textures > 0 && !d.textures) return "null texture array";
  if (n_total > 0 && d.n_materials == 0) return "primitives need at least one material";
  if (d.background_texture >= d.n_textures) return "background_texture out of range";
  const int n = (int)n_total;
  out.n_prims = n;
  out.n_triangles = d.n_triangles;

  // --- textures ------------------------------------------------------------
  out.textures.resize(kGammaLutRecords);  // the gamma table goes first (ptk_layout.h)
  {
    float* lut = reinterpret_cast<float*>(out.textures.data());
    for (int i = 0; i < 256; i++) lut[i] = (float)std::pow((double)i / 255.0, 2.2);
  }
  for (int i = 0; i < d.n_textures; i++) {
    const ptk_texture& t = d.textures[i];
    if (t.width <= 0 || t.height <= 0 || !t.rgb) return "invalid texture";
    TextureRec r;
    r.offset = out.texels.size();
    r.width = t.width; r.height = t.height;
    size_t bytes = (size_t)t.width * (size_t)t.height * 3;
    out.texels.insert(out.texels.end(), t.rgb, t.rgb + bytes);
    out.texels.insert(out.texels.end(), 8, 0);  // the kernels fetch two texels as one 8-byte word: bytes beyond the last texel must exist
    while (out.texels.size() % 16) out.texels.push_back(0);
    if (out.texels.size() > 0xfffffff0ull) return "textures exceed 4 GiB (byte offsets into the texel pool are 32 bits)";
    out.textures.push_back(r);
  }
  if (out.textures.size() == (size_t)kGammaLutRecords) { TextureRec r; r.offset = 0; r.width = r.height = 0; out.textures.push_back(r); }
  if (out.texels.empty()) out.texels.resize(16, 0);

  // --- materials -------------------------------------------------------------
  for (int i = 0; i < d.n_materials; i++) {
    const ptk_material& m = d.materials[i];
    if (m.texture >= d.n_textures) return "material texture index out of range";
    MaterialRec r;
    std::memset(&r, 0, sizeof r);
    r.type = (uint32_t)m.type;
    r.texture = m.texture < 0 ? -1 : m.texture;
    r.ior = (float)m.refraction_index;
    r.roughness = (float)m.roughness;
    r.metallic = (float)m.metallic;
    for (int k = 0; k < 3; k++) { r.albedo[k] = (float)m.albedo[k]; r.emission[k] = (float)m.emission[k]; }
    if (m.texture >= 0) {
      // a textured material's colour comes from the texture alone (Intersection::albedo, src/geometry.cpp:18-26): the
      // three words hold the texture's own record instead, which spares every textured hit a dependent fetch of it
      const TextureRec& t = out.textures[(size_t)kGammaLutRecords + (size_t)m.texture];
      const uint32_t words[3] = {(uint32_t)t.offset, (uint32_t)t.width, (uint32_t)t.height};
      std::memcpy(r.albedo, words, sizeof words);
    }
    if (m.emission[0] > 0.0 || m.emission[1] > 0.0 || m.emission[2] > 0.0) r.flags |= kMatLight;
    bool spec = (m.type & PTK_DIELECTRIC) || (m.type & PTK_MIRROR) ||
                (((m.type & PTK_SPECULAR) || (m.type & PTK_MICROFACET)) && m.roughness < 1e-5);
    if (spec) r.flags |= kMatSpecular;
    out.materials.push_back(r);
  }
  if (out.materials.empty()) { MaterialRec r; std::memset(&r, 0, sizeof r); r.type = PTK_DIFFUSE; r.texture = -1; out.materials.push_back(r); }

  // --- primitives: shading records, light list, bounding boxes -----------------------
  out.shade.resize(std::max(n, 1));
  std::memset(out.shade.data(), 0, out.shade.size() * sizeof(ShadeRec));
  std::vector<PrimRec> by_id(std::max(n, 1));
  std::memset(by_id.data(), 0, by_id.size() * sizeof(PrimRec));
  std::vector<Box> boxes(n);
  double max_abs = 1.0;
  // Triangles are independent: large scenes are converted by several threads, each on a contiguous share,
  // and the shares' lights are appended in share order so that the light list stays in id order.
  struct Share {
    std::string error;
    double max_abs = 1.0;
    std::vector<LightRec> lights;
  };
  auto convert = [&](int lo_i, int hi_i, Share& sh) {
  for (int i = lo_i; i < hi_i; i++) {
    const double* p = d.tri_pos + 9 * (size_t)i;
    const double* nr = d.tri_nrm + 9 * (size_t)i;
    const double* uv = d.tri_uv + 6 * (size_t)i;
    int mat = d.tri_material[i];
    if (mat < 0 || mat >= d.n_materials) { sh.error = "triangle material index out of range"; return; }
    for (int k = 0; k < 9; k++) {
      if (!std::isfinite(p[k])) { sh.error = "non-finite triangle vertex"; return; }
      sh.max_abs = std::max(sh.max_abs, std::fabs(p[k]));
    }
    ShadeRec& s = out.shade[i];
    for (int k = 0; k < 3; k++) { s.n0[k] = (float)nr[k]; s.n1[k] = (float)nr[3 + k]; s.n2[k] = (float)nr[6 + k]; }
    for (int k = 0; k < 2; k++) { s.t0[k] = (float)uv[k]; s.t1[k] = (float)uv[2 + k]; s.t2[k] = (float)uv[4 + k]; }
    s.material = (uint32_t)mat;
    PrimRec& r = by_id[i];
    float v0[3] = {(float)p[0], (float)p[1], (float)p[2]};
    float e1[3], e2[3];
    for (int k = 0; k < 3; k++) { e1[k] = (float)(p[3 + k] - (double)v0[k]); e2[k] = (float)(p[6 + k] - (double)v0[k]); }
    r.a = f4{v0[0], v0[1], v0[2], e1[0]};
    r.b = f4{e1[1], e1[2], e2[0], e2[1]};
    r.c = f4{e2[2], as_float((uint32_t)i), 0.f, 0.f};
    Box& b = boxes[i];
    for (int a = 0; a < 3; a++) {
      b.lo[a] = f_down(std::min(p[a], std::min(p[3 + a], p[6 + a])));
      b.hi[a] = f_up(std::max(p[a], std::max(p[3 + a], p[6 + a])));
    }
    if (out.materials[mat].flags & kMatLight) {
      LightRec l;
      std::memset(&l, 0, sizeof l);
      l.kind = kTriangle; l.id = (uint32_t)i;
      double a1[3] = {p[3] - p[0], p[4] - p[1], p[5] - p[2]}, a2[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};
      double cx = a1[1] * a2[2] - a2[1] * a1[2], cy = a1[2] * a2[0] - a2[2] * a1[0], cz = a1[0] * a2[1] - a2[0] * a1[1];
      l.area = (float)(0.5 * std::sqrt(cx * cx + cy * cy + cz * cz));
      for (int k = 0; k < 3; k++) {
        l.p0[k] = (float)p[k]; l.p1[k] = (float)p[3 + k]; l.p2[k] = (float)p[6 + k];
        l.n0[k] = (float)nr[k]; l.n1[k] = (float)nr[3 + k]; l.n2[k] = (float)nr[6 + k];
        l.emission[k] = out.materials[mat].emission[k];
      }
      l.flat = (std::memcmp(l.n0, l.n1, sizeof l.n0) == 0 && std::memcmp(l.n0, l.n2, sizeof l.n0) == 0) ? 1u : 0u;
      sh.lights.push_back(l);
    }
  }
  };
  {
    int shares = 1;
    if (d.n_triangles >= 65536) {
      shares = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
      if (const char* e = getenv("PTK_BUILD_THREADS")) shares = std::min(std::max(1, atoi(e)), 64);
    }
    std::vector<Share> share(shares);
    std::vector<std::thread> workers;
    for (int t = 1; t < shares; t++) {
      int lo_i = (int)((int64_t)d.n_triangles * t / shares), hi_i = (int)((int64_t)d.n_triangles * (t + 1) / shares);
      workers.emplace_back([&, lo_i, hi_i, t] { convert(lo_i, hi_i, share[t]); });
    }
    convert(0, (int)((int64_t)d.n_triangles / shares), share[0]);
    for (auto& w : workers) w.join();
    for (auto& sh : share) {
      if (!sh.error.empty()) return sh.error;
      max_abs = std::max(max_abs, sh.max_abs);
      out.lights.insert(out.lights.end(), sh.lights.begin(), sh.lights.end());
    }
  }
  for (int j = 0; j < d.n_spheres; j++) {
    int i = d.n_triangles + j;
    const double* c = d.sphere_center + 3 * (size_t)j;
    double rad = d.sphere_radius[j];
    int mat = d.sphere_material[j];
    if (mat < 0 || mat >= d.n_materials) return "sphere material index out of range";
    if (!std::isfinite(rad) || !std::isfinite(c[0]) || !std::isfinite(c[1]) || !std::isfinite(c[2])) return "non-finite sphere";
    ShadeRec& s = out.shade[i];
    for (int k = 0; k < 3; k++) s.n0[k] = (float)c[k];
    s.n1[0] = (float)rad;
    s.material = (uint32_t)mat;
    PrimRec& r = by_id[i];
    r.a = f4{(float)c[0], (float)c[1], (float)c[2], (float)rad};
    r.b = f4{0.f, 0.f, 0.f, 0.f};
    r.c = f4{0.f, as_float((uint32_t)i | kPrimSphereBit), 0.f, 0.f};
    Box& b = boxes[i];
    for (int a = 0; a < 3; a++) {
      b.lo[a] = f_down(c[a] - std::fabs(rad));
      b.hi[a] = f_up(c[a] + std::fabs(rad));
      max_abs = std::max(max_abs, std::max(std::fabs((double)b.lo[a]), std::fabs((double)b.hi[a])));
    }
    if (out.materials[mat].flags & kMatLight) {
      LightRec l;
      std::memset(&l, 0, sizeof l);
      l.kind = kSphere; l.id = (uint32_t)i;
      l.radius = (float)rad;
      l.area = (float)((4.0 * 3.14159265359 * rad * rad) / 2.0);  // src/geometry.cpp:252
      for (int k = 0; k < 3; k++) { l.p0[k] = (float)c[k]; l.emission[k] = out.materials[mat].emission[k]; }
      out.lights.push_back(l);
    }
  }

  // Boxes are widened by a few ulps of the largest coordinate: the fp32 slab test and the fp32
  // triangle test round independently, and a hit on a box face must not be lost to that.
  const float pad = (float)(max_abs * 2e-6);
  Box scene_box;
  scene_box.reset();
  for (auto& b : boxes) {
    for (int a = 0; a < 3; a++) { b.lo[a] -= pad; b.hi[a] += pad; }
    scene_box.grow(b);
  }
  if (n == 0) { for (int a = 0; a < 3; a++) { scene_box.lo[a] = 0.f; scene_box.hi[a] = 1.f; } }
  out.grid = make_box_grid(scene_box.lo, scene_box.hi);

  // --- background --------------------------------------------------------------
  out.background_texture = d.background_texture < 0 ? -1 : d.background_texture;
  out.background_is_color = (d.background_color[0] >= 0.0 && d.background_color[1] >= 0.0 && d.background_color[2] >= 0.0);
  for (int k = 0; k < 3; k++) out.background_color[k] = (float)d.background_color[k];

  // --- BVH -----------------------------------------------------------------------
  if (n == 0) {
    NodePair e;
    std::memset(&e, 0, sizeof e);
    // both children: the one all-zero primitive record (a degenerate triangle that no ray hits)
    set_child_box(e, 0, scene_box, out.grid); set_child_box(e, 1, scene_box, out.grid);
    e.child0 = e.child1 = leaf_code(0u, 1u, false);
    out.nodes.push_back(e);
    out.prims.assign(1, by_id[0]);
    return std::string();
  }
  if (!with_tree && n >= kDeviceBuildMin) {
    out.prims = std::move(by_id);
    out.prim_boxes.resize(6 * (size_t)n);
    Box cb;
    cb.reset();
    float diag2 = 0.f;
    for (int i = 0; i < n; i++) {
      const Box& b = boxes[i];
      float c[3], e2 = 0.f;
      for (int a = 0; a < 3; a++) {
        out.prim_boxes[6 * (size_t)i + a] = b.lo[a];
        out.prim_boxes[6 * (size_t)i + 3 + a] = b.hi[a];
        c[a] = 0.5f * (b.lo[a] + b.hi[a]);
        e2 += (b.hi[a] - b.lo[a]) * (b.hi[a] - b.lo[a]);
      }
      cb.grow(c);
      diag2 = std::max(diag2, e2);
    }
    for (int a = 0; a < 3; a++) { out.centroid_lo[a] = cb.lo[a]; out.centroid_hi[a] = cb.hi[a]; }
    out.max_diagonal = std::sqrt(diag2);
    return std::string();
  }
  const bool timing = getenv("PTK_BUILD_TIMING") != nullptr;
  auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[ptk build] %s: %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
    t0 = t1;
  };
  const char* cache_dir = getenv("PTK_BVH_CACHE");
  if (getenv("PTK_TREE_ROTATIONS") || getenv("PTK_TREE_TOP_ORDER")) cache_dir = nullptr;  // an experiment's tree is not what the cache's key describes
  std::string cache_file;
  CacheKey cache_key{0, 0};
  if (cache_dir && *cache_dir && n >= kDeviceBuildMin) {
    cache_key = hash_boxes(boxes, out.n_triangles);
    cache_file = cache_path(cache_dir, cache_key);
    std::vector<uint32_t> order;
    if (cache_load(cache_file, cache_key, n, out.nodes, order, out.max_depth, out.max_leaf)) {
      out.prims.resize((size_t)n);
      for (int i = 0; i < n; i++) out.prims[i] = by_id[order[i]];
      out.tree_from_cache = true;
      lap("tree read from the cache");
      return std::string();
    }
    out.nodes.clear();
    lap("cache miss");
  }
  // 16 = one GPU's share of a node's host cores (std::thread::hardware_concurrency() sees the whole node)
  int threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char* e = getenv("PTK_BUILD_THREADS")) threads = std::min(std::max(1, atoi(e)), Builder::kMaxThreads);
  Builder bld(boxes, threads);
  lap("centroids");
  bld.build(0, n, 0, threads);
  if (bld.out_of_slots.load()) return "internal: BVH builder ran out of node slots";
  bld.finish();
  lap("tree");
  out.max_depth = bld.max_depth;
  out.max_leaf = bld.max_leaf;
  const char* top_order = getenv("PTK_TREE_TOP_ORDER");
  if ((!top_order || atoi(top_order) != 0) && n >= kTopOrderMin) {
    // the tree may get deeper, but not so deep that the scene loses the kernel form without overflow entries
    const int levels = out.max_depth <= kTablesTreeDepth ? kTablesTreeDepth : kMaxTreeDepth;
    TopOrder top{bld.nodes, {}, {}, {}, {}, {}};
    if (top.run(levels)) out.max_depth = tree_levels(bld.nodes, 0);
    lap("top order");
  }
  {
    const char* e = getenv("PTK_TREE_ROTATIONS");
    Rotations rot{bld.nodes, std::vector<int8_t>(bld.nodes.size(), 0), out.max_depth};
    for (int p = 0, passes = e ? atoi(e) : kRotationPasses; p < passes; p++) {
      const long before = rot.applied;
      out.max_depth = rot.pass(0, 1);
      if (rot.applied == before) break;
    }
    if (timing) fprintf(stderr, "[ptk build] %ld rotations, depth %d\n", rot.applied, out.max_depth);
    lap("rotations");
  }
  if (out.max_depth > kMaxTreeDepth) return "internal: BVH deeper than the traversal stack allows";

  // Flatten.  Leaves take their primitives in depth-first order; an inner build node becomes
  // one NodePair carrying both children's boxes.
  out.prims.reserve(n);
  struct Item { int32_t build_node; int32_t out_node; };
  std::vector<int32_t> leaf_first(bld.nodes.size(), -1);
  // first pass: assign primitive slots to leaves in depth-first order
  {
    std::vector<int32_t> stack{0};
    while (!stack.empty()) {
      int32_t b = stack.back(); stack.pop_back();
      const BuildNode& bn = bld.nodes[b];
      if (bn.left < 0) {
        leaf_first[b] = (int32_t)out.prims.size();
        for (int i = bn.first; i < bn.first + bn.count; i++) out.prims.push_back(by_id[bld.items[i].id]);
      } else {
        stack.push_back(bn.right);
        stack.push_back(bn.left);
      }
    }
  }
  auto holds_sphere = [&](int32_t first, int32_t count) {
    for (int32_t i = first; i < first + count; i++)
      if (as_uint(out.prims[(size_t)i].c.y) & kPrimSphereBit) return true;
    return false;
  };
  auto child_code = [&](int32_t b, std::vector<Item>& todo) -> int32_t {
    const BuildNode& bn = bld.nodes[b];
    if (bn.left < 0) return leaf_code((uint32_t)leaf_first[b], (uint32_t)bn.count, holds_sphere(leaf_first[b], bn.count));
    int32_t idx = (int32_t)out.nodes.size();
    out.nodes.emplace_back();
    std::memset(&out.nodes.back(), 0, sizeof(NodePair));
    todo.push_back(Item{b, idx});
    return idx;
  };
  std::vector<Item> todo;
  if (bld.nodes[0].left < 0) {  // the whole scene is one leaf
    NodePair e;
    std::memset(&e, 0, sizeof e);
    // grid positions cannot express "nowhere", so the second child repeats the first: its primitives are
    // tested twice, and the second time none of them is strictly closer
    set_child_box(e, 0, bld.nodes[0].box, out.grid);
    set_child_box(e, 1, bld.nodes[0].box, out.grid);
    e.child0 = e.child1 = leaf_code(0u, (uint32_t)bld.nodes[0].count, holds_sphere(0, bld.nodes[0].count));
    out.nodes.push_back(e);
    return std::string();
  }
  out.nodes.reserve(bld.nodes.size() / 2 + 1);
  out.nodes.emplace_back();
  std::memset(&out.nodes.back(), 0, sizeof(NodePair));
  todo.push_back(Item{0, 0});
  while (!todo.empty()) {
    Item it = todo.back(); todo.pop_back();
    const BuildNode& bn = bld.nodes[it.build_node];
    // right first so that the left subtree's nodes follow their parent directly in memory
    std::vector<Item> mine;
    int32_t c0 = child_code(bn.left, mine);
    int32_t c1 = child_code(bn.right, mine);
    NodePair& np = out.nodes[it.out_node];
    set_child_box(np, 0, bld.nodes[bn.left].box, out.grid);
    set_child_box(np, 1, bld.nodes[bn.right].box, out.grid);
    np.child0 = c0; np.child1 = c1;
    for (auto r = mine.rbegin(); r != mine.rend(); ++r) todo.push_back(*r);
  }
  lap("flatten");
  if (!cache_file.empty()) {
    std::vector<uint32_t> order((size_t)n);
    for (int i = 0; i < n; i++) order[i] = as_uint(out.prims[i].c.y) & kPrimIdMask;
    cache_store(cache_file, cache_key, out.nodes, order, out.max_depth, out.max_leaf);
    lap("cache written");
  }
  return std::string();
}

CameraView make_camera_view(const ptk_camera& c)
{
  CameraView v;
  std::memset(&v, 0, sizeof v);
  v.width = c.width; v.height = c.height;
  for (int k = 0; k < 3; k++) {
    v.position[k] = (float)c.position[k]; v.forward[k] = (float)c.forward[k];
    v.right[k] = (float)c.right[k]; v.up[k] = (float)c.up[k];
  }
  double aspect = (double)c.width / (double)c.height;
  double half_height = std::tan(c.fov / 2.0);           // src/camera.cpp:29
  v.plane_h = (float)(2.0 * half_height);               // src/camera.cpp:33
  v.plane_w = (float)(2.0 * half_height * aspect);      // src/camera.cpp:30,32
  v.focus_distance = (float)c.focus_distance;
  v.defocus_radius = 0.f;
  if (c.aperture > 0 && c.focus_distance > 0)           // src/camera.cpp:44-50
    v.defocus_radius = (float)(c.focus_distance * std::tan((c.aperture / 2.0) * 0.017453292519943295));
  return v;
}

FlagView make_flag_view(const ptk_flags* f)
{
  FlagView v = {0, 1, 1, 1, 1, 1};  // src/config.h:3-8
  if (f) {
    v.debug_normal = f->debug_normal; v.russian_roulette = f->russian_roulette;
    v.direct_light_sampling = f->direct_light_sampling; v.indirect_light_sampling = f->indirect_light_sampling;
    v.importance_sample = f->importance_sample; v.check_hemisphere = f->check_hemisphere;
  }
  return v;
}

}  // namespace ptk
