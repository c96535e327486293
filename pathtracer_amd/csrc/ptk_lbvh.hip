// On-device tree construction (see ptk_lbvh.h).
//
//   1. size_histogram_kernel + key_kernel: one 64-bit sort key per primitive - the box's size class in the top four
//      bits, then 20 bits each of the box centre's x, y, z inside the scene's centre bounds, interleaved.  The size
//      classes keep the few large primitives of a scene (walls, floors) together at the top of the tree instead
//      of scattering them over leaves of small ones, whose boxes they would blow up; they are the groups that empty
//      bins of a logarithmic size histogram keep apart, so that no class boundary runs through a population of
//      similar primitives.
//   2. rocprim radix sort of (key, primitive) pairs.
//   3. split_level_kernel, one launch per tree level: every node of the level splits its range of
//      the sorted sequence where the highest differing key bit flips (binary search), or at the
//      median when the keys are equal or when the larger side could not be finished within the depth
//      limit any more, down to single primitives (or small leaves where the depth limit is near).  The host reads one
//      counter per level (the number of nodes so far) - at most kMaxTreeDepth small copies.
//   4. fit_level_kernel, deepest level first: child boxes from the primitives' boxes (leaves) or from
//      the child's own record (inner nodes); subtrees of at most kMaxLeafSize primitives are merged into one
//      leaf where the surface-area heuristic prefers that, and the four grandchildren of a node are paired
//      anew where another pairing has less surface area (regroup).
//   5. expand_level_kernel lists the nodes in use level by level; treelet_level_kernel, deepest level first, replaces
//      the tree over the six largest subtrees below every node by the cheapest one (two passes).
//   6. count_level_kernel (primitives below every node), then number_level_kernel, top level first, gives the nodes
//      still in use their final depth-first positions and the leaves their primitive ranges in the same order;
//      compact_kernel moves the nodes there and renumbers the links; gather_kernel: primitive records into leaf order.
#include "ptk_lbvh.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>  // rocprim headers use memset without including it
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

namespace ptk {

namespace {

constexpr int kBuildBlock = 256;

struct BuildRange {
  uint32_t first, count;
};

struct KeyParams {
  float lo[3], scale[3];
  float size_scale;
};

// bit k of a 20-bit value -> bit 3k
__device__ inline uint64_t spread3(uint32_t v)
{
  uint64_t x = v & 0xfffffu;
  x = (x | (x << 32)) & 0x001f00000000ffffull;
  x = (x | (x << 16)) & 0x001f0000ff0000ffull;
  x = (x | (x << 8)) & 0x100f00f00f00f00full;
  x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}

// Sizes are looked at on a logarithmic scale: bin k holds the boxes whose diagonal is 2^(-k/2) .. 2^(-(k+1)/2) of the
// largest one's (kp.size_scale = 1 / largest diagonal).
constexpr int kSizeBins = 64;
__device__ inline int size_bin(float relative_diagonal)
{
  const int b = relative_diagonal > 0.0f ? (int)(-2.0f * log2f(relative_diagonal)) : kSizeBins - 1;
  return min(max(b, 0), kSizeBins - 1);
}

__global__ __launch_bounds__(kBuildBlock) void size_histogram_kernel(const float* __restrict__ boxes, int32_t n, KeyParams kp,
                                                                     uint32_t* __restrict__ histogram)
{
  __shared__ uint32_t local[kSizeBins];
  if (threadIdx.x < kSizeBins) local[threadIdx.x] = 0u;
  __syncthreads();
  int32_t i = blockIdx.x * kBuildBlock + threadIdx.x;
  if (i < n) {
    const float* b = boxes + 6 * (size_t)i;
    const float ex = b[3] - b[0], ey = b[4] - b[1], ez = b[5] - b[2];
    atomicAdd(&local[size_bin(sqrtf(ex * ex + ey * ey + ez * ez) * kp.size_scale)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < kSizeBins && local[threadIdx.x]) atomicAdd(&histogram[threadIdx.x], local[threadIdx.x]);
}

// Sort key: the box's size class in the top four bits, then its centre as a 60-bit position on a space-filling curve.
// size_class[bin]: which class a size bin belongs to (class 0 = the largest boxes); classes are the groups of
// occupied bins that empty bins keep apart (see build_device_tree), so that a class boundary never runs through a
// population of similar boxes - a boundary inside the mesh's own range of triangle sizes would make two trees of
// interleaved triangles over the same space.
__global__ __launch_bounds__(kBuildBlock) void key_kernel(const float* __restrict__ boxes, int32_t n, KeyParams kp,
                                                          const uint8_t* __restrict__ size_class,
                                                          uint64_t* __restrict__ keys, uint32_t* __restrict__ order)
{
  int32_t i = blockIdx.x * kBuildBlock + threadIdx.x;
  if (i >= n) return;
  const float* b = boxes + 6 * (size_t)i;
  float ex = b[3] - b[0], ey = b[4] - b[1], ez = b[5] - b[2];
  // size class first, then 20 bits per axis of the box centre
  const uint32_t cls = size_class[size_bin(sqrtf(ex * ex + ey * ey + ez * ez) * kp.size_scale)];
  const uint32_t px = (uint32_t)fminf(fmaxf((0.5f * (b[0] + b[3]) - kp.lo[0]) * kp.scale[0], 0.0f), 1048575.0f);
  const uint32_t py = (uint32_t)fminf(fmaxf((0.5f * (b[1] + b[4]) - kp.lo[1]) * kp.scale[1], 0.0f), 1048575.0f);
  const uint32_t pz = (uint32_t)fminf(fmaxf((0.5f * (b[2] + b[5]) - kp.lo[2]) * kp.scale[2], 0.0f), 1048575.0f);
  keys[i] = ((uint64_t)cls << 60) | (spread3(px) << 2) | (spread3(py) << 1) | spread3(pz);
  order[i] = (uint32_t)i;
}

// Levels of median splitting needed until every part holds <= kMaxLeafSize primitives.
__device__ inline int median_levels(uint32_t n)
{
  int l = 0;
  while (n > (uint32_t)kMaxLeafSize) { n = (n + 1u) / 2u; l++; }
  return l;
}

// (whether a leaf holds a sphere is settled by the fitting pass, which looks at its primitives anyway)
__device__ inline int32_t leaf_link(uint32_t first, uint32_t count, bool spheres = false) { return leaf_code(first, count, spheres); }

// Nodes [begin, end) are the inner nodes at `depth`; each holds a range of at least two primitives.
// Ranges of at most this many primitives are not split where a key bit flips but where the surface-area heuristic is
// lowest among ALL positions of the (sorted) range: a sweep from the right stores the area of every suffix's box, a
// sweep from the left prices every position.  Bit flips are spatial medians; near the leaves, where most of a ray's
// node visits and all of its primitive tests happen, they leave boxes that overlap more than they have to.
#ifndef PTK_LBVH_SAH_RANGE
#define PTK_LBVH_SAH_RANGE 64
#endif
constexpr uint32_t kSahRange = PTK_LBVH_SAH_RANGE;

struct FBox {
  float lo[3], hi[3];
  __device__ void reset() { for (int k = 0; k < 3; k++) { lo[k] = 3.0e38f; hi[k] = -3.0e38f; } }
  __device__ void add(const float* b) { for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], b[k]); hi[k] = fmaxf(hi[k], b[3 + k]); } }
  __device__ float half_area() const
  {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
  }
};

__global__ __launch_bounds__(kBuildBlock) void split_level_kernel(const uint64_t* __restrict__ keys, BuildRange* ranges,
                                                                  NodePair* nodes, uint32_t begin, uint32_t end, int depth,
                                                                  uint32_t* counter, const float* __restrict__ boxes,
                                                                  const uint32_t* __restrict__ order)
{
  uint32_t i = begin + blockIdx.x * kBuildBlock + threadIdx.x;
  if (i >= end) return;
  const BuildRange r = ranges[i];
  const uint32_t a = r.first, b = r.first + r.count - 1u;
  const uint64_t ka = keys[a], kb = keys[b];
  uint32_t mid = 0;  // first primitive of the second part
  bool by_key = ka != kb;
  if (by_key) {
    const int bit = 63 - __clzll((long long)(ka ^ kb));
    uint32_t lo = a, hi = b;  // the bit is clear at lo and set at hi; everything above it is equal in the range
    while (hi - lo > 1u) {
      uint32_t m = lo + ((hi - lo) >> 1);
      if ((keys[m] >> bit) & 1ull) hi = m; else lo = m;
    }
    mid = hi;
    const uint32_t left = mid - a;
    const uint32_t big = left > r.count - left ? left : r.count - left;
    if (depth + 1 + median_levels(big) > kMaxTreeDepth) by_key = false;
  }
  if (!by_key) mid = a + (r.count + 1u) / 2u;
  if (kSahRange > 0u && r.count > 2u && r.count <= kSahRange) {
    float suffix[kSahRange > 0u ? kSahRange : 1u];  // suffix[k]: half the surface area of the box of primitives a + k .. b
    FBox acc;
    acc.reset();
    for (uint32_t k = r.count - 1u; k >= 1u; k--) {
      acc.add(boxes + 6 * (size_t)order[a + k]);
      suffix[k] = acc.half_area();
    }
    acc.reset();
    float best = 3.0e38f;
    uint32_t best_k = 0u;
    for (uint32_t k = 1u; k < r.count; k++) {
      acc.add(boxes + 6 * (size_t)order[a + k - 1u]);
      const float cost = acc.half_area() * (float)k + suffix[k] * (float)(r.count - k);
      if (cost < best) { best = cost; best_k = k; }
    }
    const uint32_t big = best_k > r.count - best_k ? best_k : r.count - best_k;
    if (best_k > 0u && depth + 1 + median_levels(big) <= kMaxTreeDepth) mid = a + best_k;
  }
  const uint32_t n0 = mid - a, n1 = r.count - n0;
  // A part becomes a leaf when it is one primitive, or when it is small enough for a leaf and the depth limit
  // leaves no room to split it down to single primitives.  Otherwise it is split further; the fitting pass
  // merges small subtrees back into leaves where the surface-area heuristic prefers that.
  auto child = [&](uint32_t first, uint32_t count) -> int32_t {
    int to_one = 0;
    while ((1u << to_one) < count) to_one++;
    if (count == 1u || (count <= (uint32_t)kMaxLeafSize && depth + 1 + to_one > kMaxTreeDepth)) return leaf_link(first, count);
    uint32_t idx = atomicAdd(counter, 1u);
    ranges[idx] = BuildRange{first, count};
    return (int32_t)idx;
  };
  const int32_t c0 = child(a, n0), c1 = child(mid, n1);
  nodes[i].child0 = c0;
  nodes[i].child1 = c1;
}

// A box as the nodes store it: per axis, low grid plane | extent (grid_pack in ptk_layout.h).
struct QBox {
  uint32_t x, y, z;
};

__device__ inline uint32_t q_union(uint32_t a, uint32_t b)
{
  const uint32_t lo_a = a & 0xffffu, lo_b = b & 0xffffu;
  const uint32_t lo = min(lo_a, lo_b), hi = max(lo_a + grid_extent_steps(a), lo_b + grid_extent_steps(b));
  return grid_pack(lo, hi - lo);
}

__device__ inline float q_half_area(const QBox& b, const BoxGrid& g)
{
  float dx = (float)grid_extent_steps(b.x) * g.scale[0];
  float dy = (float)grid_extent_steps(b.y) * g.scale[1];
  float dz = (float)grid_extent_steps(b.z) * g.scale[2];
  return dx * dy + dy * dz + dz * dx;
}

constexpr float kTraversalCost = 1.0f, kIntersectCost = 1.2f;  // the host builder's constants
#ifndef PTK_LBVH_MERGE_BIAS
#define PTK_LBVH_MERGE_BIAS 1.0f
#endif
constexpr float kMergeBias = PTK_LBVH_MERGE_BIAS;  // < 1: fewer, smaller leaves
// Largest leaf that merging may form.  (Leaves of up to kMaxLeafSize primitives still arise where the depth limit
// leaves no room to split further.)  A wave's leaf phase takes as many passes as its lane with the most primitives
// to test, so a few large leaves among many small ones cost every lane of the wave.
#ifndef PTK_LBVH_MERGED_LEAF_MAX
#define PTK_LBVH_MERGED_LEAF_MAX 4
#endif
constexpr uint32_t kMergedLeafMax = PTK_LBVH_MERGED_LEAF_MAX;

// Box of the child behind `link`.  An inner child whose own children are both leaves, of at most
// kMaxLeafSize primitives together, is replaced by one leaf when testing all of them costs no more
// (surface-area heuristic) than visiting the node first; `link` is updated then and the node is left unused.
__device__ inline QBox child_box(int32_t& link, const float* __restrict__ boxes, const uint32_t* __restrict__ order,
                                 const NodePair* nodes, const BoxGrid& g, uint32_t n_triangles)
{
  QBox r;
  if (link < 0) {
    const uint32_t first = leaf_first(link), count = leaf_count(link);
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    bool spheres = false;
    for (uint32_t j = 0; j < count; j++) {
      const uint32_t id = order[first + j];
      spheres = spheres || id >= n_triangles;  // primitive ids: triangles first
      const float* b = boxes + 6 * (size_t)id;
      for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], b[k]); hi[k] = fmaxf(hi[k], b[3 + k]); }
    }
    link = leaf_link(first, count, spheres);
    r.x = grid_word(lo[0], hi[0], g.origin[0], g.scale[0]);
    r.y = grid_word(lo[1], hi[1], g.origin[1], g.scale[1]);
    r.z = grid_word(lo[2], hi[2], g.origin[2], g.scale[2]);
  } else {
    const NodePair c = nodes[link];
    const QBox b0{c.x0, c.y0, c.z0}, b1{c.x1, c.y1, c.z1};
    r.x = q_union(b0.x, b1.x); r.y = q_union(b0.y, b1.y); r.z = q_union(b0.z, b1.z);
    if (c.child0 < 0 && c.child1 < 0) {
      const uint32_t n0 = leaf_count(c.child0), n1 = leaf_count(c.child1);
      // one leaf needs one range: the second has to start where the first ends (always so unless regroup() moved a leaf here)
      if (n0 + n1 <= kMergedLeafMax && leaf_first(c.child1) == leaf_first(c.child0) + n0) {
        const float a = q_half_area(r, g);
        const float split = kTraversalCost * a + kIntersectCost * (q_half_area(b0, g) * (float)n0 + q_half_area(b1, g) * (float)n1);
        const float leaf = kIntersectCost * a * (float)(n0 + n1);
        if (leaf <= kMergeBias * split) link = leaf_link(leaf_first(c.child0), n0 + n1, leaf_has_spheres(c.child0) || leaf_has_spheres(c.child1));
      }
    }
  }
  return r;
}

__device__ inline float pair_area(const QBox& p, const QBox& q, const BoxGrid& g)
{
  const QBox u{q_union(p.x, q.x), q_union(p.y, q.y), q_union(p.z, q.z)};
  return q_half_area(u, g);
}

// The sort order decides which two grandchildren share a child; where another pairing of the four gives
// the two children less surface area together, they are regrouped.  Depths stay as they are (grandchildren
// remain grandchildren), so the level lists and the depth limit are not affected.  l and r are inner
// nodes that have been fitted already and belong to the calling thread's node alone.
__device__ inline void regroup(NodePair* nodes, int32_t l, int32_t r, uint32_t* __restrict__ subtree, const BoxGrid& g)
{
  NodePair L = nodes[l], R = nodes[r];
  struct Part {
    int32_t link;
    QBox box;
  };
  Part a{L.child0, {L.x0, L.y0, L.z0}}, b{L.child1, {L.x1, L.y1, L.z1}};
  Part c{R.child0, {R.x0, R.y0, R.z0}}, d{R.child1, {R.x1, R.y1, R.z1}};
  const float now = pair_area(a.box, b.box, g) + pair_area(c.box, d.box, g);
  const float ac_bd = pair_area(a.box, c.box, g) + pair_area(b.box, d.box, g);
  const float ad_cb = pair_area(a.box, d.box, g) + pair_area(c.box, b.box, g);
  const float best = fminf(ac_bd, ad_cb);
  if (!(best < 0.98f * now)) return;
  Part t = b;
  if (ac_bd <= ad_cb) { b = c; c = t; }  // (a, c) and (b, d)
  else { b = d; d = t; }                 // (a, d) and (c, b)
  L.child0 = a.link; L.x0 = a.box.x; L.y0 = a.box.y; L.z0 = a.box.z;
  L.child1 = b.link; L.x1 = b.box.x; L.y1 = b.box.y; L.z1 = b.box.z;
  R.child0 = c.link; R.x0 = c.box.x; R.y0 = c.box.y; R.z0 = c.box.z;
  R.child1 = d.link; R.x1 = d.box.x; R.y1 = d.box.y; R.z1 = d.box.z;
  nodes[l] = L;
  nodes[r] = R;
  subtree[l] = 1u + (a.link >= 0 ? subtree[a.link] : 0u) + (b.link >= 0 ? subtree[b.link] : 0u);
  subtree[r] = 1u + (c.link >= 0 ? subtree[c.link] : 0u) + (d.link >= 0 ? subtree[d.link] : 0u);
}

// Nodes [begin, end) are one level; every deeper level has been fitted already.
__global__ __launch_bounds__(kBuildBlock) void fit_level_kernel(const float* __restrict__ boxes,
                                                                const uint32_t* __restrict__ order, NodePair* nodes,
                                                                uint32_t begin, uint32_t end, uint32_t* __restrict__ live,
                                                                uint32_t* __restrict__ subtree, BoxGrid g, uint32_t n_triangles)
{
  uint32_t i = begin + blockIdx.x * kBuildBlock + threadIdx.x;
  if (i >= end) return;
  int32_t c0 = nodes[i].child0, c1 = nodes[i].child1;
#ifndef PTK_LBVH_NO_REGROUP
  if (c0 >= 0 && c1 >= 0) regroup(nodes, c0, c1, subtree, g);
#endif
  const QBox b0 = child_box(c0, boxes, order, nodes, g, n_triangles), b1 = child_box(c1, boxes, order, nodes, g, n_triangles);
  nodes[i].child0 = c0;
  nodes[i].child1 = c1;
  // A node merged into a leaf has only leaves below it, so whatever an inner link still points at is in use.
  if (c0 >= 0) live[c0] = 1u;
  if (c1 >= 0) live[c1] = 1u;
  subtree[i] = 1u + (c0 >= 0 ? subtree[c0] : 0u) + (c1 >= 0 ? subtree[c1] : 0u);  // inner nodes in use from here down
  nodes[i].x0 = b0.x; nodes[i].y0 = b0.y; nodes[i].z0 = b0.z;
  nodes[i].x1 = b1.x; nodes[i].y1 = b1.y; nodes[i].z1 = b1.z;
}

// The nodes in use, level by level: list[begin .. end) holds one level, the inner children of its nodes are appended
// behind `counter` (which starts at `end`).  Needed once the rotations below have moved subtrees between levels.
__global__ __launch_bounds__(kBuildBlock) void expand_level_kernel(const NodePair* __restrict__ nodes, uint32_t* list,
                                                                   uint32_t begin, uint32_t end, uint32_t* counter)
{
  uint32_t k = begin + blockIdx.x * kBuildBlock + threadIdx.x;
  if (k >= end) return;
  const NodePair nd = nodes[list[k]];
  if (nd.child0 >= 0) list[atomicAdd(counter, 1u)] = (uint32_t)nd.child0;
  if (nd.child1 >= 0) list[atomicAdd(counter, 1u)] = (uint32_t)nd.child1;
}

// How the fitted tree is restructured: treelets (treelet_level_kernel below) or, with 0, rotations alone.
// Measured (82k / 1.3M triangles): treelets of 6, two passes: box tests per ray 21.5 -> 20.0 and 24.9 -> 23.4, frame rate
// +2.2 % / +0.8 %, build 8.5 -> 11.9 ms and 11.7 -> 17.7 ms; of 7: +2.3 % / +1.1 % for 17 and 26 ms; of 5: slower
// frames unless small gains are refused.  0 builds the rotations alone.
#ifndef PTK_LBVH_TREELET
#define PTK_LBVH_TREELET 6  // subtrees per treelet
#endif
#ifndef PTK_LBVH_ROTATION_PASSES
#define PTK_LBVH_ROTATION_PASSES 2
#endif
constexpr int kRotationPasses = PTK_LBVH_ROTATION_PASSES;  // passes of either kind

#if PTK_LBVH_TREELET < 3
// Tree rotations, as the host builder does them after its build (ptk_build.cpp): a child of a node trades places with a
// grandchild on the other side when the inner child left behind gets a smaller box.  One level per launch, deepest
// first, so that a thread owns its node and that node's two children; `level` is the level of these nodes (root = 1),
// a trade pushes one subtree a level down and is only made where the tree stays within `limit` levels of inner nodes.
// height[i]: levels of inner nodes from i down.
// A trade has to shrink the inner child's area by this part of it.  Measured on these trees: the few trades with large
// gains are worth +1.7 % (82k triangles) and +2.8 % (1.3M triangles) of the frame rate, taking every trade with any
// gain (tens of thousands more) loses 2 % again.
#ifndef PTK_LBVH_ROTATION_MIN_GAIN
#define PTK_LBVH_ROTATION_MIN_GAIN 0.2f
#endif
constexpr float kRotationMinGain = PTK_LBVH_ROTATION_MIN_GAIN;
__global__ __launch_bounds__(kBuildBlock) void rotate_level_kernel(NodePair* nodes, const uint32_t* __restrict__ list,
                                                                   uint32_t begin, uint32_t end, int level, int limit,
                                                                   uint8_t* height, uint32_t* __restrict__ subtree, BoxGrid g,
                                                                   uint32_t* applied)
{
  uint32_t k = begin + blockIdx.x * kBuildBlock + threadIdx.x;
  if (k >= end) return;
  const uint32_t i = list[k];
  NodePair N = nodes[i];
  int32_t link[2] = {N.child0, N.child1};
  QBox box[2] = {{N.x0, N.y0, N.z0}, {N.x1, N.y1, N.z1}};
  int hgt[2] = {link[0] >= 0 ? (int)height[link[0]] : 0, link[1] >= 0 ? (int)height[link[1]] : 0};
  float best = 0.0f;
  int best_side = -1, best_up = -1;
  QBox best_box = box[0];
  for (int side = 0; side < 2; side++) {  // link[side] keeps one of its children and takes link[1 - side]
    if (link[side] < 0 || level + 1 + hgt[1 - side] > limit) continue;
    const NodePair C = nodes[link[side]];
    const QBox gb[2] = {{C.x0, C.y0, C.z0}, {C.x1, C.y1, C.z1}};
    const float now = q_half_area(box[side], g);
    for (int up = 0; up < 2; up++) {
      const QBox& stays = gb[1 - up];
      const QBox u{q_union(box[1 - side].x, stays.x), q_union(box[1 - side].y, stays.y), q_union(box[1 - side].z, stays.z)};
      const float gain = now - q_half_area(u, g);
      if (gain > best && gain > kRotationMinGain * now) { best = gain; best_side = side; best_up = up; best_box = u; }
    }
  }
  if (best_side >= 0) {
    const int side = best_side, other = 1 - side, up = best_up;
    const int32_t c = link[side];
    NodePair C = nodes[c];
    const int32_t up_link = up ? C.child1 : C.child0, stays_link = up ? C.child0 : C.child1;
    const QBox up_box = up ? QBox{C.x1, C.y1, C.z1} : QBox{C.x0, C.y0, C.z0};
    // c: the grandchild that moves up makes room for the node's other child
    if (up) { C.child1 = link[other]; C.x1 = box[other].x; C.y1 = box[other].y; C.z1 = box[other].z; }
    else { C.child0 = link[other]; C.x0 = box[other].x; C.y0 = box[other].y; C.z0 = box[other].z; }
    nodes[c] = C;
    const int stays_h = stays_link >= 0 ? (int)height[stays_link] : 0;
    hgt[side] = 1 + max(hgt[other], stays_h);
    height[c] = (uint8_t)hgt[side];
    subtree[c] = 1u + (link[other] >= 0 ? subtree[link[other]] : 0u) + (stays_link >= 0 ? subtree[stays_link] : 0u);
    box[side] = best_box;
    link[other] = up_link;
    box[other] = up_box;
    hgt[other] = up_link >= 0 ? (int)height[up_link] : 0;
    N.child0 = link[0]; N.x0 = box[0].x; N.y0 = box[0].y; N.z0 = box[0].z;
    N.child1 = link[1]; N.x1 = box[1].x; N.y1 = box[1].y; N.z1 = box[1].z;
    nodes[i] = N;
    atomicAdd(applied, 1u);
  }
  height[i] = (uint8_t)(1 + max(hgt[0], hgt[1]));
  subtree[i] = 1u + (link[0] >= 0 ? subtree[link[0]] : 0u) + (link[1] >= 0 ? subtree[link[1]] : 0u);
}

#endif  // rotations

// Treelet restructuring (after Karras and Aila, "Fast parallel construction of high-quality bounding volume
// hierarchies"): below every node a treelet of up to K subtrees is opened (always the inner node with the largest
// box next) and the binary tree over those subtrees with the least inner surface area - found by dynamic programming
// over their subsets - replaces the one that was there.  A rotation is the case K = 3.  One level per launch, deepest
// first: a thread owns its node and everything below it.  Trees that would exceed `limit` levels are left alone.
#ifndef PTK_LBVH_TREELET_MIN_GAIN
#define PTK_LBVH_TREELET_MIN_GAIN 0.0f
#endif
template <int K>
__global__ __launch_bounds__(kBuildBlock) void treelet_level_kernel(NodePair* nodes, const uint32_t* __restrict__ list,
                                                                    uint32_t begin, uint32_t end, int level, int limit,
                                                                    uint8_t* height, uint32_t* __restrict__ subtree, BoxGrid g,
                                                                    uint32_t* applied)
{
  uint32_t k = begin + blockIdx.x * kBuildBlock + threadIdx.x;
  if (k >= end) return;
  const uint32_t i = list[k];
  constexpr int kSets = 1 << K;
  int32_t sub[K];
  QBox sb[K];
  int32_t inner[K];
  {
    const NodePair N = nodes[i];
    sub[0] = N.child0; sb[0] = QBox{N.x0, N.y0, N.z0};
    sub[1] = N.child1; sb[1] = QBox{N.x1, N.y1, N.z1};
  }
  int n_sub = 2, n_inner = 1;
  inner[0] = (int32_t)i;
  float current = 0.0f;  // surface area of the treelet's inner nodes below its root
  while (n_sub < K) {
    int open = -1;
    float largest = -1.0f;
    for (int j = 0; j < n_sub; j++) {
      if (sub[j] < 0) continue;
      const float a = q_half_area(sb[j], g);
      if (a > largest) { largest = a; open = j; }
    }
    if (open < 0) break;
    const int32_t c = sub[open];
    const NodePair C = nodes[c];
    inner[n_inner++] = c;
    current += largest;
    sub[open] = C.child0; sb[open] = QBox{C.x0, C.y0, C.z0};
    sub[n_sub] = C.child1; sb[n_sub] = QBox{C.x1, C.y1, C.z1};
    n_sub++;
  }
  bool changed = false;
  if (n_sub >= 3) {
    const uint32_t full = (1u << n_sub) - 1u;
    QBox ub[kSets];
    float cost[kSets];
    uint8_t split[kSets];
    for (uint32_t m = 1; m <= full; m++) {
      const int low = __ffs((int)m) - 1;
      const uint32_t rest = m & (m - 1u);
      if (rest == 0u) { ub[m] = sb[low]; cost[m] = 0.0f; split[m] = 0; continue; }
      ub[m] = QBox{q_union(ub[rest].x, sb[low].x), q_union(ub[rest].y, sb[low].y), q_union(ub[rest].z, sb[low].z)};
      float best = 3.0e38f;
      uint32_t best_part = 0u;
      for (uint32_t a = rest; a != 0u; a = (a - 1u) & rest) {  // a: a part without m's lowest member
        const float c = cost[m ^ a] + cost[a];
        if (c < best) { best = c; best_part = m ^ a; }
      }
      cost[m] = best + (m == full ? 0.0f : q_half_area(ub[m], g));
      split[m] = (uint8_t)best_part;
    }
    if (cost[full] < current * (1.0f - PTK_LBVH_TREELET_MIN_GAIN)) {
      // the inner nodes of the new treelet, root first; a node's children come later in the list
      uint32_t tm[K];      // subset
      int th[K];           // levels of inner nodes from it down (filled in below)
      uint32_t ts[K];      // inner nodes in use from it down
      int td[K];           // depth inside the treelet
      int n_t = 0;
      tm[n_t] = full; td[n_t] = 0; n_t++;
      bool fits = true;
      for (int t = 0; t < n_t; t++) {
        const uint32_t parts[2] = {split[tm[t]], tm[t] ^ split[tm[t]]};
        for (uint32_t part : parts) {
          if (part & (part - 1u)) { tm[n_t] = part; td[n_t] = td[t] + 1; n_t++; }
          else {
            const int32_t link = sub[__ffs((int)part) - 1];
            const int below = link >= 0 ? (int)height[link] : 0;
            if (level + td[t] + below > limit) fits = false;  // the deepest level of inner nodes under this subtree
          }
        }
      }
      if (fits) {
        for (int t = n_t - 1; t >= 0; t--) {
          const uint32_t parts[2] = {split[tm[t]], tm[t] ^ split[tm[t]]};
          int32_t link[2];
          int below[2];
          uint32_t count[2];
          for (int c = 0; c < 2; c++) {
            if (parts[c] & (parts[c] - 1u)) {
              int u = t + 1;
              while (tm[u] != parts[c]) u++;
              link[c] = inner[u]; below[c] = th[u]; count[c] = ts[u];
            } else {
              link[c] = sub[__ffs((int)parts[c]) - 1];
              below[c] = link[c] >= 0 ? (int)height[link[c]] : 0;
              count[c] = link[c] >= 0 ? subtree[link[c]] : 0u;
            }
          }
          th[t] = 1 + max(below[0], below[1]);
          ts[t] = 1u + count[0] + count[1];
          NodePair R;
          R.x0 = ub[parts[0]].x; R.y0 = ub[parts[0]].y; R.z0 = ub[parts[0]].z; R.child0 = link[0];
          R.x1 = ub[parts[1]].x; R.y1 = ub[parts[1]].y; R.z1 = ub[parts[1]].z; R.child1 = link[1];
          nodes[inner[t]] = R;
          height[inner[t]] = (uint8_t)th[t];
          subtree[inner[t]] = ts[t];
        }
        changed = true;
        atomicAdd(applied, 1u);
      }
    }
  }
  if (!changed) {
    const NodePair N = nodes[i];
    const int h0 = N.child0 >= 0 ? (int)height[N.child0] : 0, h1 = N.child1 >= 0 ? (int)height[N.child1] : 0;
    height[i] = (uint8_t)(1 + max(h0, h1));
    subtree[i] = 1u + (N.child0 >= 0 ? subtree[N.child0] : 0u) + (N.child1 >= 0 ? subtree[N.child1] : 0u);
  }
}

// Primitives below every node, deepest level first (list[begin .. end): one level).
__global__ __launch_bounds__(kBuildBlock) void count_level_kernel(const NodePair* __restrict__ nodes,
                                                                  const uint32_t* __restrict__ list, uint32_t begin,
                                                                  uint32_t end, uint32_t* __restrict__ prims_below)
{
  uint32_t k = begin + blockIdx.x * kBuildBlock + threadIdx.x;
  if (k >= end) return;
  const uint32_t i = list[k];
  const int32_t c0 = nodes[i].child0, c1 = nodes[i].child1;
  prims_below[i] = (c0 >= 0 ? prims_below[c0] : leaf_count(c0)) + (c1 >= 0 ? prims_below[c1] : leaf_count(c1));
}

// Final positions, top-down: depth-first order (a node, its first child's whole subtree, then the second
// child's), which keeps the lower levels of every subtree together in memory.  list[begin .. end): one level.
// The primitives follow the same order: a node's range starts at prim_first, its first child's primitives come first;
// leaves are re-pointed at their new ranges and new_order says which primitive goes where.  (The sort order was the
// order of the leaves as long as the tree was the one the splits had made; rotations and regrouping moved subtrees.)
__global__ __launch_bounds__(kBuildBlock) void number_level_kernel(NodePair* nodes, const uint32_t* __restrict__ list,
                                                                   const uint32_t* __restrict__ subtree, uint32_t begin,
                                                                   uint32_t end, uint32_t* new_index,
                                                                   const uint32_t* __restrict__ prims_below, uint32_t* prim_first,
                                                                   const uint32_t* __restrict__ order, uint32_t* __restrict__ new_order)
{
  uint32_t k = begin + blockIdx.x * kBuildBlock + threadIdx.x;
  if (k >= end) return;
  const uint32_t i = list[k];
  const uint32_t at = new_index[i];
  int32_t c[2] = {nodes[i].child0, nodes[i].child1};
  if (c[0] >= 0) new_index[c[0]] = at + 1u;
  if (c[1] >= 0) new_index[c[1]] = at + 1u + (c[0] >= 0 ? subtree[c[0]] : 0u);
  uint32_t first = prim_first[i];
  for (int j = 0; j < 2; j++) {
    if (c[j] >= 0) { prim_first[c[j]] = first; first += prims_below[c[j]]; continue; }
    const uint32_t old_first = leaf_first(c[j]), count = leaf_count(c[j]);
    for (uint32_t p = 0; p < count; p++) new_order[first + p] = order[old_first + p];
    c[j] = leaf_link(first, count, leaf_has_spheres(c[j]));
    first += count;
  }
  nodes[i].child0 = c[0];
  nodes[i].child1 = c[1];
}

// Nodes in use move to their final, dense positions; inner links follow.
__global__ __launch_bounds__(kBuildBlock) void compact_kernel(const NodePair* __restrict__ in, const uint32_t* __restrict__ live,
                                                              const uint32_t* __restrict__ new_index, uint32_t n_nodes,
                                                              NodePair* __restrict__ out)
{
  uint32_t i = blockIdx.x * kBuildBlock + threadIdx.x;
  if (i >= n_nodes || !live[i]) return;
  NodePair nd = in[i];
  if (nd.child0 >= 0) nd.child0 = (int32_t)new_index[nd.child0];
  if (nd.child1 >= 0) nd.child1 = (int32_t)new_index[nd.child1];
  out[new_index[i]] = nd;
}

__global__ __launch_bounds__(kBuildBlock) void gather_kernel(const PrimRec* __restrict__ by_id,
                                                             const uint32_t* __restrict__ order, int32_t n,
                                                             PrimRec* __restrict__ out)
{
  int32_t i = blockIdx.x * kBuildBlock + threadIdx.x;
  if (i >= n) return;
  out[i] = by_id[order[i]];
}

struct Scratch {  // frees whatever was allocated when the build leaves, whichever way
  std::vector<void*> ptrs;
  ~Scratch() { for (void* p : ptrs) if (p) (void)hipFree(p); }
  hipError_t alloc(void** p, size_t bytes)
  {
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) ptrs.push_back(*p);
    return e;
  }
};

inline unsigned blocks_for(uint32_t n) { return (n + kBuildBlock - 1) / kBuildBlock; }

}  // namespace

#define LBVH_HIP(call)                                                                   \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) return std::string(#call) + ": " + hipGetErrorString(e_);      \
  } while (0)

std::string build_device_tree(hipStream_t stream, const float* d_boxes, const PrimRec* d_prims_by_id, int32_t n, int32_t n_triangles,
                              const float centroid_lo[3], const float centroid_hi[3], float max_diagonal,
                              const BoxGrid& grid, DeviceTree& out)
{
  out = DeviceTree();
  if (n < 2 * kMaxLeafSize) return "device tree build needs at least " + std::to_string(2 * kMaxLeafSize) + " primitives";
  if ((int64_t)n > ((int64_t)kMaxLeafSize << kMaxTreeDepth)) return "too many primitives for the depth limit";
  const uint32_t un = (uint32_t)n;
  Scratch scratch;
  uint64_t *keys_in = nullptr, *keys = nullptr;
  uint32_t *order_in = nullptr, *order = nullptr, *counter = nullptr;
  BuildRange* ranges = nullptr;
  NodePair* nodes_full = nullptr;
  void* sort_temp = nullptr;
  LBVH_HIP(scratch.alloc((void**)&keys_in, un * sizeof(uint64_t)));
  LBVH_HIP(scratch.alloc((void**)&keys, un * sizeof(uint64_t)));
  LBVH_HIP(scratch.alloc((void**)&order_in, un * sizeof(uint32_t)));
  LBVH_HIP(scratch.alloc((void**)&order, un * sizeof(uint32_t)));
  LBVH_HIP(scratch.alloc((void**)&counter, sizeof(uint32_t)));
  LBVH_HIP(scratch.alloc((void**)&ranges, un * sizeof(BuildRange)));     // fewer than n inner nodes
  LBVH_HIP(scratch.alloc((void**)&nodes_full, un * sizeof(NodePair)));
  size_t temp_bytes = 0;
  LBVH_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_in, keys, order_in, order, (size_t)un, 0u, 64u, stream));
  LBVH_HIP(scratch.alloc(&sort_temp, temp_bytes > 0 ? temp_bytes : 16));

  hipEvent_t ev0, ev1;
  LBVH_HIP(hipEventCreate(&ev0));
  LBVH_HIP(hipEventCreate(&ev1));
  struct EventGuard {
    hipEvent_t a, b;
    ~EventGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
  } guard{ev0, ev1};
  LBVH_HIP(hipEventRecord(ev0, stream));

  KeyParams kp;
  for (int k = 0; k < 3; k++) {
    float ext = centroid_hi[k] - centroid_lo[k];
    kp.lo[k] = centroid_lo[k];
    kp.scale[k] = ext > 0.f ? 65535.0f / ext : 0.f;
  }
  kp.size_scale = max_diagonal > 0.f ? 1.0f / max_diagonal : 0.f;
  // size classes: groups of occupied size bins with at least kClassGap empty bins (a factor of 2^(kClassGap/2) in
  // size) between them, at most sixteen (further groups join the last)
  uint32_t* histogram = nullptr;
  uint8_t* size_class = nullptr;
  LBVH_HIP(scratch.alloc((void**)&histogram, kSizeBins * sizeof(uint32_t)));
  LBVH_HIP(scratch.alloc((void**)&size_class, kSizeBins));
  LBVH_HIP(hipMemsetAsync(histogram, 0, kSizeBins * sizeof(uint32_t), stream));
  size_histogram_kernel<<<blocks_for(un), kBuildBlock, 0, stream>>>(d_boxes, n, kp, histogram);
  LBVH_HIP(hipGetLastError());
  {
    uint32_t h[kSizeBins];
    uint8_t cls[kSizeBins];
    LBVH_HIP(hipMemcpyAsync(h, histogram, sizeof h, hipMemcpyDeviceToHost, stream));
    LBVH_HIP(hipStreamSynchronize(stream));
    constexpr int kClassGap = 4;
    int current = 0, empty_run = 0;
    bool seen = false;
    for (int b = 0; b < kSizeBins; b++) {
      if (h[b] == 0u) { empty_run++; cls[b] = (uint8_t)current; continue; }
      if (seen && empty_run >= kClassGap && current < 15) current++;
      seen = true;
      empty_run = 0;
      cls[b] = (uint8_t)current;
    }
    LBVH_HIP(hipMemcpyAsync(size_class, cls, sizeof cls, hipMemcpyHostToDevice, stream));
    LBVH_HIP(hipStreamSynchronize(stream));  // cls lives on this stack frame
  }
  for (int k = 0; k < 3; k++) kp.scale[k] *= 16.0f;  // 20 bits per axis
  key_kernel<<<blocks_for(un), kBuildBlock, 0, stream>>>(d_boxes, n, kp, size_class, keys_in, order_in);
  LBVH_HIP(hipGetLastError());
  LBVH_HIP(rocprim::radix_sort_pairs(sort_temp, temp_bytes, keys_in, keys, order_in, order, (size_t)un, 0u, 64u, stream));

  // root: node 0 holds the whole sequence
  const BuildRange root{0u, un};
  const uint32_t one = 1u;
  LBVH_HIP(hipMemcpyAsync(ranges, &root, sizeof root, hipMemcpyHostToDevice, stream));
  LBVH_HIP(hipMemcpyAsync(counter, &one, sizeof one, hipMemcpyHostToDevice, stream));
  std::vector<std::pair<uint32_t, uint32_t>> levels;
  uint32_t begin = 0, end = 1;
  int depth = 0;
  while (begin < end) {
    if (depth >= kMaxTreeDepth) return "internal: device tree deeper than the traversal stack allows";
    split_level_kernel<<<blocks_for(end - begin), kBuildBlock, 0, stream>>>(keys, ranges, nodes_full, begin, end, depth, counter, d_boxes, order);
    LBVH_HIP(hipGetLastError());
    uint32_t total = 0;
    LBVH_HIP(hipMemcpyAsync(&total, counter, sizeof total, hipMemcpyDeviceToHost, stream));
    LBVH_HIP(hipStreamSynchronize(stream));
    if (total < end || total >= un) return "internal: device tree node count out of range";
    levels.emplace_back(begin, end);
    begin = end;
    end = total;
    depth++;
  }
  const uint32_t n_built = end;
  uint32_t *live = nullptr, *new_index = nullptr, *subtree = nullptr;
  LBVH_HIP(scratch.alloc((void**)&live, n_built * sizeof(uint32_t)));
  LBVH_HIP(scratch.alloc((void**)&new_index, n_built * sizeof(uint32_t)));
  LBVH_HIP(scratch.alloc((void**)&subtree, n_built * sizeof(uint32_t)));
  LBVH_HIP(hipMemsetAsync(live, 0, n_built * sizeof(uint32_t), stream));
  LBVH_HIP(hipMemcpyAsync(live, &one, sizeof one, hipMemcpyHostToDevice, stream));  // the root
  LBVH_HIP(hipMemsetAsync(new_index, 0, sizeof(uint32_t), stream));                 // stays node 0
  for (size_t l = levels.size(); l-- > 0;) {
    fit_level_kernel<<<blocks_for(levels[l].second - levels[l].first), kBuildBlock, 0, stream>>>(
        d_boxes, order, nodes_full, levels[l].first, levels[l].second, live, subtree, grid, (uint32_t)n_triangles);
    LBVH_HIP(hipGetLastError());
  }
  // the nodes in use by level (merging has dropped some; rotations move subtrees between levels)
  uint32_t* list = nullptr;
  uint8_t* height = nullptr;
  uint32_t* applied = nullptr;
  LBVH_HIP(scratch.alloc((void**)&list, n_built * sizeof(uint32_t)));
  LBVH_HIP(scratch.alloc((void**)&height, n_built * sizeof(uint8_t)));
  LBVH_HIP(scratch.alloc((void**)&applied, sizeof(uint32_t)));
  std::vector<std::pair<uint32_t, uint32_t>> lists;  // positions in `list`
  auto relevel = [&]() -> std::string {
    lists.clear();
    const uint32_t zero = 0u;
    LBVH_HIP(hipMemcpyAsync(list, &zero, sizeof zero, hipMemcpyHostToDevice, stream));  // the root
    LBVH_HIP(hipMemcpyAsync(counter, &one, sizeof one, hipMemcpyHostToDevice, stream));
    uint32_t lb = 0, le = 1;
    while (lb < le) {
      if ((int)lists.size() >= kMaxTreeDepth) return "internal: device tree deeper than the traversal stack allows";
      expand_level_kernel<<<blocks_for(le - lb), kBuildBlock, 0, stream>>>(nodes_full, list, lb, le, counter);
      LBVH_HIP(hipGetLastError());
      uint32_t total = 0;
      LBVH_HIP(hipMemcpyAsync(&total, counter, sizeof total, hipMemcpyDeviceToHost, stream));
      LBVH_HIP(hipStreamSynchronize(stream));
      if (total < le || total > n_built) return "internal: device tree level lists out of range";
      lists.emplace_back(lb, le);
      lb = le;
      le = total;
    }
    return std::string();
  };
  {
    std::string why = relevel();
    if (!why.empty()) return why;
  }
  const int depth_limit = (int)lists.size();  // rotations keep the tree as deep as it is
  for (int pass = 0; pass < kRotationPasses; pass++) {
    LBVH_HIP(hipMemsetAsync(applied, 0, sizeof(uint32_t), stream));
    for (size_t l = lists.size(); l-- > 0;) {
#if PTK_LBVH_TREELET >= 3
      treelet_level_kernel<PTK_LBVH_TREELET><<<blocks_for(lists[l].second - lists[l].first), kBuildBlock, 0, stream>>>(
          nodes_full, list, lists[l].first, lists[l].second, (int)l + 1, depth_limit, height, subtree, grid, applied);
#else
      rotate_level_kernel<<<blocks_for(lists[l].second - lists[l].first), kBuildBlock, 0, stream>>>(
          nodes_full, list, lists[l].first, lists[l].second, (int)l + 1, depth_limit, height, subtree, grid, applied);
#endif
      LBVH_HIP(hipGetLastError());
    }
    uint32_t n_applied = 0;
    LBVH_HIP(hipMemcpyAsync(&n_applied, applied, sizeof n_applied, hipMemcpyDeviceToHost, stream));
    LBVH_HIP(hipStreamSynchronize(stream));
    if (std::getenv("PTK_LBVH_REPORT")) fprintf(stderr, "[ptk lbvh] restructuring pass %d: %u changes\n", pass, n_applied);
    if (n_applied == 0u) break;
    std::string why = relevel();
    if (!why.empty()) return why;
  }
  uint32_t *prims_below = nullptr, *prim_first = nullptr;
  LBVH_HIP(scratch.alloc((void**)&prims_below, n_built * sizeof(uint32_t)));
  LBVH_HIP(scratch.alloc((void**)&prim_first, n_built * sizeof(uint32_t)));
  LBVH_HIP(hipMemsetAsync(prim_first, 0, sizeof(uint32_t), stream));  // the root's range starts at 0
  for (size_t l = lists.size(); l-- > 0;) {
    count_level_kernel<<<blocks_for(lists[l].second - lists[l].first), kBuildBlock, 0, stream>>>(
        nodes_full, list, lists[l].first, lists[l].second, prims_below);
    LBVH_HIP(hipGetLastError());
  }
  uint32_t* new_order = order_in;  // (the unsorted order is no longer needed)
  for (size_t l = 0; l < lists.size(); l++) {
    number_level_kernel<<<blocks_for(lists[l].second - lists[l].first), kBuildBlock, 0, stream>>>(
        nodes_full, list, subtree, lists[l].first, lists[l].second, new_index, prims_below, prim_first, order, new_order);
    LBVH_HIP(hipGetLastError());
  }
  uint32_t n_nodes = 0;
  LBVH_HIP(hipMemcpyAsync(&n_nodes, subtree, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  LBVH_HIP(hipStreamSynchronize(stream));
  if (n_nodes < 1 || n_nodes > n_built) return "internal: device tree compaction out of range";
  PrimRec* prims = nullptr;
  NodePair* nodes = nullptr;
  LBVH_HIP(hipMalloc((void**)&prims, (size_t)un * sizeof(PrimRec)));
  hipError_t e = hipMalloc((void**)&nodes, (size_t)n_nodes * sizeof(NodePair));
  if (e != hipSuccess) { (void)hipFree(prims); return std::string("hipMalloc(nodes): ") + hipGetErrorString(e); }
  compact_kernel<<<blocks_for(n_built), kBuildBlock, 0, stream>>>(nodes_full, live, new_index, n_built, nodes);
  e = hipGetLastError();
  if (e != hipSuccess) { (void)hipFree(prims); (void)hipFree(nodes); return std::string("compact_kernel: ") + hipGetErrorString(e); }
  gather_kernel<<<blocks_for(un), kBuildBlock, 0, stream>>>(d_prims_by_id, new_order, n, prims);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipEventRecord(ev1, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, ev0, ev1);
  if (e != hipSuccess) {
    (void)hipFree(prims);
    (void)hipFree(nodes);
    return std::string("device tree build: ") + hipGetErrorString(e);
  }
  if (std::getenv("PTK_LBVH_REPORT")) {  // diagnostic: leaves by size and by level of the finished tree
    std::vector<NodePair> h(n_nodes);
    if (hipMemcpy(h.data(), nodes, (size_t)n_nodes * sizeof(NodePair), hipMemcpyDeviceToHost) == hipSuccess) {
      unsigned long long by_size[9] = {0}, by_level[kMaxTreeDepth + 2] = {0}, prims_by_level[kMaxTreeDepth + 2] = {0};
      std::vector<std::pair<int32_t, int>> todo{{0, 1}};
      while (!todo.empty()) {
        auto [node, level] = todo.back();
        todo.pop_back();
        for (int32_t link : {h[node].child0, h[node].child1}) {
          if (link >= 0) { todo.emplace_back(link, level + 1); continue; }
          const uint32_t count = leaf_count(link);
          by_size[count]++;
          by_level[level < kMaxTreeDepth + 1 ? level : kMaxTreeDepth + 1]++;
          prims_by_level[level < kMaxTreeDepth + 1 ? level : kMaxTreeDepth + 1] += count;
        }
      }
      fprintf(stderr, "[ptk lbvh] %u nodes; leaves of 1..8 primitives:", n_nodes);
      for (int k = 1; k <= 8; k++) fprintf(stderr, " %llu", by_size[k]);
      fprintf(stderr, "\n[ptk lbvh] leaves (primitives) by level:");
      for (int l = 1; l <= kMaxTreeDepth + 1; l++) if (by_level[l]) fprintf(stderr, " %d:%llu(%llu)", l, by_level[l], prims_by_level[l]);
      fprintf(stderr, "\n");
    }
  }
  out.nodes = nodes;
  out.prims = prims;
  out.n_nodes = (int32_t)n_nodes;
  out.max_depth = (int32_t)lists.size();  // leaves of the deepest inner level
  out.max_leaf = kMaxLeafSize;
  out.build_ms = ms;
  return std::string();
}

}  // namespace ptk
