// On-device construction of the acceleration structure: the part of Scene::compute_bvh ->
// BVH::construct (src/scene.cpp:80, src/bvh.cpp:54-83) that the host builder of ptk_build.cpp
// otherwise does, for callers that load or change large scenes and cannot wait for a host build.
// Same node and primitive layout (ptk_layout.h), same depth and leaf-size limits, so the traversal
// kernels run unchanged; the tree is a linear BVH (primitives sorted by size class and along a space-filling
// curve, ranges split where the sort key's highest differing bit flips, small ranges by surface area, a few
// rotations at the end), not a surface-area-heuristic one.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "ptk_layout.h"

namespace ptk {

struct DeviceTree {
  NodePair* nodes = nullptr;  // hipMalloc'ed, n_nodes records; the caller owns it
  PrimRec* prims = nullptr;   // hipMalloc'ed, n primitives in leaf order; the caller owns it
  int32_t n_nodes = 0;
  int32_t max_depth = 0;
  int32_t max_leaf = 0;
  float build_ms = 0.f;       // device time from the first kernel to the last
};

// d_boxes: 6 floats (lo.xyz, hi.xyz) per primitive, d_prims_by_id: the primitives' records in id order,
// both in device memory; n >= 8; primitives n_triangles .. n - 1 are spheres (leaves that hold one are marked).  grid: the scene's box grid (NodePair).  centroid_lo/hi bound the box centres, max_diagonal is the longest box
// diagonal.  Work is enqueued on `stream` and the call returns after it has finished.
// Returns an empty string on success, otherwise what failed.
std::string build_device_tree(hipStream_t stream, const float* d_boxes, const PrimRec* d_prims_by_id, int32_t n, int32_t n_triangles,
                              const float centroid_lo[3], const float centroid_hi[3], float max_diagonal,
                              const BoxGrid& grid, DeviceTree& out);

}  // namespace ptk
