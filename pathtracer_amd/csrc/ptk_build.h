// Host-side scene flattening: turns the C-ABI scene description into the
// device layout of ptk_layout.h.  Stands where Scene::compute_bvh ->
// BVH::construct (src/scene.cpp:80, src/bvh.cpp:54-83) stands in the
// reference, but builds a different tree: binned surface-area-heuristic
// splits, at most kMaxLeafSize primitives per leaf, depth capped so that the
// per-lane traversal stack can never overflow.  The closest hit of a ray does
// not depend on the tree, apart from the reference's own box-test quirks
// (0.01 entry distance and rejection of zero-thickness boxes,
// src/bvh.cpp:12, src/aabb.cpp:62).
#pragma once

#include <string>
#include <vector>

#include "../../include/ptk.h"
#include "ptk_layout.h"

namespace ptk {

struct HostScene {
  std::vector<NodePair> nodes;
  std::vector<PrimRec> prims;       // BVH leaf order
  std::vector<ShadeRec> shade;      // by primitive id
  std::vector<MaterialRec> materials;
  std::vector<LightRec> lights;     // by primitive id, ascending (Scene::m_lights order, src/scene.cpp:60)
  std::vector<TextureRec> textures;
  std::vector<uint8_t> texels;
  int32_t background_texture = -1;
  int32_t background_is_color = 0;
  float background_color[3] = {-1.f, -1.f, -1.f};
  int32_t n_prims = 0;
  int32_t n_triangles = 0;
  BoxGrid grid = {{0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}};  // see NodePair
  int32_t max_depth = 0;
  int32_t max_leaf = 0;
  bool tree_from_cache = false;  // the tree was read from the on-disk cache instead of being built
  // Filled instead of `nodes` when the tree is left to the device builder: `prims` is then in id order and
  // these are the padded boxes (lo.xyz, hi.xyz per primitive) the tree has to be built over.
  std::vector<float> prim_boxes;
  float centroid_lo[3] = {0.f, 0.f, 0.f}, centroid_hi[3] = {0.f, 0.f, 0.f};  // bounds of the box centres
  float max_diagonal = 0.f;                                                  // longest box diagonal

  // Pointers into this object's own vectors (host memory).
  SceneView host_view() const;
};

// Returns an empty string on success, otherwise the reason the description was rejected.
// with_tree = false: everything except the tree (see HostScene::prim_boxes); scenes of fewer than
// kDeviceBuildMin primitives get their tree here regardless.
//
// Tree cache: with the environment variable PTK_BVH_CACHE naming a directory, a host-built tree is stored
// there under a 128-bit hash of the primitives' boxes and the builder's limits, and a later build of the
// same geometry reads it back instead of building (files are checked in full before use: sizes, link
// ranges, depth, and that the leaf order is a permutation).
constexpr int kDeviceBuildMin = 64;
std::string build_host_scene(const ptk_scene_desc& desc, HostScene& out, bool with_tree = true);

CameraView make_camera_view(const ptk_camera& cam);
FlagView make_flag_view(const ptk_flags* flags);

}  // namespace ptk
