// Scene: the public surface of src/scene.h:14-43 on top of the GPU kernels.
//
// The reference keeps primitives, materials and a pointer-based BVH in host memory and answers every
// find_intersection call on the calling thread.  Here compute_bvh() is the upload point: it flattens the
// scene, builds the acceleration structure and sends everything to the GPU (ptk_upload_scene); queries
// then run there.  Lifecycle is the reference's: add everything, call compute_bvh() once, then render.
#pragma once

#include <array>
#include <cstdint>
#include <filesystem>
#include <memory>
#include <optional>
#include <vector>

#include "geometry.h"
#include "image.h"
#include "material.h"
#include "ray.h"
#include "vec.h"

struct ptk_ctx;
struct ptk_group;

class Scene
{
 public:
  Scene();
  ~Scene();
  Scene(const Scene&) = delete;
  Scene& operator=(const Scene&) = delete;

  void compute_bvh();
  void add_primitive(const Primitive& p);
  void add_primitives(const std::vector<Primitive>::iterator begin, const std::vector<Primitive>::iterator end);
  Material* add_material(const Material& m);
  std::vector<Primitive> load_obj(const std::filesystem::path& filename);
  // One closest-hit query (a batch of one on the GPU; use find_intersections for many rays).
  std::optional<Intersection> find_intersection(const Ray&) const;
  std::vector<std::optional<Intersection>> find_intersections(const std::vector<Ray>& rays) const;
  int primitive_count();
  pt::dvec3 center() const;
  pt::dvec3 size() const;
  void set_background_texture(std::unique_ptr<Image> texture);
  void set_background_color(const pt::dvec3& color);
  int light_count() const;
  std::vector<Primitive> lights() const;

  // Where the scene lives after compute_bvh(): a group of GPUs (PT_DEVICES=0,1,... in the environment; one GPU,
  // PT_DEVICE or 0, when that is unset) and the context of the group's first GPU, which answers ray queries.
  ptk_group* group() const { return m_group; }
  ptk_ctx* context() const { return m_ctx; }
  // How Renderer::render shares a frame between the group's GPUs: PTK_SPLIT_SAMPLES unless PT_SPLIT=tile.
  int split() const { return m_split; }
  // Number of box / primitive tests etc. since the scene was uploaded (print_stats, src/geometry.cpp:273-278)
  void print_stats() const;

 private:
  uint32_t m_count;
  std::vector<Primitive> m_primitives;
  std::vector<Primitive> m_lights;
  size_t m_material_count = 0;
  std::array<Material, 256> m_materials;
  std::unique_ptr<Image> m_background_texture;
  pt::dvec3 m_background_color;
  AABB m_bounds;
  ptk_group* m_group = nullptr;
  ptk_ctx* m_ctx = nullptr;  // m_group's first context
  int m_split = 0;
};
