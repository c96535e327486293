// Wavefront OBJ / MTL reader for Scene::load_obj.  Produces what the reference gets out of its vendored
// tinyobjloader 2.0.0 for the constructs its assets use: v / vt / vn / f / usemtl / mtllib (o, g, s are
// ignored), faces triangulated (quads split like src/tiny_obj_loader.h:1587-1605, larger polygons as a
// fan), numbers parsed to float (tinyobj's real_t), negative (relative) indices, and the MTL keys
// Kd, Ke, illum, Pr, Pm, map_Kd with tinyobj's defaults (illum 0, Pr = Pm = 0, Ke = 0).
#pragma once

#include <filesystem>
#include <string>
#include <vector>

namespace pt {

struct ObjMaterial {
  std::string name;
  float diffuse[3] = {0.f, 0.f, 0.f};
  float emission[3] = {0.f, 0.f, 0.f};
  float roughness = 0.f, metallic = 0.f;
  int illum = 0;
  std::string diffuse_texname;
};

struct ObjVertex {
  float pos[3] = {0, 0, 0};
  float normal[3] = {0, 0, 0};
  float uv[2] = {0, 0};
  bool has_normal = false, has_uv = false;
};

struct ObjMesh {
  std::vector<ObjVertex> vertices;  // three per triangle, file order
  std::vector<int> material_ids;    // one per triangle, -1 = none
  std::vector<ObjMaterial> materials;
  std::string warning;
};

// Returns false and fills `error` when the file cannot be read or parsed.
bool load_obj_file(const std::filesystem::path& filename, const std::filesystem::path& mtl_search_path, ObjMesh& out,
                   std::string& error);

}  // namespace pt
