// Scene on top of the C ABI (include/ptk.h).
#include "scene.h"

#include <chrono>
#include <cstdlib>
#include <iostream>

#include "../../include/ptk.h"
#include "obj_loader.h"

AABB::AABB(const Sphere& s) : min(s.center - pt::dvec3(s.radius)), max(s.center + pt::dvec3(s.radius)) {}  // src/aabb.cpp:11-15
AABB::AABB(const Triangle& t) : min(pt::min(pt::min(t.v0, t.v1), t.v2)), max(pt::max(pt::max(t.v0, t.v1), t.v2)) {}  // src/aabb.cpp:17-21

AABB merge(const AABB& a, const AABB& b)
{
  AABB bb;
  bb.min = pt::min(a.min, b.min);
  bb.max = pt::max(a.max, b.max);
  return bb;
}

// src/aabb.cpp:31-43
AABB compute_bounding_volume(const std::vector<Primitive>::const_iterator& begin,
                             const std::vector<Primitive>::const_iterator& end)
{
  AABB bbox;
  bbox.min = pt::dvec3(+1e9);
  bbox.max = pt::dvec3(-1e9);
  for (auto it = begin; it != end; it++) {
    bbox.min = pt::min(bbox.min, it->bbox.min);
    bbox.max = pt::max(bbox.max, it->bbox.max);
  }
  return bbox;
}

// src/geometry.cpp:244-254
double Primitive::sample_area() const
{
  if (type == Type::TRIANGLE) return 0.5 * pt::length(pt::cross(triangle.v1 - triangle.v0, triangle.v2 - triangle.v0));
  return (4.0 * 3.14159265359 * sphere.radius * sphere.radius) / 2.0;
}

Scene::Scene() : m_count(0U), m_background_texture(nullptr), m_background_color(-1.0) {}

void register_stats_scene(const Scene* s);

Scene::~Scene()
{
  register_stats_scene(nullptr);
  if (m_group) ptk_group_destroy(m_group);
}

void Scene::set_background_texture(std::unique_ptr<Image> texture) { m_background_texture = std::move(texture); }

void Scene::set_background_color(const pt::dvec3& color) { m_background_color = color; }

int Scene::light_count() const { return (int)m_lights.size(); }

int Scene::primitive_count() { return (int)m_primitives.size(); }

std::vector<Primitive> Scene::lights() const { return m_lights; }

pt::dvec3 Scene::center() const { return m_bounds.center(); }

pt::dvec3 Scene::size() const { return m_bounds.size(); }

// src/scene.cpp:49-54; the reference asserts on overflow, here the process ends with a message
Material* Scene::add_material(const Material& m)
{
  if (m_material_count >= m_materials.size()) {
    std::cerr << "Scene::add_material: more than " << m_materials.size() << " materials" << std::endl;
    std::exit(1);
  }
  m_materials[m_material_count] = m;
  return &m_materials[m_material_count++];
}

// src/scene.cpp:56-62: ids in insertion order, emissive primitives join the light list
void Scene::add_primitive(const Primitive& p)
{
  Primitive p_new = p;
  p_new.id = m_count++;
  if (p_new.is_light()) m_lights.push_back(p_new);
  m_primitives.push_back(p_new);
}

void Scene::add_primitives(const std::vector<Primitive>::iterator begin, const std::vector<Primitive>::iterator end)
{
  m_primitives.reserve(m_primitives.size() + (size_t)(end - begin));
  for (auto it = begin; it != end; it++) add_primitive(*it);
}

// src/scene.cpp:93-245
std::vector<Primitive> Scene::load_obj(const std::filesystem::path& filename)
{
  std::cout << __FUNCTION__ << " Filename: " << filename.string() << std::endl;
  std::filesystem::path mtl_search_path = filename.parent_path();
  std::cout << __FUNCTION__ << " mtl search path: " << mtl_search_path.string() << std::endl;

  pt::ObjMesh mesh;
  std::string error;
  const bool timing = std::getenv("PT_TIMING") != nullptr;  // prints where a large file's load time goes
  auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    auto t1 = std::chrono::steady_clock::now();
    std::cerr << "[pt load_obj] " << what << ": " << std::chrono::duration<double>(t1 - t0).count() << " s" << std::endl;
    t0 = t1;
  };
  if (!pt::load_obj_file(filename, mtl_search_path, mesh, error)) {
    if (!error.empty()) std::cerr << __FUNCTION__ << " " << error;
    std::exit(1);  // as the reference does (src/scene.cpp:103-108)
  }
  lap("parse");
  if (!mesh.warning.empty()) std::cout << __FUNCTION__ << " " << mesh.warning;
  std::cout << __FUNCTION__ << " Materials: " << mesh.materials.size() << std::endl;

  auto offset = m_material_count;
  for (const pt::ObjMaterial& m : mesh.materials) {
    Material material;
    switch (m.illum) {  // src/scene.cpp:127-147
      case 3: case 8: material.type = Material::SPECULAR; break;
      case 5: material.type = Material::MICROFACET; break;
      case 4: case 6: case 7: case 9: material.type = Material::DIELECTRIC; break;
      default: material.type = Material::DIFFUSE; break;
    }
    material.albedo = pt::dvec3(m.diffuse[0], m.diffuse[1], m.diffuse[2]);
    material.emission = pt::dvec3(m.emission[0], m.emission[1], m.emission[2]);
    material.roughness = m.roughness;
    material.metallic = m.metallic;
    if (!m.diffuse_texname.empty()) {
      auto diffuse_texname = mtl_search_path / std::filesystem::path(m.diffuse_texname);
      Image* texture = new Image();  // lives as long as the process, like the reference's (src/scene.cpp:157)
      if (texture->load(diffuse_texname)) {
        std::cout << __FUNCTION__ << " Loaded texture " << diffuse_texname << " (" << texture->width() << ", "
                  << texture->height() << ", " << texture->channels() << ")" << std::endl;
      } else {
        std::cerr << __FUNCTION__ << " Failed to load " << diffuse_texname << std::endl;
      }
      material.texture = texture;
    }
    (void)add_material(material);
  }

  std::vector<Primitive> triangles;
  Material default_albedo;
  default_albedo.albedo = pt::dvec3(0.5);
  Material* default_material = add_material(default_albedo);  // src/scene.cpp:215

  size_t triangle_count = mesh.vertices.size() / 3;
  triangles.reserve(triangle_count);
  for (size_t i = 0; i < triangle_count; i++) {
    const pt::ObjVertex* v = &mesh.vertices[i * 3];
    Triangle tri;
    tri.v0 = pt::dvec3(v[0].pos[0], v[0].pos[1], v[0].pos[2]);
    tri.v1 = pt::dvec3(v[1].pos[0], v[1].pos[1], v[1].pos[2]);
    tri.v2 = pt::dvec3(v[2].pos[0], v[2].pos[1], v[2].pos[2]);
    tri.t0 = pt::dvec2(v[0].uv[0], v[0].uv[1]);
    tri.t1 = pt::dvec2(v[1].uv[0], v[1].uv[1]);
    tri.t2 = pt::dvec2(v[2].uv[0], v[2].uv[1]);
    tri.n0 = pt::dvec3(v[0].normal[0], v[0].normal[1], v[0].normal[2]);
    tri.n1 = pt::dvec3(v[1].normal[0], v[1].normal[1], v[1].normal[2]);
    tri.n2 = pt::dvec3(v[2].normal[0], v[2].normal[1], v[2].normal[2]);
    int mid = mesh.material_ids[i];
    if (mesh.materials.empty() || mid < 0) {
      // The reference indexes m_materials[offset - 1] for faces without usemtl (src/scene.cpp:234), i.e.
      // whatever material was added last before this file; the file's default material is used instead.
      triangles.push_back(Primitive(tri, default_material));
    } else {
      triangles.push_back(Primitive(tri, &m_materials[offset + (size_t)mid]));
    }
  }
  lap("primitives");
  std::cout << __FUNCTION__ << " Triangles: " << triangles.size() << std::endl;
  return triangles;
}

// src/scene.cpp:80: after this the scene is immutable; here it is also the upload to the GPU
void Scene::compute_bvh()
{
  m_bounds = compute_bounding_volume(m_primitives.begin(), m_primitives.end());
  if (!m_group) {
    std::vector<int> devices;
    if (const char* e = std::getenv("PT_DEVICES")) {  // "0,1,2,3"
      for (const char* q = e; *q;) {
        char* end = nullptr;
        long v = std::strtol(q, &end, 10);
        if (end == q) break;
        devices.push_back((int)v);
        q = (*end == ',') ? end + 1 : end;
        if (*end != ',' && *end != 0) break;
      }
    }
    if (devices.empty()) {
      const char* e = std::getenv("PT_DEVICE");
      devices.push_back(e ? std::atoi(e) : 0);
    }
    if (const char* e = std::getenv("PT_SPLIT")) m_split = (std::string(e) == "tile") ? PTK_SPLIT_BANDS : PTK_SPLIT_SAMPLES;
    if (ptk_group_create(devices.data(), (int)devices.size(), &m_group) != PTK_OK) {
      std::cerr << "Scene::compute_bvh: " << ptk_group_last_error(nullptr) << std::endl;
      std::exit(1);
    }
    m_ctx = ptk_group_context(m_group, 0);
  }
  std::vector<double> tri_pos, tri_nrm, tri_uv, sph_c, sph_r;
  std::vector<int32_t> tri_mat, sph_mat;
  // primitive ids must stay what add_primitive assigned; the C ABI numbers triangles first, then spheres,
  // which is the reference's own order (models before spheres, src/main.cpp:117-145)
  bool seen_sphere = false;
  tri_pos.reserve(m_primitives.size() * 9); tri_nrm.reserve(m_primitives.size() * 9);
  tri_uv.reserve(m_primitives.size() * 6); tri_mat.reserve(m_primitives.size());
  for (const Primitive& p : m_primitives) {
    int mat = (int)(p.material - &m_materials[0]);
    if (p.type == Primitive::TRIANGLE) {
      if (seen_sphere) {
        std::cerr << "Scene::compute_bvh: triangles must be added before spheres" << std::endl;
        std::exit(1);
      }
      const Triangle& t = p.triangle;
      for (const pt::dvec3* q : {&t.v0, &t.v1, &t.v2}) { tri_pos.push_back(q->x); tri_pos.push_back(q->y); tri_pos.push_back(q->z); }
      for (const pt::dvec3* q : {&t.n0, &t.n1, &t.n2}) { tri_nrm.push_back(q->x); tri_nrm.push_back(q->y); tri_nrm.push_back(q->z); }
      for (const pt::dvec2* q : {&t.t0, &t.t1, &t.t2}) { tri_uv.push_back(q->x); tri_uv.push_back(q->y); }
      tri_mat.push_back(mat);
    } else {
      seen_sphere = true;
      sph_c.push_back(p.sphere.center.x); sph_c.push_back(p.sphere.center.y); sph_c.push_back(p.sphere.center.z);
      sph_r.push_back(p.sphere.radius);
      sph_mat.push_back(mat);
    }
  }
  std::vector<const Image*> images;
  auto texture_index = [&](const Image* im) -> int {
    if (!im || !im->valid()) return -1;
    for (size_t i = 0; i < images.size(); i++)
      if (images[i] == im) return (int)i;
    images.push_back(im);
    return (int)images.size() - 1;
  };
  std::vector<ptk_material> mats(m_material_count);
  for (size_t i = 0; i < m_material_count; i++) {
    const Material& m = m_materials[i];
    ptk_material& o = mats[i];
    o.type = (int32_t)m.type;
    o.texture = texture_index(m.texture);
    o.albedo[0] = m.albedo.x; o.albedo[1] = m.albedo.y; o.albedo[2] = m.albedo.z;
    o.emission[0] = m.emission.x; o.emission[1] = m.emission.y; o.emission[2] = m.emission.z;
    o.refraction_index = m.refraction_index; o.roughness = m.roughness; o.metallic = m.metallic;
  }
  int background = texture_index(m_background_texture.get());
  std::vector<ptk_texture> texs(images.size());
  for (size_t i = 0; i < images.size(); i++) {
    texs[i].rgb = images[i]->data();
    texs[i].width = images[i]->width();
    texs[i].height = images[i]->height();
  }
  ptk_scene_desc d{};
  d.n_triangles = (int32_t)tri_mat.size();
  d.tri_pos = tri_pos.data(); d.tri_nrm = tri_nrm.data(); d.tri_uv = tri_uv.data(); d.tri_material = tri_mat.data();
  d.n_spheres = (int32_t)sph_mat.size();
  d.sphere_center = sph_c.data(); d.sphere_radius = sph_r.data(); d.sphere_material = sph_mat.data();
  d.n_materials = (int32_t)mats.size();
  d.materials = mats.data();
  d.n_textures = (int32_t)texs.size();
  d.textures = texs.data();
  d.background_texture = background;
  d.background_color[0] = m_background_color.x; d.background_color[1] = m_background_color.y; d.background_color[2] = m_background_color.z;
  int32_t builder = PTK_BUILDER_HOST_SAH;
  if (const char* e = std::getenv("PTK_BUILDER")) builder = (std::string(e) == "device") ? PTK_BUILDER_DEVICE_LBVH : PTK_BUILDER_HOST_SAH;
  if (ptk_group_upload_scene(m_group, &d, builder) != PTK_OK) {
    std::cerr << "Scene::compute_bvh: " << ptk_group_last_error(m_group) << std::endl;
    std::exit(1);
  }
  register_stats_scene(this);
}

std::vector<std::optional<Intersection>> Scene::find_intersections(const std::vector<Ray>& rays) const
{
  std::vector<std::optional<Intersection>> out(rays.size());
  if (!m_ctx || rays.empty()) return out;
  std::vector<ptk_ray> in(rays.size());
  for (size_t i = 0; i < rays.size(); i++) {
    in[i].origin[0] = rays[i].origin.x; in[i].origin[1] = rays[i].origin.y; in[i].origin[2] = rays[i].origin.z;
    in[i].direction[0] = rays[i].direction.x; in[i].direction[1] = rays[i].direction.y; in[i].direction[2] = rays[i].direction.z;
  }
  std::vector<ptk_hit> hits(rays.size());
  if (ptk_intersect_batch(m_ctx, in.data(), (int32_t)in.size(), hits.data()) != PTK_OK) {
    std::cerr << "Scene::find_intersections: " << ptk_last_error(m_ctx) << std::endl;
    std::exit(1);
  }
  Scene* self = const_cast<Scene*>(this);
  for (size_t i = 0; i < rays.size(); i++) {
    const ptk_hit& h = hits[i];
    if (!h.hit) continue;
    Intersection s;
    s.id = h.id;
    s.t = h.t;
    s.point = pt::dvec3(h.point[0], h.point[1], h.point[2]);
    s.normal = pt::dvec3(h.normal[0], h.normal[1], h.normal[2]);
    s.uv = pt::dvec2(h.uv[0], h.uv[1]);
    s.material = &self->m_materials[(size_t)h.material];
    s.inside = h.inside != 0;
    out[i] = s;
  }
  return out;
}

std::optional<Intersection> Scene::find_intersection(const Ray& ray) const
{
  return find_intersections(std::vector<Ray>{ray})[0];
}

void Scene::print_stats() const
{
  if (!m_group) return;
  ptk_stats st;  // whole-run totals over every device of the group, as the reference's global counters are
  if (ptk_group_get_stats(m_group, &st) != PTK_OK) return;
  std::cout << "Path Segments: " << st.segments << std::endl;
  std::cout << "Intersection Queries: " << st.queries << std::endl;
  if (st.prim_tests) std::cout << "Intersection Test Count: " << st.prim_tests << std::endl;
}

// print_stats (src/geometry.cpp:273-278) reports on the scene uploaded last
static const Scene* g_stats_scene = nullptr;
void register_stats_scene(const Scene* s) { g_stats_scene = s; }
void print_stats()
{
  if (g_stats_scene) g_stats_scene->print_stats();
}
