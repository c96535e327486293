// pt: the command-line driver, same arguments, JSON keys and console output as the reference's
// src/main.cpp:172-270:
//   pt <config.json> <output.png> [samples=8] [bounces=3] [batch_size=16]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include "geometry.h"
#include "image.h"
#include "json.h"
#include "material.h"
#include "renderer.h"
#include "scene.h"

using pt::Json;

struct SimpleSphere : public Sphere {
  Material::Type type;
  pt::dvec3 albedo;
  pt::dvec3 emissive;
  std::string texture;
  double metallic;
  double roughness;
  bool hidden;
};

struct Config {
  bool print_progress;
  int max_bounce;
  int samples_per_pixel;
  int batch_size;
  int image_width;
  int image_height;
  pt::dvec3 camera_position;
  pt::dvec3 camera_target;
  double camera_fov;
  double camera_aperture;
  double camera_focus_distance;
  std::vector<std::string> models;
  std::vector<SimpleSphere> spheres;
  std::string background_texture;
  pt::dvec3 background_color;
};

static pt::dvec3 vec_or(const Json& j, const std::string& key, pt::dvec3 def)
{
  if (!j.contains(key)) return def;
  const Json& v = j[key];
  return pt::dvec3(v["x"].number(), v["y"].number(), v["z"].number());
}
static double num_or(const Json& j, const std::string& key, double def) { return j.contains(key) ? j[key].number() : def; }
static bool bool_or(const Json& j, const std::string& key, bool def) { return j.contains(key) ? j[key].boolean() : def; }
static std::string str_or(const Json& j, const std::string& key, const std::string& def) { return j.contains(key) ? j[key].string() : def; }

// keys and defaults of src/main.cpp:68-87
static SimpleSphere sphere_from_json(const Json& j)
{
  SimpleSphere s;
  s.hidden = bool_or(j, "hidden", false);
  s.center = vec_or(j, "center", pt::dvec3(0.0));
  s.radius = num_or(j, "radius", 1.0);
  s.albedo = vec_or(j, "albedo", pt::dvec3(1.0));
  s.emissive = vec_or(j, "emissive", pt::dvec3(0.0));
  s.texture = str_or(j, "texture", std::string());
  s.metallic = num_or(j, "metallic", 0.5);
  s.roughness = num_or(j, "roughness", 0.5);
  std::string type = str_or(j, "type", "DIFFUSE");
  if (type == "SPECULAR") s.type = Material::SPECULAR;
  else if (type == "TRANSMISSIVE") s.type = Material::DIELECTRIC;
  else s.type = Material::DIFFUSE;
  return s;
}

// keys and defaults of src/main.cpp:89-111
static Config config_from_json(const Json& j)
{
  Config c{};
  c.print_progress = bool_or(j, "print_progress", false);
  c.image_width = (int)num_or(j, "image_width", 640);
  c.image_height = (int)num_or(j, "image_height", 480);
  c.camera_position = vec_or(j, "camera_position", pt::dvec3(0.0, 0.0, 10.0));
  c.camera_target = vec_or(j, "camera_target", pt::dvec3(0.0, 0.0, 0.0));
  c.camera_fov = num_or(j, "camera_fov", 45);
  c.camera_aperture = num_or(j, "camera_aperture", 0);
  c.camera_focus_distance = num_or(j, "camera_focus_distance", 0);
  c.background_texture = str_or(j, "background_texture", std::string());
  c.background_color = vec_or(j, "background_color", pt::dvec3(-1.0));
  if (j.contains("models"))
    for (const Json& m : j["models"].arr) c.models.push_back(m.string());
  if (j.contains("spheres"))
    for (const Json& s : j["spheres"].arr) c.spheres.push_back(sphere_from_json(s));
  return c;
}

// src/main.cpp:113-170
static std::tuple<std::unique_ptr<Scene>, std::unique_ptr<Camera>> setup_scene(const Config& config)
{
  auto scene = std::make_unique<Scene>();
  for (const std::string& path : config.models) {
    auto mesh = scene->load_obj(path);
    AABB bbox = compute_bounding_volume(mesh.begin(), mesh.end());
    std::cout << "Mesh Size: " << bbox.size() << ", Mesh Center: " << bbox.center() << std::endl;
    scene->add_primitives(mesh.begin(), mesh.end());
  }
  for (const auto& s : config.spheres) {
    if (s.hidden) continue;
    Material* material = scene->add_material(Material());
    material->type = s.type;
    material->albedo = s.albedo;
    material->emission = s.emissive;
    material->roughness = s.roughness;
    material->metallic = s.metallic;
    if (!s.texture.empty()) {
      material->texture = new Image();  // lives as long as the process
      if (material->texture->load(s.texture)) std::cout << "Loaded texture " << s.texture << std::endl;
      else std::cerr << "Failed to load texture " << s.texture << std::endl;
    }
    Primitive p(Sphere(s.center, s.radius), material);
    scene->add_primitive(p);
  }
  if (!config.background_texture.empty()) {
    auto image = std::make_unique<Image>();
    if (image->load(config.background_texture)) {
      std::cout << "Loaded environment texture " << config.background_texture << std::endl;
      scene->set_background_texture(std::move(image));
    } else {
      std::cerr << "Failed to load environment texture " << config.background_texture << std::endl;
      exit(1);
    }
  } else {
    scene->set_background_color(config.background_color);
  }
  auto camera = std::make_unique<Camera>(config.image_width, config.image_height, config.camera_fov,
                                         config.camera_aperture, config.camera_focus_distance);
  camera->look_at(config.camera_position, config.camera_target);
  std::cout << "Distance camera position to camera target: " << pt::distance(config.camera_position, config.camera_target)
            << std::endl;
  scene->compute_bvh();
  return std::make_tuple(std::move(scene), std::move(camera));
}

int main(int argc, char** argv)
{
  if (argc < 3) {
    std::cerr << "Usage: " << argv[0] << " <path to config.json> <output path> <samples> <bounces> <batch_size>" << std::endl;
    return 1;
  }
  std::filesystem::path config_path = std::filesystem::path(argv[1]);
  std::filesystem::path result_path = std::filesystem::path(argv[2]);
  std::ifstream file(config_path);
  if (!file.is_open()) {
    std::cerr << "Could not open config: " << config_path.string() << std::endl;
    return 1;
  }
  std::stringstream buffer;
  buffer << file.rdbuf();
  Config config;
  try {
    std::string text = buffer.str();
    config = config_from_json(pt::JsonParser(text).parse());
  } catch (const std::exception& e) {
    std::cerr << "Could not parse config: " << e.what() << std::endl;
    return 1;
  }
  config.samples_per_pixel = 4 <= argc ? std::atoi(argv[3]) : 8;
  config.max_bounce = 5 <= argc ? std::atoi(argv[4]) : 3;
  config.batch_size = 6 <= argc ? std::atoi(argv[5]) : 16;

  auto [scene, camera] = setup_scene(config);

  std::cout << "Samples Per Pixel: " << config.samples_per_pixel << std::endl;
  std::cout << "Max Bounce Depth: " << config.max_bounce << std::endl;
  std::cout << "Camera Position: " << camera->position() << std::endl;
  std::cout << "Camera Direction: " << camera->direction() << std::endl;
  std::cout << "Camera Resolution: " << camera->resolution() << std::endl;
  std::cout << "Scene Size: " << scene->size() << std::endl;
  std::cout << "Scene Center: " << scene->center() << std::endl;
  std::cout << "Primitive Count: " << scene->primitive_count() << std::endl;

  Renderer renderer(camera.get(), scene.get(), config.max_bounce);
  if (const char* seed = std::getenv("PT_SEED")) renderer.set_seed((uint32_t)std::strtoul(seed, nullptr, 10));

  auto start = std::chrono::high_resolution_clock::now();
  int batch = config.batch_size;
  if (0 < batch) {
    while (renderer.total_samples < config.samples_per_pixel) {
      int todo = config.samples_per_pixel - renderer.total_samples;
      if (batch > todo) batch = todo;
      renderer.render(batch, config.print_progress);
      printf("%d/%d samples per pixel\n", renderer.total_samples, config.samples_per_pixel);
      auto path = result_path.parent_path() / std::filesystem::path("latest.png");
      renderer.save_image(path);
    }
  } else {
    renderer.render(config.samples_per_pixel, config.print_progress);
  }
  auto end = std::chrono::high_resolution_clock::now();
  std::chrono::duration<double, std::milli> duration = end - start;
  double seconds = duration.count() / 1000.0;
  int minutes = (int)(seconds / 60);
  seconds -= (minutes * 60);

  print_stats();
  printf("%d Samples/Pixel\n", renderer.total_samples);
  fprintf(stdout, "Render time: %dm%.3fs\n", minutes, seconds);

  renderer.save_image(result_path);
  return 0;
}
