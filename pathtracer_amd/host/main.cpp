// pt -- command-line front end of the GPU render path.
//
// The command line, the JSON keys with their defaults and the lines printed are the reference's
// (src/main.cpp:68-111 keys, :174-210 arguments, :227-266 output), so that scripts and configuration files
// written for it keep working:
//
//   pt <config.json> <output.png> [samples=8] [bounces=3] [batch_size=16] [--gpus N] [--split spp|tile] [--seed S]
//
// The scene is assembled straight from the JSON document into Scene / Camera calls; the three flags are
// additions (GPUs 0..N-1 of the node share every render call; see Scene::group()).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <iterator>
#include <memory>
#include <string>
#include <vector>

#include "geometry.h"
#include "image.h"
#include "json.h"
#include "material.h"
#include "renderer.h"
#include "scene.h"

namespace {

namespace fs = std::filesystem;
using pt::Json;

struct CommandLine {
  fs::path config, output;
  int samples = 8, bounces = 3, batch = 16;  // src/main.cpp:194-210
  int gpus = 0;                              // 0: leave PT_DEVICES / PT_DEVICE as they are
  std::string split;
  const char* seed = nullptr;
};

// Positional arguments in the reference's order, flags anywhere.  Returns false when the two required paths are missing.
bool read_command_line(int argc, char** argv, CommandLine& cl)
{
  std::vector<const char*> positional;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    const bool has_value = i + 1 < argc;
    if (a == "--gpus" && has_value) cl.gpus = std::atoi(argv[++i]);
    else if (a == "--split" && has_value) cl.split = argv[++i];
    else if (a == "--seed" && has_value) cl.seed = argv[++i];
    else positional.push_back(argv[i]);
  }
  if (positional.size() < 2) return false;
  cl.config = positional[0];
  cl.output = positional[1];
  int* numbers[] = {&cl.samples, &cl.bounces, &cl.batch};
  for (size_t k = 0; k < 3 && k + 2 < positional.size(); k++) *numbers[k] = std::atoi(positional[k + 2]);
  return true;
}

// Typed access to one JSON object with the reference's "key missing -> default" rule (src/main.cpp:50-66).
class Fields
{
 public:
  explicit Fields(const Json& object) : m_object(object) {}
  double number(const char* key, double otherwise) const { return has(key) ? m_object[key].number() : otherwise; }
  bool flag(const char* key, bool otherwise) const { return has(key) ? m_object[key].boolean() : otherwise; }
  std::string text(const char* key) const { return has(key) ? m_object[key].string() : std::string(); }
  pt::dvec3 vector(const char* key, pt::dvec3 otherwise) const
  {
    if (!has(key)) return otherwise;
    const Json& v = m_object[key];
    return pt::dvec3(v["x"].number(), v["y"].number(), v["z"].number());
  }
  const std::vector<Json>& list(const char* key) const
  {
    static const std::vector<Json> none;
    return has(key) ? m_object[key].arr : none;
  }

 private:
  bool has(const char* key) const { return m_object.contains(key); }
  const Json& m_object;
};

// "models": OBJ files, in order; their triangles take the first primitive ids (src/main.cpp:117-122)
void add_models(Scene& scene, const Fields& config)
{
  for (const Json& entry : config.list("models")) {
    std::vector<Primitive> mesh = scene.load_obj(entry.string());
    const AABB box = compute_bounding_volume(mesh.begin(), mesh.end());
    std::cout << "Mesh Size: " << box.size() << ", Mesh Center: " << box.center() << std::endl;
    scene.add_primitives(mesh.begin(), mesh.end());
  }
}

// "spheres": one material and one primitive each unless "hidden" (src/main.cpp:68-87 keys, :124-145 set-up)
void add_spheres(Scene& scene, const Fields& config)
{
  for (const Json& entry : config.list("spheres")) {
    const Fields s(entry);
    if (s.flag("hidden", false)) continue;
    Material* m = scene.add_material(Material());
    const std::string type = s.text("type");
    m->type = type == "SPECULAR" ? Material::SPECULAR : type == "TRANSMISSIVE" ? Material::DIELECTRIC : Material::DIFFUSE;
    m->albedo = s.vector("albedo", pt::dvec3(1.0));
    m->emission = s.vector("emissive", pt::dvec3(0.0));
    m->roughness = s.number("roughness", 0.5);
    m->metallic = s.number("metallic", 0.5);
    const std::string texture = s.text("texture");
    if (!texture.empty()) {
      Image* image = new Image();  // owned by nobody, like every material texture (Material::texture is a plain pointer)
      const bool ok = image->load(texture);
      (ok ? std::cout << "Loaded texture " : std::cerr << "Failed to load texture ") << texture << std::endl;
      m->texture = image;
    }
    scene.add_primitive(Primitive(Sphere(s.vector("center", pt::dvec3(0.0)), s.number("radius", 1.0)), m));
  }
}

// "background_texture" wins over "background_color" (src/main.cpp:147-158); an unreadable map ends the run
void set_background(Scene& scene, const Fields& config)
{
  const std::string map = config.text("background_texture");
  if (map.empty()) {
    scene.set_background_color(config.vector("background_color", pt::dvec3(-1.0)));
    return;
  }
  auto image = std::make_unique<Image>();
  if (!image->load(map)) {
    std::cerr << "Failed to load environment texture " << map << std::endl;
    std::exit(1);
  }
  std::cout << "Loaded environment texture " << map << std::endl;
  scene.set_background_texture(std::move(image));
}

std::unique_ptr<Camera> make_camera(const Fields& config)
{
  auto camera = std::make_unique<Camera>((int)config.number("image_width", 640), (int)config.number("image_height", 480),
                                         config.number("camera_fov", 45), config.number("camera_aperture", 0),
                                         config.number("camera_focus_distance", 0));
  const pt::dvec3 position = config.vector("camera_position", pt::dvec3(0.0, 0.0, 10.0));
  const pt::dvec3 target = config.vector("camera_target", pt::dvec3(0.0, 0.0, 0.0));
  camera->look_at(position, target);
  std::cout << "Distance camera position to camera target: " << pt::distance(position, target) << std::endl;
  return camera;
}

// Renders in batches with a preview file after each (src/main.cpp:242-255); returns the seconds it took.
double render_all(Renderer& renderer, const CommandLine& cl, bool progress)
{
  const auto begin = std::chrono::steady_clock::now();
  if (cl.batch <= 0) {
    renderer.render(cl.samples, progress);
  } else {
    const fs::path preview = cl.output.parent_path() / "latest.png";
    while (renderer.total_samples < cl.samples) {
      renderer.render(std::min(cl.batch, cl.samples - renderer.total_samples), progress);
      std::printf("%d/%d samples per pixel\n", renderer.total_samples, cl.samples);
      renderer.save_image(preview);
    }
  }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count();
}

}  // namespace

int main(int argc, char** argv)
{
  CommandLine cl;
  if (!read_command_line(argc, argv, cl)) {
    std::cerr << "Usage: " << argv[0] << " <path to config.json> <output path> <samples> <bounces> <batch_size>" << std::endl;
    return 1;
  }
  std::ifstream file(cl.config);
  if (!file.is_open()) {
    std::cerr << "Could not open config: " << cl.config.string() << std::endl;
    return 1;
  }
  const std::string text((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
  Json document;
  try {
    document = pt::JsonParser(text).parse();
  } catch (const std::exception& e) {
    std::cerr << "Could not parse config: " << e.what() << std::endl;
    return 1;
  }
  if (cl.gpus > 0) {
    std::string list;
    for (int d = 0; d < cl.gpus; d++) list += (d ? "," : "") + std::to_string(d);
    setenv("PT_DEVICES", list.c_str(), 1);
  }
  if (!cl.split.empty()) setenv("PT_SPLIT", cl.split.c_str(), 1);
  if (!cl.seed) cl.seed = std::getenv("PT_SEED");

  Scene scene;
  std::unique_ptr<Camera> camera;
  bool progress = false;
  try {
    const Fields config(document);
    progress = config.flag("print_progress", false);
    add_models(scene, config);
    add_spheres(scene, config);
    set_background(scene, config);
    camera = make_camera(config);
  } catch (const std::exception& e) {
    std::cerr << "Could not parse config: " << e.what() << std::endl;
    return 1;
  }
  scene.compute_bvh();

  std::cout << "Samples Per Pixel: " << cl.samples << std::endl;
  std::cout << "Max Bounce Depth: " << cl.bounces << std::endl;
  std::cout << "Camera Position: " << camera->position() << std::endl;
  std::cout << "Camera Direction: " << camera->direction() << std::endl;
  std::cout << "Camera Resolution: " << camera->resolution() << std::endl;
  std::cout << "Scene Size: " << scene.size() << std::endl;
  std::cout << "Scene Center: " << scene.center() << std::endl;
  std::cout << "Primitive Count: " << scene.primitive_count() << std::endl;

  Renderer renderer(camera.get(), &scene, cl.bounces);
  if (cl.seed) renderer.set_seed((uint32_t)std::strtoul(cl.seed, nullptr, 10));
  const double seconds = render_all(renderer, cl, progress);

  print_stats();
  std::printf("%d Samples/Pixel\n", renderer.total_samples);
  std::printf("Render time: %dm%.3fs\n", (int)(seconds / 60.0), seconds - 60.0 * (int)(seconds / 60.0));
  renderer.save_image(cl.output);
  return 0;
}
