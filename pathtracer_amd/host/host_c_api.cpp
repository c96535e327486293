// Plain-C handles on the host library for scripted checks (ctypes): image decode / encode and OBJ loading
// through exactly the code paths Scene and Image use.  Nothing here touches the GPU.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <fstream>
#include <iterator>
#include <string>

#include "image.h"
#include "json.h"
#include "scene.h"

extern "C" {

// Decode an image file as Image::load does (3 channels, rows bottom-up).  Returns a malloc'd buffer or NULL.
unsigned char* pt_host_image_load(const char* path, int* width, int* height)
{
  Image im;
  if (!im.load(path)) return nullptr;
  *width = im.width();
  *height = im.height();
  size_t bytes = (size_t)im.width() * (size_t)im.height() * 3;
  unsigned char* out = (unsigned char*)std::malloc(bytes);
  std::memcpy(out, im.data(), bytes);
  return out;
}

void pt_host_free(void* p) { std::free(p); }

// Write rows (top-down) as a PNG through Image::write.  Returns 1 on success.
int pt_host_image_write(const char* path, const unsigned char* rgb, int width, int height)
{
  Image im(width, height, 3);
  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) im.set_pixel(x, y, const_cast<unsigned char*>(rgb + ((size_t)y * width + x) * 3));
  return im.write(path) ? 1 : 0;
}

// Scene::load_obj + add_primitives on a fresh scene.  Fills up to `cap` triangles:
// pos/nrm n*9 doubles, uv n*6, per-triangle material type, albedo (n*3), emission (n*3), has_texture.
// Returns the triangle count (which may exceed cap).
int pt_host_load_obj(const char* path, int cap, double* pos, double* nrm, double* uv, int* mat_type, double* albedo,
                     double* emission, int* has_texture, int* n_lights)
{
  Scene scene;
  auto mesh = scene.load_obj(path);
  scene.add_primitives(mesh.begin(), mesh.end());
  int n = (int)mesh.size();
  for (int i = 0; i < n && i < cap; i++) {
    const Triangle& t = mesh[i].triangle;
    const pt::dvec3* P[3] = {&t.v0, &t.v1, &t.v2};
    const pt::dvec3* N[3] = {&t.n0, &t.n1, &t.n2};
    const pt::dvec2* T[3] = {&t.t0, &t.t1, &t.t2};
    for (int k = 0; k < 3; k++) {
      pos[9 * i + 3 * k] = P[k]->x; pos[9 * i + 3 * k + 1] = P[k]->y; pos[9 * i + 3 * k + 2] = P[k]->z;
      nrm[9 * i + 3 * k] = N[k]->x; nrm[9 * i + 3 * k + 1] = N[k]->y; nrm[9 * i + 3 * k + 2] = N[k]->z;
      uv[6 * i + 2 * k] = T[k]->x; uv[6 * i + 2 * k + 1] = T[k]->y;
    }
    const Material* m = mesh[i].material;
    mat_type[i] = (int)m->type;
    albedo[3 * i] = m->albedo.x; albedo[3 * i + 1] = m->albedo.y; albedo[3 * i + 2] = m->albedo.z;
    emission[3 * i] = m->emission.x; emission[3 * i + 1] = m->emission.y; emission[3 * i + 2] = m->emission.z;
    has_texture[i] = (m->texture && m->texture->valid()) ? 1 : 0;
  }
  if (n_lights) *n_lights = scene.light_count();
  return n;
}

// Parses a configuration file with the driver's JSON reader and flattens it to "path=value" lines (objects by
// key, arrays by index; numbers with 17 digits).  Returns the number of bytes needed (including the final 0), or -1
// when the file cannot be read or parsed.
int pt_host_json_flatten(const char* path, char* out, int cap)
{
  std::ifstream file(path);
  if (!file.is_open()) return -1;
  const std::string text((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
  std::string flat;
  try {
    const pt::Json doc = pt::JsonParser(text).parse();
    struct Walk {
      std::string& flat;
      void operator()(const pt::Json& v, const std::string& at)
      {
        char number[64];
        switch (v.kind) {
          case pt::Json::Object: for (const auto& kv : v.obj) (*this)(kv.second, at.empty() ? kv.first : at + "." + kv.first); break;
          case pt::Json::Array: for (size_t i = 0; i < v.arr.size(); i++) (*this)(v.arr[i], at + "[" + std::to_string(i) + "]"); break;
          case pt::Json::Number: std::snprintf(number, sizeof number, "%.17g", v.num); flat += at + "=" + number + "\n"; break;
          case pt::Json::String: flat += at + "=" + v.str + "\n"; break;
          case pt::Json::Bool: flat += at + "=" + (v.b ? "true" : "false") + "\n"; break;
          default: flat += at + "=null\n"; break;
        }
      }
    } walk{flat};
    walk(doc, "");
  } catch (const std::exception&) {
    return -1;
  }
  if (out && cap > 0) {
    const size_t n = std::min(flat.size(), (size_t)cap - 1);
    std::memcpy(out, flat.data(), n);
    out[n] = 0;
  }
  return (int)flat.size() + 1;
}

}  // extern "C"
