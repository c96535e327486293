#include "renderer.h"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <ctime>
#include <iomanip>
#include <iostream>

#include "../../include/ptk.h"
#include "config.h"
#include "image.h"

static void check(ptk_group* group, int rc, const char* what)
{
  if (rc != PTK_OK) {
    std::cerr << what << ": " << ptk_group_last_error(group) << std::endl;
    std::exit(1);
  }
}

Renderer::Renderer(Camera* camera, Scene* scene, int max_bounce) : m_scene(scene), m_camera(camera), m_max_bounce(max_bounce)
{
  ptk_group* ctx = m_scene->group();
  if (!ctx) {
    std::cerr << "Renderer: call Scene::compute_bvh() before constructing a Renderer" << std::endl;
    std::exit(1);
  }
  ptk_camera c{};
  c.width = m_camera->width();
  c.height = m_camera->height();
  pt::dvec3 p = m_camera->position(), f = m_camera->direction(), r = m_camera->right(), u = m_camera->up();
  c.position[0] = p.x; c.position[1] = p.y; c.position[2] = p.z;
  c.forward[0] = f.x; c.forward[1] = f.y; c.forward[2] = f.z;
  c.right[0] = r.x; c.right[1] = r.y; c.right[2] = r.z;
  c.up[0] = u.x; c.up[1] = u.y; c.up[2] = u.z;
  c.fov = m_camera->fov_radians();
  c.aperture = m_camera->aperture();
  c.focus_distance = m_camera->focus_distance();
  check(ctx, ptk_group_set_camera(ctx, &c), "ptk_group_set_camera");  // allocates and clears the W*H buffer (src/renderer.cpp:19-25)
  ptk_flags fl{PT_DEBUG_NORMAL, PT_RUSSIAN_ROULETTE, PT_DIRECT_LIGHT_SAMPLING, PT_INDIRECT_LIGHT_SAMPLING,
               PT_IMPORTANCE_SAMPLE, PT_CHECK_HEMISPHERE};
  check(ctx, ptk_group_set_flags(ctx, &fl), "ptk_group_set_flags");
}

// src/renderer.cpp:27-52
void Renderer::render(int samples, bool print_progress)
{
  ptk_group* ctx = m_scene->group();
  const int split = m_scene->split();
  ptk_render_params p{};
  p.samples = samples;
  p.max_bounce = m_max_bounce;
  p.seed = m_seed;
  p.sample_first = total_samples;
  p.sample_stride = 1;
  if (print_progress && (ptk_group_size(ctx) == 1 || split == PTK_SPLIT_SAMPLES)) {
    // the reference prints one line per image row; rows finish together here, so progress is per band
    const int band = 64;
    for (int y = 0; y < m_camera->height(); y += band) {
      printf("Progress: %.2f%%\n", (double(y) / double(m_camera->height())) * 100.0);
      p.row_begin = y;
      p.row_end = std::min(m_camera->height(), y + band);
      check(ctx, ptk_group_render(ctx, &p, PTK_SPLIT_SAMPLES), "ptk_group_render");
      check(ctx, ptk_group_synchronize(ctx), "ptk_group_synchronize");
    }
  } else {
    check(ctx, ptk_group_render(ctx, &p, split), "ptk_group_render");
    check(ctx, ptk_group_synchronize(ctx), "ptk_group_synchronize");
  }
  total_samples += samples;
}

std::vector<float> Renderer::read_buffer()
{
  ptk_group* ctx = m_scene->group();
  std::vector<float> rgb((size_t)m_camera->width() * (size_t)m_camera->height() * 3);
  check(ctx, ptk_group_read_accum(ctx, rgb.data(), nullptr), "ptk_group_read_accum");
  return rgb;
}

// src/renderer.cpp:180-201.  The per-pixel work (mean, ACES curve of src/renderer.cpp:174-178, gamma 1/2.2 of
// src/util.h:131-134, map_pixel of src/renderer.cpp:10) runs on the device; W*H*3 bytes come back.
void Renderer::save_image(const std::filesystem::path& path)
{
  ptk_group* ctx = m_scene->group();
  std::vector<uint8_t> rgb8((size_t)m_camera->width() * (size_t)m_camera->height() * 3);
  check(ctx, ptk_group_read_display_u8(ctx, rgb8.data(), nullptr), "ptk_group_read_display_u8");
  Image output(m_camera->width(), m_camera->height(), 3);
  for (int y = 0; y < m_camera->height(); y++)
    for (int x = 0; x < m_camera->width(); x++)
      output.set_pixel(x, y, &rgb8[((size_t)y * m_camera->width() + x) * 3]);
  output.write(path);

  auto now = std::chrono::system_clock::now();
  std::time_t current_time = std::chrono::system_clock::to_time_t(now);
  std::cout << "[" << std::put_time(std::localtime(&current_time), "%Y-%m-%d %H:%M:%S") << "] Save image to " << path
            << std::endl;
}

