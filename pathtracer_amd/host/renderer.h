// Renderer: the public surface of src/renderer.h:15-32.
//
// render(samples) launches the path kernel for `samples` more samples of every pixel and adds them to the
// accumulation buffer on the GPU (the reference's running mean, src/renderer.cpp:39-47, kept as sum + count);
// save_image() reads the buffer back and applies the reference's display transform on the host
// (ACES, gamma, 8-bit truncation, PNG: src/renderer.cpp:174-201).
#pragma once

#include <cstdint>
#include <filesystem>
#include <vector>

#include "camera.h"
#include "scene.h"

class Renderer
{
 public:
  Renderer(Camera* camera, Scene* scene, int max_bounce);
  void render(int samples, bool print_progress = false);
  void save_image(const std::filesystem::path& path);

  int total_samples = 0;

  // Not in the reference (its generator is seeded from std::random_device): the seed of the counter-based
  // random numbers.  Two renders with the same seed are identical.
  void set_seed(uint32_t seed) { m_seed = seed; }
  // Mean radiance per pixel, W*H*3 floats, row 0 = top (what save_image tone-maps).
  std::vector<float> read_buffer();

 private:
  Scene* m_scene;
  Camera* m_camera;
  int m_max_bounce;
  uint32_t m_seed = 1;
};
