// Camera: the public surface of src/camera.h:6-29.  Ray generation (get_ray, src/camera.cpp:19-60:
// sub-pixel jitter, thin-lens depth of field) runs on the GPU from the state kept here.
#pragma once

#include "ray.h"
#include "vec.h"

class Camera
{
 public:
  // fov in degrees; aperture is the defocus angle in degrees; focus_distance 0 disables depth of field
  Camera(int width, int height, double fov, double aperture, double focus_distance);
  int width() const { return m_width; }
  int height() const { return m_height; }
  void set_position(const pt::dvec3& position);
  void set_forward(const pt::dvec3& forward);
  void look_at(const pt::dvec3& position, const pt::dvec3& target);
  pt::dvec3 position() const { return m_position; }
  pt::dvec3 direction() const { return m_forward; }
  pt::ivec2 resolution() const { return {m_width, m_height}; }
  // state the kernels need (src/camera.h:21-26)
  pt::dvec3 right() const { return m_right; }
  pt::dvec3 up() const { return m_up; }
  double fov_radians() const { return m_fov; }
  double aperture() const { return m_aperture; }
  double focus_distance() const { return m_focus_distance; }

 private:
  const int m_width, m_height;
  const pt::dvec3 m_world_up;
  pt::dvec3 m_position, m_forward, m_right, m_up;
  double m_fov;
  double m_focus_distance;
  double m_aperture;

  void compute();
};
