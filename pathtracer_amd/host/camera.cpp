// Camera state (src/camera.cpp:6-17, 62-89).
#include "camera.h"

Camera::Camera(int width, int height, double fov, double aperture, double focus_distance)
    : m_width(width),
      m_height(height),
      m_world_up(0.0, 1.0, 0.0),
      m_position(0.0, 0.0, 0.0),
      m_forward(0.0, 0.0, -1.0),
      m_fov(pt::radians(fov)),
      m_focus_distance(focus_distance),
      m_aperture(aperture)
{
  compute();
}

void Camera::set_position(const pt::dvec3& position)
{
  m_position = position;
  compute();
}

void Camera::set_forward(const pt::dvec3& forward)
{
  m_forward = forward;
  compute();
}

// right and up are deliberately not normalised (src/camera.cpp:78-82)
void Camera::compute()
{
  m_right = pt::cross(m_forward, m_world_up);
  m_up = pt::cross(m_right, m_forward);
}

void Camera::look_at(const pt::dvec3& position, const pt::dvec3& target)
{
  m_position = position;
  m_forward = pt::normalize(target - position);
  compute();
}
