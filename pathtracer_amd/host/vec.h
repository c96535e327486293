// Minimal double-precision vector types for the host API.
//
// The reference's public signatures use glm::dvec3 / dvec2 / ivec2 (e.g. src/camera.h:13-18,
// src/scene.h:26-29).  glm is not a dependency here; these types have the same member names and memory
// layout (x,y,z doubles), and a converting constructor that accepts any type with .x/.y/.z members, so a
// caller holding glm vectors can pass them unchanged.
#pragma once

#include <cmath>
#include <ostream>
#include <utility>

namespace pt {

struct dvec2 {
  double x = 0, y = 0;
  dvec2() = default;
  dvec2(double x_, double y_) : x(x_), y(y_) {}
  explicit dvec2(double s) : x(s), y(s) {}
  double& operator[](int i) { return i == 0 ? x : y; }
  const double& operator[](int i) const { return i == 0 ? x : y; }
};

struct ivec2 {
  int x = 0, y = 0;
};

struct dvec3 {
  double x = 0, y = 0, z = 0;
  dvec3() = default;
  dvec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
  explicit dvec3(double s) : x(s), y(s), z(s) {}
  template <class V, class = decltype(std::declval<V>().x + std::declval<V>().y + std::declval<V>().z)>
  dvec3(const V& v) : x(double(v.x)), y(double(v.y)), z(double(v.z)) {}
  double& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  const double& operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};

inline dvec3 operator+(const dvec3& a, const dvec3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline dvec3 operator-(const dvec3& a, const dvec3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline dvec3 operator*(const dvec3& a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline dvec3 operator*(double s, const dvec3& a) { return {a.x * s, a.y * s, a.z * s}; }
inline dvec3 operator/(const dvec3& a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline dvec3 operator-(const dvec3& a) { return {-a.x, -a.y, -a.z}; }
inline double dot(const dvec3& a, const dvec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline dvec3 cross(const dvec3& a, const dvec3& b)
{
  return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline double length(const dvec3& a) { return std::sqrt(dot(a, a)); }
inline dvec3 normalize(const dvec3& a) { return a * (1.0 / std::sqrt(dot(a, a))); }
inline double distance(const dvec3& a, const dvec3& b) { return length(a - b); }
inline dvec3 min(const dvec3& a, const dvec3& b) { return {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
inline dvec3 max(const dvec3& a, const dvec3& b) { return {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }
inline double radians(double degrees) { return degrees * 0.017453292519943295; }

inline std::ostream& operator<<(std::ostream& os, const dvec3& v) { return os << "[" << v.x << ", " << v.y << ", " << v.z << "]"; }
inline std::ostream& operator<<(std::ostream& os, const ivec2& v) { return os << "[" << v.x << ", " << v.y << "]"; }

}  // namespace pt
