// Intersection, Sphere, Triangle, Primitive: the host-side value types of src/geometry.h:14-71.
// The intersection routines themselves (ray_vs_triangle, ray_vs_sphere, Primitive::intersect) run on
// the GPU; these types carry scene data to it and hit records back.
#pragma once

#include <cstdint>
#include <optional>
#include <vector>

#include "aabb.h"
#include "material.h"
#include "ray.h"
#include "vec.h"

struct Intersection {
  uint32_t id;
  double t;
  pt::dvec3 point;
  pt::dvec3 normal;
  pt::dvec2 uv;
  Material* material;
  bool inside;
  inline bool is_closer_than(const Intersection& other) const { return t < other.t; }
};

struct Sphere {
  pt::dvec3 center;
  double radius;
  Sphere() : center(0, 0, 0), radius(1) {}
  Sphere(const pt::dvec3& c, double r) : center(c), radius(r) {}
};

// vertex order is counter-clockwise
struct Triangle {
  pt::dvec3 v0, v1, v2;  // vertex position
  pt::dvec3 n0, n1, n2;  // normal
  pt::dvec2 t0, t1, t2;  // texture coordinate
  Triangle() {}
  // flat normal
  pt::dvec3 normal() const { return pt::normalize(pt::cross(v1 - v0, v2 - v0)); }
};

struct Primitive {
  enum Type : uint8_t { SPHERE, TRIANGLE };
  Type type;
  Sphere sphere;      // valid when type == SPHERE
  Triangle triangle;  // valid when type == TRIANGLE
  Material* material;
  AABB bbox;
  uint32_t id = 0;

  Primitive(const Sphere& s, Material* m) : type(SPHERE), sphere(s), material(m), bbox(AABB(s)) {}
  Primitive(const Triangle& t, Material* m) : type(TRIANGLE), triangle(t), material(m), bbox(AABB(t)) {}
  bool is_light() const { return material->emission.x > 0.0 || material->emission.y > 0.0 || material->emission.z > 0.0; }
  double sample_area() const;
};

AABB compute_bounding_volume(const std::vector<Primitive>::const_iterator& begin,
                             const std::vector<Primitive>::const_iterator& end);

void print_stats();
