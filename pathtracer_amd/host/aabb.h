// AABB, as src/aabb.h:12-41 (the slab test itself runs on the GPU).
#pragma once

#include <cstddef>

#include "vec.h"

struct Sphere;
struct Triangle;

struct AABB {
  pt::dvec3 min;
  pt::dvec3 max;

  AABB() {}
  AABB(const pt::dvec3& a, const pt::dvec3& b) : min(pt::min(a, b)), max(pt::max(a, b)) {}
  AABB(const Sphere&);
  AABB(const Triangle&);

  inline pt::dvec3 size() const { return max - min; }
  inline pt::dvec3 center() const { return min + size() / 2.0; }

  inline size_t longest_axis() const
  {
    auto s = size();
    if (s.y >= s.x && s.y >= s.z) return 1;
    if (s.z >= s.x && s.z >= s.y) return 2;
    return 0;
  }
};

AABB merge(const AABB& a, const AABB& b);
