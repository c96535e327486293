// Ray, as src/ray.h:5-9.
#pragma once

#include "vec.h"

struct Ray {
  pt::dvec3 origin;
  pt::dvec3 direction;
  inline pt::dvec3 point_at(double t) const { return origin + direction * t; }
};
