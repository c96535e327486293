// Material, as src/material.h:11-36 (same fields, same defaults).  BSDF sampling and evaluation
// (the reference's BxDF class, src/material.cpp) run inside the GPU kernels and have no host-side twin.
#pragma once

#include <cstdint>

#include "image.h"
#include "vec.h"

struct Material {
  enum Type : uint8_t {
    DIFFUSE = 1 << 0,
    SPECULAR = 1 << 1,
    DIELECTRIC = 1 << 2,
    MIRROR = 1 << 3,
    MICROFACET = 1 << 4,
  };

  Type type = DIFFUSE;
  pt::dvec3 albedo = pt::dvec3(0.8);
  pt::dvec3 emission = pt::dvec3(0.0);
  double refraction_index = 1.52;
  double roughness = 0.5;
  double metallic = 0.5;  // 0.0 for dielectrics, 1.0 for metals
  Image* texture = nullptr;  // not owned (the reference leaks these: src/scene.cpp:157, src/main.cpp:136)

  inline bool is_type(Type other_type) const { return type & other_type; }

  inline bool is_perfectly_specular() const
  {
    constexpr double THRESHOLD = 1e-5;
    return is_type(Material::DIELECTRIC) || is_type(Material::MIRROR) ||
           ((is_type(Material::SPECULAR) || is_type(Material::MICROFACET)) && (roughness < THRESHOLD));
  }
};
