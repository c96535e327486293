// Per-path work of the reference's renderer, restated for one GPU lane per
// path: camera ray (src/camera.cpp:19-60), the bounce loop of
// Renderer::trace_ray with one-light next-event estimation
// (src/renderer.cpp:57-171, recursion unrolled into throughput form), BSDF
// sampling / evaluation (src/material.cpp:110-242), texture and environment
// lookups (src/image.cpp:40-85, src/scene.cpp:21-45).
//
// Random numbers: the reference draws from one process-global std::mt19937
// (src/util.h:72-78); here every path owns a counter-based stream keyed by
// (seed, pixel, global sample index), so a frame can be split over launches
// and GPUs by sample index or by rows and still be the same set of paths.
#pragma once

#include "ptk_traverse.h"

namespace ptk {

constexpr float kPi = 3.14159265359f;  // src/util.h:12

// ---------------------------------------------------------------- random numbers
// Draw i of a stream is mix32((k0 + i * 0x9E3779B9) ^ k1); `c` carries k0 + i * 0x9E3779B9 from draw to
// draw, so that a draw costs one addition instead of a (quarter-rate) integer multiply.
struct Rng {
  uint32_t c, k1;
};

PTK_HD uint32_t mix32(uint32_t x)
{
  x ^= x >> 16; x *= 0x21f0aaadu;
  x ^= x >> 15; x *= 0x735a2d97u;
  x ^= x >> 15;
  return x;
}

PTK_HD Rng rng_init(uint32_t seed, uint32_t pixel, uint32_t sample)
{
  Rng r;
  const uint32_t k0 = mix32(pixel + mix32(seed + 0x9E3779B9u));
  r.k1 = mix32(sample + 0x85EBCA6Bu) ^ ((k0 << 13) | (k0 >> 19));
  r.c = k0;
  return r;
}

// uniform in [0,1), 24 bits
PTK_HD float rnd(Rng& r)
{
#ifdef PTK_WHATIF_CHEAP_RNG  // (what-if build for weighing the generator: one multiply instead of two; other numbers, similar paths)
  uint32_t x = (r.c ^ r.k1) * 0x21f0aaadu;
  x ^= x >> 15;
#else
  uint32_t x = mix32(r.c ^ r.k1);
#endif
  r.c += 0x9E3779B9u;
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}

// ---------------------------------------------------------------- small helpers
template <class P>
PTK_HD v3 ld3(P p) { return V3(p[0], p[1], p[2]); }  // P: pointer to three floats, in global memory or LDS

// util.h:83-89 (y is the polar axis)
PTK_HD v3 spherical_dir(float cos_theta, float sin_theta, float phi_turns)
{
  float s, c;
  sincos_turns(phi_turns, &s, &c);
  return V3(c * sin_theta, cos_theta, s * sin_theta);
}

// random_unit_vector, util.h:91-96: theta = acos(1 - 2 r1)
PTK_HD v3 unit_vector_from(float r0, float r1)
{
  float ct = 1.0f - 2.0f * r1;
  float st = sqrtf(fmaxf(0.0f, 1.0f - ct * ct));
  return spherical_dir(ct, st, r0);
}

PTK_HD v3 reflect(v3 I, v3 N) { return I - N * (dot(N, I) * 2.0f); }
PTK_HD v3 refract(v3 I, v3 N, float eta)
{
  float dn = dot(N, I);
  float k = 1.0f - eta * eta * (1.0f - dn * dn);
  if (k >= 0.0f) return I * eta - N * (eta * dn + sqrtf(k));
  return V3(0.0f, 0.0f, 0.0f);
}

PTK_HD v3 pow3(v3 c, float e) { return V3(fast_pow(c.x, e), fast_pow(c.y, e), fast_pow(c.z, e)); }

// Orthonormal frame with local +y along the normal (local_to_world, src/util.h:63-70).
// The reference builds it in float and inverts it numerically; the inverse of an
// orthonormal matrix is its transpose, which is what to_local uses.
struct Frame {
  v3 r, u, f;
};
PTK_HD Frame make_frame(v3 up)
{
  Frame fr;
  fr.u = normalize(up);
  v3 ref = (fabsf(fr.u.z) < 0.9f) ? V3(0.0f, 0.0f, 1.0f) : V3(1.0f, 0.0f, 0.0f);
  fr.r = normalize(cross(ref, fr.u));
  fr.f = cross(fr.u, fr.r);
  return fr;
}
PTK_HD v3 to_local(const Frame& fr, v3 w) { return V3(dot(fr.r, w), dot(fr.u, w), dot(fr.f, w)); }
PTK_HD v3 to_world(const Frame& fr, v3 l) { return fr.r * l.x + fr.u * l.y + fr.f * l.z; }

// ---------------------------------------------------------------- textures
struct Texel {
  int r, g, b;
};

// glm::mix on u8vec3 converts the blended value back to uint8 (truncation) at every lerp, src/image.cpp:78;
// out-of-range blends (negative fractions from texture coordinates below 0) keep their low byte
PTK_HD Texel mix_u8(Texel a, Texel b, float t)
{
  // One rounded product and one fused multiply-add, spelled out: the truncation below turns a last-bit difference
  // into a whole step of the texel's value, so how the sum is contracted must not be left to each instantiation of
  // the kernel template.
  Texel r;
  const float s = 1.0f - t;
  r.r = ((int)fmaf((float)b.r, t, (float)a.r * s)) & 255;
  r.g = ((int)fmaf((float)b.g, t, (float)a.g * s)) & 255;
  r.b = ((int)fmaf((float)b.b, t, (float)a.b * s)) & 255;
  return r;
}

// Image::sample, src/image.cpp:62-85, with Image::pixel, src/image.cpp:40-48 (a tap outside the image gives
// INVALID_COLOR, which as a u8vec3 is (1,0,1)).  Straight-line: the two taps of a row are six consecutive bytes,
// fetched as ONE 8-byte word from a clamped position (the texel pool ends in padding, global memory takes
// unaligned loads, and byte offsets into the pool fit 32 bits: build_host_scene checks), and which taps lie outside
// the image is settled by selects afterwards.  Both fetches are in flight together; taken one tap at a time behind
// its own bounds check, the four fetches of a sample each waited for the one before.
struct TextureView {
  uint32_t offset;  // of the first texel, in bytes from the start of the texel pool
  int32_t width, height;
};
PTK_HD TextureView texture_view(const SceneView& sc, int tex)
{
  const TextureRec r = sc.textures[tex];
  return TextureView{(uint32_t)r.offset, r.width, r.height};
}
PTK_HD Texel sample_texture(const SceneView& sc, const TextureView& t, float u, float v)
{
  float x = u * (float)(t.width - 1);
  float y = v * (float)(t.height - 1);
  // int(x) truncates toward zero; NaN and out-of-range coordinates must land outside the image
  int x0 = (x > -2.0e9f && x < 2.0e9f) ? (int)x : -0x40000000;
  int y0 = (y > -2.0e9f && y < 2.0e9f) ? (int)y : -0x40000000;
  float fx = x - (float)x0, fy = y - (float)y0;
  const bool x_in0 = (uint32_t)x0 < (uint32_t)t.width, x_in1 = (uint32_t)(x0 + 1) < (uint32_t)t.width;
  const bool y_in0 = (uint32_t)y0 < (uint32_t)t.height, y_in1 = (uint32_t)(y0 + 1) < (uint32_t)t.height;
  const bool at_left_edge = x0 < 0;  // x0 == -1: the right tap is the texel the clamped fetch starts at
  auto clampi = [](int i, int last) { return i < 0 ? 0 : (i > last ? last : i); };
  const int xb = clampi(x0, t.width - 1), yb0 = clampi(y0, t.height - 1), yb1 = clampi(y0 + 1, t.height - 1);
  const uint32_t base = t.offset + (uint32_t)xb * 3u;
  const uint32_t row_bytes = (uint32_t)t.width * 3u;
  const uint8_t* q0 = sc.texels + (base + (uint32_t)yb0 * row_bytes);
  const uint8_t* q1 = sc.texels + (base + (uint32_t)yb1 * row_bytes);
  Texel a0, a1, b0, b1;  // row y0: texel at xb and its right neighbour; row y0 + 1 likewise
#if defined(__HIP_DEVICE_COMPILE__)
  struct __attribute__((packed)) Pair { uint32_t lo, hi; };
  const Pair w0 = *reinterpret_cast<const Pair*>(q0), w1 = *reinterpret_cast<const Pair*>(q1);
  a0.r = (int)(w0.lo & 0xffu); a0.g = (int)((w0.lo >> 8) & 0xffu); a0.b = (int)((w0.lo >> 16) & 0xffu);
  a1.r = (int)(w0.lo >> 24); a1.g = (int)(w0.hi & 0xffu); a1.b = (int)((w0.hi >> 8) & 0xffu);
  b0.r = (int)(w1.lo & 0xffu); b0.g = (int)((w1.lo >> 8) & 0xffu); b0.b = (int)((w1.lo >> 16) & 0xffu);
  b1.r = (int)(w1.lo >> 24); b1.g = (int)(w1.hi & 0xffu); b1.b = (int)((w1.hi >> 8) & 0xffu);
#else
  a0.r = q0[0]; a0.g = q0[1]; a0.b = q0[2]; a1.r = q0[3]; a1.g = q0[4]; a1.b = q0[5];
  b0.r = q1[0]; b0.g = q1[1]; b0.b = q1[2]; b1.r = q1[3]; b1.g = q1[4]; b1.b = q1[5];
#endif
  const Texel outside = {1, 0, 1};
  auto pick = [&](bool inside, const Texel& p) { Texel r = outside; if (inside) r = p; return r; };
  const Texel tl = pick(x_in0 && y_in0, a0), tr = pick(x_in1 && y_in0, at_left_edge ? a0 : a1);
  const Texel bl = pick(x_in0 && y_in1, b0), br = pick(x_in1 && y_in1, at_left_edge ? b0 : b1);
  return mix_u8(mix_u8(tl, tr, fx), mix_u8(bl, br, fx), fy);
}

// pow(texel / 255, 2.2), which is what both users of a texture sample compute from it (src/geometry.cpp:18-26,
// src/scene.cpp:33-45): a table of the 256 possible values (gamma_lut in ptk_layout.h) instead of a logarithm and
// an exponential per channel.
PTK_HD v3 texel_to_linear(const SceneView& sc, Texel p)
{
  const float* lut = gamma_lut(sc);
  return V3(lut[p.r], lut[p.g], lut[p.b]);
}

// Scene::sample_background, src/scene.cpp:21-45; equirectangular, src/util.h:142-149
PTK_HD v3 background(const SceneView& sc, v3 dir)
{
  if (sc.background_texture >= 0) {
    float u = atan2f(dir.z, dir.x) * 0.1591f + 0.5f;
    float v = asinf(dir.y) * 0.3183f + 0.5f;
    return texel_to_linear(sc, sample_texture(sc, texture_view(sc, sc.background_texture), u, v));
  }
  if (sc.background_is_color) return ld3(sc.background_color);
  float a = 0.5f * (dir.y + 1.0f);
  return V3(1.0f, 1.0f, 1.0f) * (1.0f - a) + V3(0.5f, 0.7f, 1.0f) * a;
}

// Intersection::albedo, src/geometry.cpp:18-26
template <class M>
PTK_HD v3 surface_albedo(const SceneView& sc, const M& m, const Surface& s)
{
#ifndef PTK_WHATIF_NO_TEXTURE  // (what-if build for weighing the texture path: wrong colours, same paths)
  if (m.texture >= 0) {  // the material record carries its texture's record in place of a colour (MaterialRec::albedo)
    const TextureView t = {as_uint(m.albedo[0]), (int32_t)as_uint(m.albedo[1]), (int32_t)as_uint(m.albedo[2])};
    return texel_to_linear(sc, sample_texture(sc, t, s.uv[0], s.uv[1]));
  }
#endif
  return ld3(m.albedo);
}

// ---------------------------------------------------------------- camera
// Camera::get_ray, src/camera.cpp:19-60.  The image plane spans +-2*tan(fov/2) and the
// jitter is centred on the integer pixel corner, exactly as there.
PTK_HD void camera_ray(const CameraView& cam, int x, int y, Rng& g, v3& origin, v3& direction)
{
  float jx = rnd(g) - 0.5f, jy = rnd(g) - 0.5f;
  float uvx = (((float)x + jx) / (float)cam.width) * 2.0f - 1.0f;
  float uvy = (((float)y + jy) / (float)cam.height) * 2.0f - 1.0f;
  v3 right = ld3(cam.right), up = ld3(cam.up);
  v3 dir = normalize(ld3(cam.forward) + right * (cam.plane_w * uvx) - up * (cam.plane_h * uvy));
  origin = ld3(cam.position);
  direction = dir;
  if (cam.defocus_radius > 0.0f) {
    float dx, dy;
    for (;;) {  // random_in_unit_disk, src/util.h:98-106
      dx = 2.0f * rnd(g) - 1.0f;
      dy = 2.0f * rnd(g) - 1.0f;
      if (dx * dx + dy * dy < 1.0f) break;
    }
    v3 offset = right * (dx * cam.defocus_radius) + up * (dy * cam.defocus_radius);
    origin = origin + offset;
    direction = normalize(dir * cam.focus_distance - offset);  // focus_point - origin
  }
}

// ---------------------------------------------------------------- BSDF
// material.cpp:100-106
PTK_HD float schlick_reflectance(float cosine, float ri)
{
  float r0 = (1.0f - ri) / (1.0f + ri);
  r0 = r0 * r0;
  float m = 1.0f - cosine;
  float m2 = m * m;
  return r0 + (1.0f - r0) * (m2 * m2 * m);
}

// BxDF::sample, src/material.cpp:110-122.  Draw order: dielectric takes one uniform (the
// reflect/refract choice), every other type two (direction).
// The outgoing direction in the shading frame, worked out where a model reads it: the diffuse model, most of what a
// wave shades, never does, and computed ahead of the branches it would be nine instructions for every shade step.
struct LocalOut {
  const Frame& fr;
  v3 d;  // direction of the segment that arrived (the outgoing direction is -d)
  PTK_HD v3 get() const { return to_local(fr, -d); }
};

template <class M>
PTK_HD v3 bsdf_sample(const FlagView& fl, const M& m, bool inside, const LocalOut& out, Rng& g)
{
  if (m.type == 4u /* DIELECTRIC, :225-240 */) {
    float r = rnd(g);
    const v3 wo = out.get();
    float ri = inside ? (1.0f / m.ior) : m.ior;
    v3 N = V3(0.0f, 1.0f, 0.0f);
    float cos_theta = fminf(wo.y, 1.0f);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    if ((ri * sin_theta > 1.0f) || (schlick_reflectance(cos_theta, ri) > r)) return reflect(-wo, N);
    return refract(-wo, N, ri);
  }
  float r0 = rnd(g), r1 = rnd(g);
  if (m.type == 2u /* SPECULAR, :157-162 */)
    return reflect(-out.get(), V3(0.0f, 1.0f, 0.0f)) + unit_vector_from(r0, r1) * m.roughness;
  if (m.type == 16u /* MICROFACET */ && fl.importance_sample) {  // Sample_Beckmann, :57-67
    float alpha = sqf(m.roughness), alpha2 = sqf(alpha);
    float tan2 = -alpha2 * logf(1.0f - (r1 / kPi));  // tan^2(theta)
    float ct = 1.0f / sqrtf(1.0f + tan2);
    float st = sqrtf(fmaxf(0.0f, 1.0f - ct * ct));
    v3 H = spherical_dir(ct, st, r0);
    return reflect(-out.get(), H);
  }
  // diffuse, :138-144: theta = acos(sqrt(r1))
  float ct = sqrtf(r1);
  float st = sqrtf(fmaxf(0.0f, 1.0f - r1));
  return spherical_dir(ct, st, r0);
}

// BxDF::eval, src/material.cpp:124-136
template <class M>
PTK_HD v3 bsdf_eval(const FlagView& fl, const M& m, v3 albedo, v3 V, v3 L)
{
  if (m.type == 4u) return V3(1.0f, 1.0f, 1.0f);  // :242
  if (m.type == 16u) {                            // :175-215
    if (fl.check_hemisphere && (L.y < 0.0f || V.y < 0.0f)) return V3(0.0f, 0.0f, 0.0f);
    float alpha = sqf(m.roughness), alpha2 = sqf(alpha);
    v3 H = normalize(V + L);
    float NoV = V.y, NoL = L.y, NoH = H.y;
    float VoH = fabsf(dot(V, H));
    v3 f0 = V3(0.04f, 0.04f, 0.04f) * (1.0f - m.metallic) + albedo * m.metallic;
    float p1 = 1.0f - VoH, p2 = p1 * p1;
    float p5 = p2 * p2 * p1;
    v3 F = f0 + (V3(1.0f, 1.0f, 1.0f) - f0) * p5;
    float k = alpha / 2.0f;  // G1_GGX_Schlick, :81-86
    float G = (fmaxf(NoV, 0.001f) / (NoV * (1.0f - k) + k)) * (fmaxf(NoL, 0.001f) / (NoL * (1.0f - k) + k));
    float NoH2 = sqf(NoH);
    float x = ((1.0f - NoH2) / NoH2) / alpha2;  // tan^2(theta_h) / alpha^2
    // D_Beckmann (:44-55) = exp(-x) / (alpha2 NoH^4); PDF_Beckmann (:69-79) = exp(-x) / (alpha2 NoH^3 pi).
    // With importance sampling the reference divides one by the other; exp(-x) cancels to pi / NoH.  In
    // fp32 exp(-x) underflows long before it does in the reference's doubles (x > 87 vs x > 745), so the
    // quotient is formed analytically and 0/0 = NaN is kept only where doubles would underflow too.
    float d_over_pdf;
    if (fl.importance_sample) d_over_pdf = (x > 745.0f) ? (0.0f / 0.0f) : kPi / NoH;
    else d_over_pdf = (expf(-x) / (alpha2 * sqf(NoH2))) / (NoL / kPi);
    v3 specular = F * (G / (4.0f * NoV * NoL));
    return specular * (NoL * d_over_pdf);
  }
  return albedo;  // diffuse :153, specular :164
}

PTK_HD float luma(v3 c) { return dot(c, V3(0.2126f, 0.7152f, 0.0722f)); }  // src/renderer.cpp:55

// ---------------------------------------------------------------- the path, as a per-lane state machine
// Renderer::trace_ray (src/renderer.cpp:57-126) is a recursion whose every level makes up to two
// closest-hit queries: the path segment itself and one shadow ray (Renderer::sample_lights,
// src/renderer.cpp:130-171).  Here a lane carries one path as explicit state, so that ALL its
// queries -- whatever the bounce, segment or shadow -- run through one traversal loop shared by
// the wave.  path_consume() is called whenever the lane's query has finished; it uses the result
// and sets up the lane's next query:
//
//   segment finished, miss      -> add background, path ends
//   segment finished, hit       -> Russian roulette, BSDF sample, emission, pick a light;
//                                  next query = shadow ray (if a light was sampled) else next segment
//   shadow ray finished         -> add the light's contribution if the light is what was hit;
//                                  next query = next segment
//   path ends                   -> its radiance goes to the pixel's accumulator; the lane asks for work
//
// Work is handed out as items: a tile of 1 << slot_bits pixel slots times `samples` samples,
// item k = sample (k >> slot_bits) of slot (k & mask).  A lane is not tied to a pixel: whichever
// lane is free takes the next item, so the lanes of a wave stay busy until the whole tile is done
// instead of each draining its own pixel's samples.
//
// The arithmetic in recursion order: with beta the product of rr_weight * BSDF value along the
// path, a level contributes beta * rr_weight * (emission + direct) and a Russian-roulette kill
// contributes beta * emission, exactly the values the reference's recursion returns.
struct PathState {
  Traversal tr;        // the query in flight
  Rng g;
  v3 beta, L;
  v3 pending;          // shadow ray in flight: radiance to add if it reaches the light
  v3 next_d;           // shadow ray in flight: direction of the path segment that follows it
  uint32_t bits;       // pixel slot of the tile that the path belongs to (bits 0-7) and depth (bits 8 up), one register
  bool shadow;         // the query in flight is a shadow ray
  bool prev_specular;
  bool alive;          // false once the tile has no work items left for this lane
  PTK_HD uint32_t slot() const { return bits & 0xffu; }
  PTK_HD int32_t depth() const { return (int32_t)(bits >> 8); }
};

struct PathParams {
  int32_t x0, y0;                // pixel of slot 0
  int32_t slot_bits, row_bits;   // 1 << slot_bits slots, 1 << row_bits of them to a tile row
  int32_t x_end, y_end;          // slots at or beyond these pixel coordinates are outside the image
  int32_t samples, max_bounce, sample_first, sample_stride;
  uint32_t seed;
  float camera_shift;            // camera_entry_shift() of the launch
};

// How far every camera ray of a launch can be advanced before it can meet the scene's bounds: the camera's
// distance from them (less the lens radius, where ray origins are spread) when that is more than four times
// the bounds' diagonal, otherwise 0.  Same reason as scene_entry_shift: box tests lose precision with the
// origin's distance from the scene.  One number per launch, so it costs the kernel no registers.
PTK_HD float camera_entry_shift(const SceneView& sc, const CameraView& cam)
{
  float dist2 = 0.0f, diag2 = 0.0f;
  for (int a = 0; a < 3; a++) {
    const float lo = sc.grid.origin[a], hi = sc.grid.origin[a] + 65535.0f * sc.grid.scale[a];
    const float p = cam.position[a];
    const float out = p < lo ? lo - p : (p > hi ? p - hi : 0.0f);
    dist2 += out * out;
    diag2 += (hi - lo) * (hi - lo);
  }
  const float lens = cam.defocus_radius * sqrtf(fmaxf(dot(ld3(cam.right), ld3(cam.right)), dot(ld3(cam.up), ld3(cam.up))));
  const float dist = sqrtf(dist2) - lens, diag = sqrtf(diag2);
  if (!(dist > 4.0f * diag)) return 0.0f;
  return dist - fmaxf(0.5f * diag, dist * (1.0f / 1024.0f));
}

// Where a path keeps the two vectors that only wait while its shadow ray is traced (the light's contribution and the
// direction of the segment that follows): in the path state itself, or wherever the caller's policy puts them (the
// render kernel parks them in LDS, which takes six registers out of the traversal loops).
struct ParkInState {
  PTK_HD void store(PathState& ps, v3 pending, v3 next_d) const { ps.pending = pending; ps.next_d = next_d; }
  PTK_HD void load(const PathState& ps, v3& pending, v3& next_d) const { pending = ps.pending; next_d = ps.next_d; }
};

struct SumAccum {  // accumulator of a one-pixel tile
  v3 sum;
  PTK_HD void add(uint32_t, v3 L) { sum += L; }
};

PTK_HD void path_init(PathState& ps)
{
  ps.L = ps.beta = ps.pending = ps.next_d = V3(0.0f, 0.0f, 0.0f);
  ps.bits = 0;
  ps.shadow = ps.prev_specular = false;
  ps.alive = true;
  ps.tr.node = kTravDone;
}

// Give a free lane work item `item`.  Returns true when the lane is still free afterwards (the item
// lay outside the image, or needed no query) and wants another item; false when the lane has a
// query in flight or the items have run out (then ps.alive is false).
template <class Stack, bool DEFER = false>
PTK_HD bool path_begin_item(PathState& ps, Stack& st, const SceneView& sc, const CameraView& cam,
                            const PathParams& pp, uint32_t item, Counters& cnt)
{
  if (item >= ((uint32_t)pp.samples << pp.slot_bits)) {
    ps.alive = false;
    ps.tr.node = kTravDone;
    return false;
  }
  uint32_t slot = item & ((1u << pp.slot_bits) - 1u);
  int px = pp.x0 + (int)(slot & ((1u << pp.row_bits) - 1u)), py = pp.y0 + (int)(slot >> pp.row_bits);
  if (px >= pp.x_end || py >= pp.y_end) return true;
  uint32_t index = (uint32_t)(pp.sample_first + (int32_t)(item >> pp.slot_bits) * pp.sample_stride);
  ps.g = rng_init(pp.seed, (uint32_t)(py * cam.width + px), index);
  v3 o, d;
  camera_ray(cam, px, py, ps.g, o, d);
  cnt.camera_rays++;
  if (pp.max_bounce <= 0) return true;  // every sample is black without a single query (src/renderer.cpp:59-61)
  ps.bits = slot;  // depth 0
  ps.prev_specular = false;
  ps.shadow = false;
  ps.L = V3(0.0f, 0.0f, 0.0f);
  ps.beta = V3(1.0f, 1.0f, 1.0f);
  cnt.segments++;
  cnt.queries++;
  // a camera far outside the scene: start the traversal in front of the scene's bounds (camera_entry_shift; d is a unit vector)
  trav_begin<Stack, DEFER>(ps.tr, st, sc, o + d * pp.camera_shift, d, kTMax, kNoPrim);
  return false;
}

// Start the path segment that leaves the current vertex (on primitive from_id) in direction d.
// Returns true when the bounce limit ends the path instead.
template <class Stack, bool DEFER = false>
PTK_HD bool path_next_segment(PathState& ps, Stack& st, const SceneView& sc, const PathParams& pp, const FlagView& fl,
                              v3 d, uint32_t from_id, Counters& cnt)
{
  ps.bits += 1u << 8;  // one level deeper (max_bounce < 2^23 keeps the field from overflowing)
  ps.shadow = false;
  if (!fl.indirect_light_sampling || ps.depth() >= pp.max_bounce) return true;
  cnt.segments++;
  cnt.queries++;
  trav_begin<Stack, DEFER>(ps.tr, st, sc, ps.tr.o, d, kTMax, from_id);
  return false;
}

// Next-event estimation set-up, Renderer::sample_lights (src/renderer.cpp:130-171): one uniformly chosen
// emissive primitive, weight = area * cos(light) / dist^2 * light_count, BSDF evaluated in the LIGHT's
// frame, no cosine at the shading point and no 1/pi -- all as the reference does it.  Everything except
// the visibility query is evaluated here; the result waits in ps.pending until the shadow ray returns.
// Returns false when no shadow ray is needed.
template <class M, class Mem>
PTK_HD bool nee_setup(PathState& ps, const SceneView& sc, const FlagView& fl, const M& m, const Surface& s,
                      v3 albedo, v3 incoming, v3 scale, v3& to_light_out, float& distance_out, float& t_stop_out,
                      uint32_t& light_id_out, v3& pending_out, const Mem& mem)
{
  if (sc.n_lights == 0) return false;
  int idx = (int)(rnd(ps.g) * (float)sc.n_lights);  // Scene::random_light, src/scene.cpp:64-71
  if (idx > sc.n_lights - 1) idx = sc.n_lights - 1;
  const auto& light = mem.light(sc, idx);  // by reference: fields are fetched where they are used, not all 28 words at once
  float r1 = rnd(ps.g), r2 = rnd(ps.g);
  v3 q, normal;
  if (light.kind == kTriangle) {  // Primitive::sample_point, src/geometry.cpp:228-241
    float t = sqrtf(r1);
    float u = 1.0f - t, v = r2 * t, w = 1.0f - u - v;
    q = ld3(light.p0) * u + ld3(light.p1) * v + ld3(light.p2) * w;
    // what the shadow ray's hit record will carry: a light keeps its interpolated, unflipped normal (of three equal
    // normals that is the normal itself, up to the rounding of u + v + w)
    normal = ld3(light.n0);
    if (!light.flat) normal = normal * u + ld3(light.n1) * v + ld3(light.n2) * w;
  } else {
    v3 c = ld3(light.p0);
    v3 n = normalize(s.point - c);
    v3 dir = unit_vector_from(r1, r2);
    if (!(dot(dir, n) > 0.0f)) dir = -dir;
    q = c + dir * light.radius;
    normal = dir;  // refined below
  }
  v3 to_light = q - s.point;
  float distance = length(to_light);
  to_light = to_light * frcp(distance);
  if (s.id == light.id) return false;  // src/renderer.cpp:146 (the reference traces the ray first; the result is unused)
  if (light.kind == kSphere) {
    // The shadow ray meets the sphere at its FIRST intersection, which is not the sampled point when that
    // point lies beyond the silhouette; the hit record's normal is taken there and faces the ray.
    v3 oc = ld3(light.p0) - s.point;
    float h = dot(to_light, oc);
    v3 l = oc - to_light * h;
    float disc = light.radius * light.radius - dot(l, l);
    float t = h - sqrtf(fmaxf(disc, 0.0f));
    if (!(t > kTMin)) t = h + sqrtf(fmaxf(disc, 0.0f));
    normal = ((s.point + to_light * t) - ld3(light.p0)) * (1.0f / light.radius);
    if (dot(to_light, normal) > 0.0f) normal = -normal;
  }
#ifdef PTK_WHATIF_NO_NEE_MATH  // (what-if build for weighing the light term's arithmetic: wrong radiance, same rays)
  pending_out = scale;
  light_id_out = light.id; to_light_out = to_light; distance_out = distance;
  t_stop_out = light.kind == kTriangle ? distance * 0.999f - 1e-3f : -1.0f;
  return true;
#endif
  float light_cos = fmaxf(dot(normal, -to_light), 0.0f);
  float weight = light.area * light_cos * frcp(distance * distance);
  // Only the microfacet model looks at the directions; every other BxDF::eval is a constant
  // (src/material.cpp:153,164,242), so the light's frame is built for that model alone.
  v3 f;
  if (m.type == 16u) {
    Frame fr = make_frame(normal);
    f = bsdf_eval(fl, m, albedo, to_local(fr, -incoming), to_local(fr, to_light));
  } else {
    f = bsdf_eval(fl, m, albedo, V3(0.0f, 1.0f, 0.0f), V3(0.0f, 1.0f, 0.0f));
  }
  pending_out = scale * ld3(light.emission) * f * (weight * (float)sc.n_lights);
  light_id_out = light.id;
  to_light_out = to_light;
  distance_out = distance;
  // A flat light is met by this ray at `distance` and nowhere else, so any other primitive hit clearly
  // before that decides the query (the light is not the closest hit) and the traversal may stop there.
  // A sphere light can be met before the sampled point, so its shadow rays run to the end.
  t_stop_out = light.kind == kTriangle ? distance * 0.999f - 1e-3f : -1.0f;
  return true;
}

// A path segment's query has finished: shade the vertex it found.  Returns true when the path ends here.
template <class Stack, class Mem = GlobalMem, class Park = ParkInState, bool DEFER = false>
PTK_HD bool path_shade(PathState& ps, Stack& st, const SceneView& sc, const FlagView& fl, const PathParams& pp,
                       Counters& cnt, const Mem& mem = Mem(), const Park& park = Park())
{
  const v3 o = ps.tr.o, d = ps.tr.d;
#ifdef PTK_BURN_SHADE
  for (int burn = 0; burn < PTK_BURN_SHADE; burn++) PTK_BURN12(ps.tr.node);
#endif
  if (ps.tr.hit.id == kNoPrim) {
    ps.L += ps.beta * background(sc, d);  // src/renderer.cpp:67-69
    return true;
  }
  PTK_MARK("shade.surface");
  Surface s;
  make_surface(sc, o, d, ps.tr.hit, s, mem);
  const auto& m = mem.material(sc, s.material);
  if (fl.debug_normal) {  // normal_as_color, src/renderer.cpp:17,74-76
    ps.L += ps.beta * (V3(s.normal.x + 1.0f, s.normal.y + 1.0f, s.normal.z + 1.0f) * 0.5f);
    return true;
  }
  PTK_MARK("shade.albedo");
  v3 albedo = surface_albedo(sc, m, s);
  v3 emission = ld3(m.emission);
  PTK_MARK("shade.rr");
  float rr_weight = 1.0f;
  if (fl.russian_roulette && ps.depth() > 3) {  // src/renderer.cpp:78-90
    float p = luma(albedo);
    if (rnd(ps.g) >= p) {
      ps.L += ps.beta * emission;
      return true;
    }
    rr_weight = frcp(p);
  }
  PTK_MARK("shade.frame");
  Frame fr = make_frame(s.normal);
  const LocalOut out = {fr, d};
  PTK_MARK("shade.sample");
  v3 wi = bsdf_sample(fl, m, s.inside, out, ps.g);
  PTK_MARK("shade.eval");
  bool specular = (m.flags & kMatSpecular) != 0;
  v3 scale = ps.beta * rr_weight;
  if (!fl.direct_light_sampling || ps.depth() == 0 || specular || ps.prev_specular) ps.L += scale * emission;
  v3 next_d = to_world(fr, wi);  // not renormalised, as in the reference (src/renderer.cpp:121)
  ps.beta = scale * (m.type == 16u ? bsdf_eval(fl, m, albedo, out.get(), wi) : bsdf_eval(fl, m, albedo, wi, wi));  // only the microfacet model reads V
  ps.prev_specular = specular;
  ps.tr.o = s.point;
  v3 to_light, pending;
  float distance, t_stop;
  uint32_t light_id;
  PTK_MARK("shade.nee");
  if (fl.direct_light_sampling && !specular &&
      nee_setup(ps, sc, fl, m, s, albedo, d, scale, to_light, distance, t_stop, light_id, pending, mem)) {
    ps.shadow = true;
    park.store(ps, pending, next_d);
    cnt.queries++;
    // the light is at `distance`: nothing farther can be the closest hit that counts
    // the light the ray must reach stays in the query (Traversal::target) until the result is used
    trav_begin<Stack, DEFER>(ps.tr, st, sc, s.point, to_light, distance * 1.001f + 1e-3f, s.id, light_id, t_stop);
    return false;
  }
  return path_next_segment<Stack, DEFER>(ps, st, sc, pp, fl, next_d, s.id, cnt);
}

// A shadow ray's query has finished: add the light's contribution if the light is what it reached,
// then start the path's next segment.  Returns true when the path ends here.
template <class Stack, class Park = ParkInState, bool DEFER = false>
PTK_HD bool path_consume_shadow(PathState& ps, Stack& st, const SceneView& sc, const FlagView& fl, const PathParams& pp,
                                Counters& cnt, const Park& park = Park())
{
  v3 pending, next_d;
  park.load(ps, pending, next_d);
  if (ps.tr.hit.id == ps.tr.target) ps.L += pending;  // src/renderer.cpp:146
  // the shadow ray left the vertex the next segment leaves: same origin, same primitive to skip
  return path_next_segment<Stack, DEFER>(ps, st, sc, pp, fl, next_d, ps.tr.skip_id, cnt);
}

// The lane's query has finished: use its result and set up the next query of the path.  Returns
// true when the path has ended; its radiance has then been added to the accumulator and the lane is free.
template <class Stack, class Accum>
PTK_HD bool path_consume(PathState& ps, Stack& st, const SceneView& sc, const FlagView& fl, const PathParams& pp,
                         Accum& acc, Counters& cnt)
{
  bool ended;
  if (ps.shadow) {
    if (ps.tr.hit.id == ps.tr.target) ps.L += ps.pending;  // src/renderer.cpp:146
    ended = path_next_segment(ps, st, sc, pp, fl, ps.next_d, ps.tr.skip_id, cnt);
  } else {
    ended = path_shade(ps, st, sc, fl, pp, cnt);
  }
  if (ended) acc.add(ps.slot(), ps.L);
  return ended;
}

// One pixel run to completion by one lane on its own (the host-compiled checks use this; the
// kernel interleaves the same steps across a wave).
template <class Stack, bool STATS>
PTK_HD v3 render_pixel(const SceneView& sc, const CameraView& cam, const FlagView& fl, int px, int py, int samples,
                       int max_bounce, int sample_first, int sample_stride, uint32_t seed, Stack& st, Counters& cnt)
{
  PathParams pp;
  pp.x0 = px; pp.y0 = py; pp.slot_bits = 0; pp.row_bits = 0;
  pp.x_end = px + 1; pp.y_end = py + 1;
  pp.samples = samples; pp.max_bounce = max_bounce; pp.sample_first = sample_first; pp.sample_stride = sample_stride;
  pp.seed = seed;
  pp.camera_shift = camera_entry_shift(sc, cam);
  PathState ps;
  path_init(ps);
  SumAccum acc;
  acc.sum = V3(0.0f, 0.0f, 0.0f);
  uint32_t next = 0;
  bool need = true;
  for (;;) {
    while (need) need = path_begin_item(ps, st, sc, cam, pp, next++, cnt);
    if (!ps.alive) break;
    while (!trav_done(ps.tr)) {
      if (trav_at_inner(ps.tr)) trav_inner<Stack, STATS>(ps.tr, st, sc, cnt);
      else trav_leaf<Stack, STATS>(ps.tr, st, sc, cnt);
    }
    need = path_consume(ps, st, sc, fl, pp, acc, cnt);
  }
  return acc.sum;
}

}  // namespace ptk
