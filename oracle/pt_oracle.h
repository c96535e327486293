/*
 * pt_oracle.h -- CPU restatement of the reference path tracer's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library links, loads or
 * calls this code; it exists so that tests, the smoke check and the
 * benchmark's cpu_baseline leg have an independent statement of what the
 * reference computes.
 *
 * PARITY UNPINNED by reference tests: the reference holds no tests, golden
 * vectors or fixtures, and it cannot be built offline (its CMakeLists.txt
 * fetches glm 1.0.0 and nlohmann/json 3.11.3 from the network).  The only
 * reference outputs available are (1) the renders under doc/results/ and
 * (2) counters / the Cornell BVH topology recorded from a reference run
 * when the reference was surveyed; tests/ checks this file against both.
 * Of the one published render of cornell_box.json the tests use the light, the
 * walls, the textured floor, the field of view and the mirror and the glass
 * sphere (mirror reflection, dielectric refraction, environment lookup), on
 * pixels that do not depend on the bunny missing from the checkout.  Nothing
 * the reference holds pins hit ids or distances, the microfacet model or
 * sphere lights: for those this file stands on its line-by-line citations.
 *
 * Arithmetic follows the reference: double precision everywhere except the
 * places where the reference itself drops to float (shading frame, frame
 * inverse, frame * vector products, the plane offset `d` of the triangle
 * test).  Vector formulas are those published for glm 1.0.0
 * (g-truc/glm @ 33b0eb9f, pinned in the reference's CMakeLists.txt:14-18).
 *
 * One deliberate difference: the reference draws random numbers from a
 * process-global, unseeded std::mt19937 (src/util.h:72-78, src/scene.cpp:64-71),
 * which makes its output irreproducible.  The oracle draws from a counter
 * based generator keyed by (seed, pixel, sample index, draw index) so that
 * renders are repeatable and independent of thread count.  Only the
 * distributions are part of the reference's behaviour.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Material::Type, src/material.h:12-18 */
enum {
  ORC_DIFFUSE = 1 << 0,
  ORC_SPECULAR = 1 << 1,
  ORC_DIELECTRIC = 1 << 2,
  ORC_MIRROR = 1 << 3,
  ORC_MICROFACET = 1 << 4
};

/* src/config.h:3-8 */
typedef struct {
  int debug_normal;            /* PT_DEBUG_NORMAL            default 0 */
  int russian_roulette;        /* PT_RUSSIAN_ROULETTE        default 1 */
  int direct_light_sampling;   /* PT_DIRECT_LIGHT_SAMPLING   default 1 */
  int indirect_light_sampling; /* PT_INDIRECT_LIGHT_SAMPLING default 1 */
  int importance_sample;       /* PT_IMPORTANCE_SAMPLE       default 1 */
  int check_hemisphere;        /* PT_CHECK_HEMISPHERE        default 1 */
} orc_flags;

/* Intersection, src/geometry.h:14-24 (material pointer -> index) */
typedef struct {
  int32_t hit; /* 0 = miss */
  uint32_t id;
  double t;
  double point[3];
  double normal[3];
  double uv[2];
  int32_t material;
  int32_t inside;
} orc_hit;

typedef struct {
  uint64_t segments;   /* trace_ray bodies,  renderer.cpp:63  (bounce_counter) */
  uint64_t queries;    /* Scene::find_intersection calls (segments + shadow rays) */
  uint64_t prim_tests; /* Primitive::intersect calls, geometry.cpp:162 */
  uint64_t box_tests;  /* ray_vs_aabb calls */
  uint64_t camera_rays;
} orc_counters;

typedef struct orc_scene orc_scene;

typedef struct {
  int width, height;
  double world_up[3], forward[3], position[3], right[3], up[3];
  double fov; /* radians */
  double focus_distance;
  double aperture; /* degrees (defocus angle) */
} orc_camera;

orc_flags orc_default_flags(void);

orc_scene* orc_scene_new(void);
void orc_scene_free(orc_scene*);
/* rows bottom-up, 3 bytes per pixel: the layout Image::load produces (image.cpp:25-31) */
int orc_add_texture(orc_scene*, const uint8_t* rgb, int width, int height);
int orc_add_material(orc_scene*, int type, const double albedo[3], const double emission[3],
                     double refraction_index, double roughness, double metallic, int texture);
/* pos, nrm: n*9 doubles (v0 v1 v2); uv: n*6 doubles; material: n indices.  ids continue the scene counter. */
void orc_add_triangles(orc_scene*, int n, const double* pos, const double* nrm, const double* uv,
                       const int32_t* material);
void orc_add_sphere(orc_scene*, const double center[3], double radius, int material);
void orc_set_background_texture(orc_scene*, int texture);
void orc_set_background_color(orc_scene*, const double color[3]);
void orc_compute_bvh(orc_scene*);
int orc_primitive_count(const orc_scene*);
int orc_light_count(const orc_scene*);
/* light_ids: out, at most cap entries */
int orc_light_ids(const orc_scene*, uint32_t* light_ids, int cap);

/* BVH topology dump, preorder.  Per node 4 ints: is_leaf, first, count, 0 for leaves
 * (first = index into prim_order), or 0, left_node, right_node, 0 for inner nodes.
 * Returns node count (writes at most cap nodes). prim_order gets the primitive ids in BVH order. */
int orc_bvh_dump(const orc_scene*, int32_t* nodes, int cap, uint32_t* prim_order);

/* rays: n*6 doubles (origin, direction). */
void orc_intersect_batch(const orc_scene*, const double* rays, int n, orc_hit* hits, orc_counters* counters);

void orc_camera_init(orc_camera*, int width, int height, double fov_degrees, double aperture, double focus_distance);
void orc_camera_look_at(orc_camera*, const double position[3], const double target[3]);
/* Camera::get_ray with explicit random inputs: jitter_rnd in [0,1)^2, disk in the unit disk. */
void orc_camera_ray(const orc_camera*, int x, int y, const double jitter_rnd[2], const double disk[2], double ray[6]);

/* Renderer::render: running mean over `samples` more samples per pixel into buffer (W*H*3 doubles),
 * given `total_samples` already accumulated.  Sample indices are global (total_samples + s). */
void orc_render(const orc_scene*, const orc_camera*, const orc_flags*, int samples, int max_bounce, uint32_t seed,
                int total_samples, double* buffer, int threads, orc_counters* counters);
/* Same, restricted to pixel rows [y0,y1). */
void orc_render_rows(const orc_scene*, const orc_camera*, const orc_flags*, int samples, int max_bounce, uint32_t seed,
                     int total_samples, double* buffer, int y0, int y1, int threads, orc_counters* counters);

/* Renderer::save_image's pixel transform (ACES, gamma, u8 truncation); out: W*H*3 bytes, row 0 = top. */
void orc_tonemap_u8(const double* buffer, int n_pixels, uint8_t* out);

/* ---- pure functions exposed for known-answer tests ---- */
void orc_local_to_world(const double up[3], float m[9]);                 /* column-major r,u,f; util.h:63-70 */
void orc_mat3_inverse(const float m[9], float out[9]);                   /* glm::inverse(mat3) */
void orc_equirectangular(const double v[3], double uv[2]);               /* util.h:142-149 */
void orc_image_sample(const orc_scene*, int texture, double u, double v, double rgb[3]); /* image.cpp:62-85 */
void orc_sample_background(const orc_scene*, const double dir[3], double rgb[3]);       /* scene.cpp:33-45 */
double orc_reflectance(double cosine, double refraction_index);          /* material.cpp:100-106 */
void orc_aces_tone_map(const double x[3], double out[3]);                /* renderer.cpp:174-178 */
int orc_ray_vs_triangle(const double ray[6], const double tri[9], double* t);   /* geometry.cpp:108-157 */
int orc_ray_vs_sphere(const double ray[6], const double center[3], double radius, double* t); /* :28-49 */
int orc_ray_vs_aabb(const double ray[6], const double bmin[3], const double bmax[3], double tmin, double tmax); /* aabb.cpp:45-65 */
/* BxDF::eval for a material (no texture) in local space, material.cpp:124-136 */
void orc_bxdf_eval(const orc_flags*, int type, const double albedo[3], double roughness, double metallic,
                   const double wo[3], const double wi[3], double out[3]);
/* BxDF::sample with explicit random numbers r[0..2] */
void orc_bxdf_sample(const orc_flags*, int type, double roughness, double refraction_index, int inside,
                     const double wo[3], const double r[3], double wi[3]);
/* the counter-based generator: draw `n` uniforms of path (pixel, sample) */
void orc_rng_draws(uint32_t seed, uint32_t pixel, uint32_t sample, int n, double* out);

#ifdef __cplusplus
}
#endif
#endif
