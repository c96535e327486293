/*
 * pt_oracle.c -- CPU restatement of the reference path tracer's hot path.
 * TEST INFRASTRUCTURE ONLY (see pt_oracle.h).  PARITY UNPINNED by reference
 * tests (the reference has none); pinned only by the checks described in the
 * header.  Each function cites the reference lines it follows; paths are
 * relative to the reference checkout.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC pt_oracle.c -lm
 * (-ffp-contract=off: the reference is built for baseline x86-64, which has
 * no fused multiply-add, so no contraction may happen here either.)
 */
#include "pt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* vectors (glm 1.0.0 closed forms)                                    */
/* ------------------------------------------------------------------ */
typedef struct { double x, y, z; } d3;
typedef struct { float x, y, z; } f3;
typedef struct { float c[3][3]; } m3f; /* c[col][row], column-major like glm::mat3 */

static const double ORC_PI = 3.14159265359; /* util.h:12 */

static inline d3 D3(double x, double y, double z) { d3 r = {x, y, z}; return r; }
static inline d3 add(d3 a, d3 b) { return D3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline d3 sub(d3 a, d3 b) { return D3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline d3 mulv(d3 a, d3 b) { return D3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline d3 muls(d3 a, double s) { return D3(a.x * s, a.y * s, a.z * s); }
static inline d3 divs(d3 a, double s) { return D3(a.x / s, a.y / s, a.z / s); }
static inline d3 neg(d3 a) { return D3(-a.x, -a.y, -a.z); }
static inline double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline d3 cross(d3 x, d3 y)
{
  return D3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline double length(d3 a) { return sqrt(dot(a, a)); }
/* glm::normalize = v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x) */
static inline d3 normalize(d3 a) { return muls(a, 1.0 / sqrt(dot(a, a))); }
static inline d3 vmin(d3 a, d3 b) { return D3(b.x < a.x ? b.x : a.x, b.y < a.y ? b.y : a.y, b.z < a.z ? b.z : a.z); }
static inline d3 vmax(d3 a, d3 b) { return D3(a.x < b.x ? b.x : a.x, a.y < b.y ? b.y : a.y, a.z < b.z ? b.z : a.z); }
static inline double comp(d3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
static inline d3 reflect(d3 I, d3 N) { return sub(I, muls(N, dot(N, I) * 2.0)); }
static inline d3 refract(d3 I, d3 N, double eta)
{
  double dn = dot(N, I);
  double k = 1.0 - eta * eta * (1.0 - dn * dn);
  if (k >= 0.0) return sub(muls(I, eta), muls(N, eta * dn + sqrt(k)));
  return D3(0, 0, 0);
}
static inline d3 vpow(d3 a, double e) { return D3(pow(a.x, e), pow(a.y, e), pow(a.z, e)); }
static inline double sq(double x) { return x * x; }

static inline f3 F3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 to_f3(d3 a) { return F3((float)a.x, (float)a.y, (float)a.z); }
static inline d3 to_d3(f3 a) { return D3(a.x, a.y, a.z); }
static inline float dotf(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline f3 crossf(f3 x, f3 y)
{
  return F3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline f3 normalizef(f3 a)
{
  float s = 1.0f / sqrtf(dotf(a, a));
  return F3(a.x * s, a.y * s, a.z * s);
}
/* glm: mat3 * vec3, computed in the MATRIX's precision (float) even when the
 * caller passes a dvec3 (renderer.cpp:100,121,154-155). */
static inline f3 m3f_mul(const m3f* m, f3 v)
{
  return F3(m->c[0][0] * v.x + m->c[1][0] * v.y + m->c[2][0] * v.z,
            m->c[0][1] * v.x + m->c[1][1] * v.y + m->c[2][1] * v.z,
            m->c[0][2] * v.x + m->c[1][2] * v.y + m->c[2][2] * v.z);
}
/* glm::inverse(mat3): adjugate over determinant (compute_inverse<3,3>) */
static m3f m3f_inverse(const m3f* pm)
{
#define M(c_, r_) pm->c[c_][r_]
  float ood = 1.0f / (+M(0, 0) * (M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2)) - M(1, 0) * (M(0, 1) * M(2, 2) - M(2, 1) * M(0, 2)) +
                      M(2, 0) * (M(0, 1) * M(1, 2) - M(1, 1) * M(0, 2)));
  m3f inv;
  inv.c[0][0] = +(M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2)) * ood;
  inv.c[1][0] = -(M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2)) * ood;
  inv.c[2][0] = +(M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1)) * ood;
  inv.c[0][1] = -(M(0, 1) * M(2, 2) - M(2, 1) * M(0, 2)) * ood;
  inv.c[1][1] = +(M(0, 0) * M(2, 2) - M(2, 0) * M(0, 2)) * ood;
  inv.c[2][1] = -(M(0, 0) * M(2, 1) - M(2, 0) * M(0, 1)) * ood;
  inv.c[0][2] = +(M(0, 1) * M(1, 2) - M(1, 1) * M(0, 2)) * ood;
  inv.c[1][2] = -(M(0, 0) * M(1, 2) - M(1, 0) * M(0, 2)) * ood;
  inv.c[2][2] = +(M(0, 0) * M(1, 1) - M(1, 0) * M(0, 1)) * ood;
#undef M
  return inv;
}

/* util.h:63-70.  normalize(up) is evaluated in double, then everything is float. */
static m3f local_to_world(d3 up)
{
  f3 u = to_f3(normalize(up));
  f3 reference = (fabsf(u.z) < 0.9f) ? F3(0, 0, 1) : F3(1, 0, 0);
  f3 r = normalizef(crossf(reference, u));
  f3 f = crossf(u, r);
  m3f m;
  m.c[0][0] = r.x; m.c[0][1] = r.y; m.c[0][2] = r.z;
  m.c[1][0] = u.x; m.c[1][1] = u.y; m.c[1][2] = u.z;
  m.c[2][0] = f.x; m.c[2][1] = f.y; m.c[2][2] = f.z;
  return m;
}

/* ------------------------------------------------------------------ */
/* counter-based random numbers (replaces util.h:72-78, scene.cpp:64-71) */
/* ------------------------------------------------------------------ */
typedef struct { uint32_t k0, k1, i; } rng_t;

static inline uint32_t mix32(uint32_t x)
{
  x ^= x >> 16; x *= 0x21f0aaadu;
  x ^= x >> 15; x *= 0x735a2d97u;
  x ^= x >> 15;
  return x;
}
static inline rng_t rng_init(uint32_t seed, uint32_t pixel, uint32_t sample)
{
  rng_t r;
  r.k0 = mix32(pixel + mix32(seed + 0x9E3779B9u));
  r.k1 = mix32(sample + 0x85EBCA6Bu) ^ ((r.k0 << 13) | (r.k0 >> 19));
  r.i = 0;
  return r;
}
/* uniform in [0,1) with 24 random bits: exactly representable in float and double */
static inline double random_double(rng_t* r)
{
  uint32_t x = mix32((r->k0 + r->i * 0x9E3779B9u) ^ r->k1);
  r->i++;
  return (double)(x >> 8) * (1.0 / 16777216.0);
}

/* ------------------------------------------------------------------ */
/* scene                                                               */
/* ------------------------------------------------------------------ */
typedef struct { int w, h; uint8_t* data; } image_t;

typedef struct { /* Material, material.h:11-36 */
  int type;
  d3 albedo, emission;
  double refraction_index, roughness, metallic;
  int texture; /* -1 = none */
} material_t;

enum { PRIM_SPHERE = 0, PRIM_TRIANGLE = 1 };

typedef struct { /* Primitive, geometry.h:54-71 */
  int type;
  d3 v0, v1, v2, n0, n1, n2;
  double t0[2], t1[2], t2[2];
  d3 center; double radius;
  int material;
  d3 bbmin, bbmax;
  uint32_t id;
} prim_t;

typedef struct { /* BVH::Node, bvh.h:14-21 */
  int left, right; /* -1,-1 for a leaf */
  d3 bbmin, bbmax;
  int begin, end; /* leaf range in bvh_prims */
} node_t;

struct orc_scene {
  image_t* textures; int n_textures;
  material_t* materials; int n_materials;
  prim_t* prims; int n_prims, cap_prims;
  int* lights; int n_lights; /* indices into prims, in insertion order (scene.cpp:60) */
  int background_texture; d3 background_color; /* scene.cpp:15: colour defaults to -1 */
  prim_t* bvh_prims; /* BVH owns a sorted copy, bvh.cpp:47 */
  node_t* nodes; int n_nodes, cap_nodes;
};

orc_flags orc_default_flags(void)
{
  orc_flags f = {0, 1, 1, 1, 1, 1};
  return f;
}

orc_scene* orc_scene_new(void)
{
  orc_scene* s = (orc_scene*)calloc(1, sizeof(orc_scene));
  s->background_texture = -1;
  s->background_color = D3(-1.0, -1.0, -1.0);
  return s;
}

void orc_scene_free(orc_scene* s)
{
  if (!s) return;
  for (int i = 0; i < s->n_textures; i++) free(s->textures[i].data);
  free(s->textures); free(s->materials); free(s->prims); free(s->lights); free(s->bvh_prims); free(s->nodes);
  free(s);
}

int orc_add_texture(orc_scene* s, const uint8_t* rgb, int width, int height)
{
  s->textures = (image_t*)realloc(s->textures, sizeof(image_t) * (size_t)(s->n_textures + 1));
  image_t* im = &s->textures[s->n_textures];
  im->w = width; im->h = height;
  size_t bytes = (size_t)width * (size_t)height * 3;
  im->data = (uint8_t*)malloc(bytes ? bytes : 1);
  memcpy(im->data, rgb, bytes);
  return s->n_textures++;
}

int orc_add_material(orc_scene* s, int type, const double albedo[3], const double emission[3], double refraction_index,
                     double roughness, double metallic, int texture)
{
  s->materials = (material_t*)realloc(s->materials, sizeof(material_t) * (size_t)(s->n_materials + 1));
  material_t* m = &s->materials[s->n_materials];
  m->type = type;
  m->albedo = D3(albedo[0], albedo[1], albedo[2]);
  m->emission = D3(emission[0], emission[1], emission[2]);
  m->refraction_index = refraction_index; m->roughness = roughness; m->metallic = metallic;
  m->texture = texture;
  return s->n_materials++;
}

/* Primitive::is_light, geometry.cpp:225 */
static int is_light(const orc_scene* s, const prim_t* p)
{
  d3 e = s->materials[p->material].emission;
  return e.x > 0.0 || e.y > 0.0 || e.z > 0.0;
}

/* Scene::add_primitive, scene.cpp:56-62 */
static void add_primitive(orc_scene* s, prim_t* p)
{
  if (s->n_prims == s->cap_prims) {
    s->cap_prims = s->cap_prims ? s->cap_prims * 2 : 64;
    s->prims = (prim_t*)realloc(s->prims, sizeof(prim_t) * (size_t)s->cap_prims);
  }
  p->id = (uint32_t)s->n_prims;
  if (is_light(s, p)) {
    s->lights = (int*)realloc(s->lights, sizeof(int) * (size_t)(s->n_lights + 1));
    s->lights[s->n_lights++] = s->n_prims;
  }
  s->prims[s->n_prims++] = *p;
}

void orc_add_triangles(orc_scene* s, int n, const double* pos, const double* nrm, const double* uv,
                       const int32_t* material)
{
  for (int i = 0; i < n; i++) {
    prim_t p;
    memset(&p, 0, sizeof p);
    p.type = PRIM_TRIANGLE;
    const double* q = pos + 9 * (size_t)i;
    p.v0 = D3(q[0], q[1], q[2]); p.v1 = D3(q[3], q[4], q[5]); p.v2 = D3(q[6], q[7], q[8]);
    q = nrm + 9 * (size_t)i;
    p.n0 = D3(q[0], q[1], q[2]); p.n1 = D3(q[3], q[4], q[5]); p.n2 = D3(q[6], q[7], q[8]);
    q = uv + 6 * (size_t)i;
    p.t0[0] = q[0]; p.t0[1] = q[1]; p.t1[0] = q[2]; p.t1[1] = q[3]; p.t2[0] = q[4]; p.t2[1] = q[5];
    p.material = material[i];
    /* AABB(const Triangle&), aabb.cpp:17-21 */
    p.bbmin = vmin(vmin(p.v0, p.v1), p.v2);
    p.bbmax = vmax(vmax(p.v0, p.v1), p.v2);
    add_primitive(s, &p);
  }
}

void orc_add_sphere(orc_scene* s, const double center[3], double radius, int material)
{
  prim_t p;
  memset(&p, 0, sizeof p);
  p.type = PRIM_SPHERE;
  p.center = D3(center[0], center[1], center[2]);
  p.radius = radius;
  p.material = material;
  /* AABB(const Sphere&), aabb.cpp:11-15 */
  p.bbmin = D3(p.center.x - radius, p.center.y - radius, p.center.z - radius);
  p.bbmax = D3(p.center.x + radius, p.center.y + radius, p.center.z + radius);
  add_primitive(s, &p);
}

void orc_set_background_texture(orc_scene* s, int texture) { s->background_texture = texture; }
void orc_set_background_color(orc_scene* s, const double c[3]) { s->background_color = D3(c[0], c[1], c[2]); }
int orc_primitive_count(const orc_scene* s) { return s->n_prims; }
int orc_light_count(const orc_scene* s) { return s->n_lights; }
int orc_light_ids(const orc_scene* s, uint32_t* ids, int cap)
{
  for (int i = 0; i < s->n_lights && i < cap; i++) ids[i] = s->prims[s->lights[i]].id;
  return s->n_lights;
}

/* ------------------------------------------------------------------ */
/* BVH build, bvh.cpp:54-83                                            */
/* ------------------------------------------------------------------ */

/* std::sort as libstdc++ (GCC 11) implements it, because the reference sorts
 * with a comparator that has many ties on axis-aligned scenes (bvh.cpp:69-70)
 * and the resulting order decides the tree: introsort (median-of-3 pivot on
 * first+1, mid, last-1; depth limit 2*floor(log2 n); threshold 16) followed
 * by a final insertion sort.  For n <= 16 this is a plain, stable insertion
 * sort. */
typedef struct { prim_t* a; int axis; } sort_ctx;
static inline int less_at(const sort_ctx* c, const prim_t* x, const prim_t* y)
{
  return comp(x->bbmin, c->axis) < comp(y->bbmin, c->axis);
}
static inline void swap_prim(prim_t* x, prim_t* y) { prim_t t = *x; *x = *y; *y = t; }

static void sift_down(const sort_ctx* c, prim_t* first, long hole, long len, prim_t value)
{ /* std::__adjust_heap */
  const long top = hole;
  long child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (less_at(c, first + child, first + (child - 1))) child--;
    first[hole] = first[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  long parent = (hole - 1) / 2; /* std::__push_heap */
  while (hole > top && less_at(c, first + parent, &value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}
static void heap_sort(const sort_ctx* c, prim_t* first, prim_t* last)
{ /* std::__partial_sort(first,last,last) = heap_select + sort_heap */
  long len = last - first;
  if (len >= 2) {
    for (long parent = (len - 2) / 2;; parent--) {
      sift_down(c, first, parent, len, first[parent]);
      if (parent == 0) break;
    }
  }
  while (last - first > 1) {
    --last;
    prim_t value = *last;
    *last = *first;
    sift_down(c, first, 0, last - first, value);
  }
}
static void move_median_to_first(const sort_ctx* c, prim_t* result, prim_t* a, prim_t* b, prim_t* cc)
{
  if (less_at(c, a, b)) {
    if (less_at(c, b, cc)) swap_prim(result, b);
    else if (less_at(c, a, cc)) swap_prim(result, cc);
    else swap_prim(result, a);
  } else if (less_at(c, a, cc)) swap_prim(result, a);
  else if (less_at(c, b, cc)) swap_prim(result, cc);
  else swap_prim(result, b);
}
static prim_t* unguarded_partition(const sort_ctx* c, prim_t* first, prim_t* last, prim_t* pivot)
{
  for (;;) {
    while (less_at(c, first, pivot)) ++first;
    --last;
    while (less_at(c, pivot, last)) --last;
    if (!(first < last)) return first;
    swap_prim(first, last);
    ++first;
  }
}
static void introsort_loop(const sort_ctx* c, prim_t* first, prim_t* last, long depth_limit)
{
  while (last - first > 16) {
    if (depth_limit == 0) { heap_sort(c, first, last); return; }
    --depth_limit;
    prim_t* mid = first + (last - first) / 2;
    move_median_to_first(c, first, first + 1, mid, last - 1);
    prim_t* cut = unguarded_partition(c, first + 1, last, first);
    introsort_loop(c, cut, last, depth_limit);
    last = cut;
  }
}
static void unguarded_linear_insert(const sort_ctx* c, prim_t* last)
{
  prim_t val = *last;
  prim_t* next = last - 1;
  while (less_at(c, &val, next)) { *last = *next; last = next; --next; }
  *last = val;
}
static void insertion_sort(const sort_ctx* c, prim_t* first, prim_t* last)
{
  if (first == last) return;
  for (prim_t* i = first + 1; i != last; ++i) {
    if (less_at(c, i, first)) {
      prim_t val = *i;
      memmove(first + 1, first, (size_t)(i - first) * sizeof(prim_t));
      *first = val;
    } else unguarded_linear_insert(c, i);
  }
}
static void std_sort(prim_t* first, prim_t* last, int axis)
{
  sort_ctx c = {first, axis};
  if (first == last) return;
  long n = last - first, lg = 0;
  while ((n >> (lg + 1)) > 0) lg++;
  introsort_loop(&c, first, last, lg * 2);
  if (last - first > 16) {
    insertion_sort(&c, first, first + 16);
    for (prim_t* i = first + 16; i != last; ++i) unguarded_linear_insert(&c, i);
  } else insertion_sort(&c, first, last);
}

/* AABB::longest_axis, aabb.h:28-40 */
static int longest_axis(d3 bmin, d3 bmax)
{
  d3 s = sub(bmax, bmin);
  if (s.y >= s.x && s.y >= s.z) return 1;
  if (s.z >= s.x && s.z >= s.y) return 2;
  return 0;
}

static int new_node(orc_scene* s)
{
  if (s->n_nodes == s->cap_nodes) {
    s->cap_nodes = s->cap_nodes ? s->cap_nodes * 2 : 64;
    s->nodes = (node_t*)realloc(s->nodes, sizeof(node_t) * (size_t)s->cap_nodes);
  }
  return s->n_nodes++;
}

static int construct(orc_scene* s, int begin, int end)
{
  int ni = new_node(s);
  /* compute_bounding_volume, aabb.cpp:31-43 */
  d3 bmin = D3(+1e9, +1e9, +1e9), bmax = D3(-1e9, -1e9, -1e9);
  for (int i = begin; i < end; i++) {
    bmin = vmin(bmin, s->bvh_prims[i].bbmin);
    bmax = vmax(bmax, s->bvh_prims[i].bbmax);
  }
  int axis = longest_axis(bmin, bmax);
  int count = end - begin;
  int left = -1, right = -1;
  if (count > 5) { /* split_threshold, bvh.h:30 */
    std_sort(s->bvh_prims + begin, s->bvh_prims + end, axis);
    int middle = begin + count / 2;
    left = construct(s, begin, middle);
    right = construct(s, middle, end);
  }
  node_t* n = &s->nodes[ni]; /* after recursion: nodes may have been reallocated */
  n->bbmin = bmin; n->bbmax = bmax; n->left = left; n->right = right; n->begin = begin; n->end = end;
  return ni;
}

void orc_compute_bvh(orc_scene* s)
{
  free(s->bvh_prims); free(s->nodes);
  s->nodes = NULL; s->n_nodes = s->cap_nodes = 0;
  s->bvh_prims = (prim_t*)malloc(sizeof(prim_t) * (size_t)(s->n_prims ? s->n_prims : 1));
  memcpy(s->bvh_prims, s->prims, sizeof(prim_t) * (size_t)s->n_prims);
  construct(s, 0, s->n_prims);
}

int orc_bvh_dump(const orc_scene* s, int32_t* nodes, int cap, uint32_t* prim_order)
{
  for (int i = 0; i < s->n_nodes && i < cap; i++) {
    const node_t* n = &s->nodes[i];
    int leaf = n->left < 0;
    nodes[4 * i + 0] = leaf;
    nodes[4 * i + 1] = leaf ? n->begin : n->left;
    nodes[4 * i + 2] = leaf ? (n->end - n->begin) : n->right;
    nodes[4 * i + 3] = 0;
  }
  if (prim_order)
    for (int i = 0; i < s->n_prims; i++) prim_order[i] = s->bvh_prims[i].id;
  return s->n_nodes;
}

/* ------------------------------------------------------------------ */
/* intersection                                                        */
/* ------------------------------------------------------------------ */
typedef struct { d3 origin, direction; } ray_t;
static inline d3 point_at(const ray_t* r, double t) { return add(r->origin, muls(r->direction, t)); } /* ray.h:8 */

typedef struct { /* Intersection, geometry.h:14-24 */
  int hit;
  uint32_t id;
  double t;
  d3 point, normal;
  double uv[2];
  int material;
  int inside;
} isect_t;

/* aabb.cpp:45-65 */
static int ray_vs_aabb(const ray_t* r, d3 bbmin, d3 bbmax, double tmin, double tmax)
{
  for (int axis = 0; axis < 3; axis++) {
    double adinv = 1.0 / comp(r->direction, axis);
    double t0 = (comp(bbmin, axis) - comp(r->origin, axis)) * adinv;
    double t1 = (comp(bbmax, axis) - comp(r->origin, axis)) * adinv;
    if (t0 < t1) {
      if (t0 > tmin) tmin = t0;
      if (t1 < tmax) tmax = t1;
    } else {
      if (t1 > tmin) tmin = t1;
      if (t0 < tmax) tmax = t0;
    }
    if (tmax <= tmin) return 0;
  }
  return 1;
}

/* geometry.cpp:28-49; Interval::surrounds is the open interval (util.h:25) */
static int ray_vs_sphere(const ray_t* r, d3 center, double radius, double tmin, double tmax, double* t)
{
  d3 oc = sub(center, r->origin);
  double a = dot(r->direction, r->direction);
  double h = dot(r->direction, oc);
  double c = dot(oc, oc) - radius * radius;
  double discriminant = h * h - a * c;
  if (discriminant < 0) return 0;
  double sqrtd = sqrt(discriminant);
  double root = (h - sqrtd) / a;
  if (!(tmin < root && root < tmax)) {
    root = (h + sqrtd) / a;
    if (!(tmin < root && root < tmax)) return 0;
  }
  *t = root;
  return 1;
}

/* geometry.cpp:108-157 */
static int ray_vs_triangle(const ray_t* r, d3 tv0, d3 tv1, d3 tv2, double tmin, double* t_out)
{
  d3 v0v1 = sub(tv1, tv0);
  d3 v0v2 = sub(tv2, tv0);
  d3 N = cross(v0v1, v0v2);
  double NdotRayDirection = dot(N, r->direction);
  if (fabs(NdotRayDirection) < tmin) return 0;
  float d = (float)(-dot(N, tv0)); /* :125, single precision on purpose */
  double t = -(dot(N, r->origin) + (double)d) / NdotRayDirection;
  *t_out = t;
  if (t < tmin) return 0;
  d3 P = point_at(r, t);
  d3 Ne;
  d3 v0p = sub(P, tv0);
  Ne = cross(v0v1, v0p);
  if (dot(N, Ne) < 0) return 0;
  d3 v2v1 = sub(tv2, tv1);
  d3 v1p = sub(P, tv1);
  Ne = cross(v2v1, v1p);
  if (dot(N, Ne) < 0) return 0;
  d3 v2v0 = sub(tv0, tv2);
  d3 v2p = sub(P, tv2);
  Ne = cross(v2v0, v2p);
  if (dot(N, Ne) < 0) return 0;
  return 1;
}

/* geometry.cpp:71-84 */
static void barycentric(d3 a, d3 b, d3 c, d3 p, double* u, double* v, double* w)
{
  d3 v0 = sub(b, a), v1 = sub(c, a), v2 = sub(p, a);
  double d00 = dot(v0, v0);
  double d01 = dot(v0, v1);
  double d11 = dot(v1, v1);
  double d20 = dot(v2, v0);
  double d21 = dot(v2, v1);
  double inv_denom = 1.0 / (d00 * d11 - d01 * d01);
  *v = (d11 * d20 - d01 * d21) * inv_denom;
  *w = (d00 * d21 - d01 * d20) * inv_denom;
  *u = 1.0 - *v - *w;
}

/* Primitive::intersect, geometry.cpp:159-223 */
static int prim_intersect(const orc_scene* s, const prim_t* p, const ray_t* ray, isect_t* out, orc_counters* cnt)
{
  cnt->prim_tests++;
  double t;
  const double ti_min = 0.00001, ti_max = 1e9;
  const material_t* mat = &s->materials[p->material];
  if (p->type == PRIM_SPHERE) {
    if (!ray_vs_sphere(ray, p->center, p->radius, ti_min, ti_max, &t)) return 0;
    out->hit = 1; out->id = p->id; out->t = t;
    out->point = point_at(ray, t);
    d3 normal = divs(sub(out->point, p->center), p->radius);
    if (dot(ray->direction, normal) > 0.0) { out->normal = neg(normal); out->inside = 0; }
    else { out->normal = normal; out->inside = 1; }
    out->material = p->material;
    out->uv[0] = out->uv[1] = 0.0;
    if (mat->texture >= 0) { /* Sphere::texcoord, geometry.cpp:56-64 */
      d3 q = normalize(sub(out->point, p->center));
      double theta = acos(-q.y);
      double phi = atan2(-q.z, q.x) + ORC_PI;
      out->uv[0] = phi / (2 * ORC_PI);
      out->uv[1] = theta / ORC_PI;
    }
    return 1;
  }
  if (!ray_vs_triangle(ray, p->v0, p->v1, p->v2, ti_min, &t)) return 0;
  out->hit = 1; out->id = p->id; out->t = t;
  out->point = point_at(ray, t);
  double u, v, w;
  barycentric(p->v0, p->v1, p->v2, out->point, &u, &v, &w);
  /* Triangle::normal(point): interpolated, NOT normalised, geometry.cpp:100-105 */
  d3 normal = add(add(muls(p->n0, u), muls(p->n1, v)), muls(p->n2, w));
  if (is_light(s, p)) { out->normal = normal; out->inside = 1; }
  else if (dot(ray->direction, normal) > 0.0) { out->normal = neg(normal); out->inside = 0; }
  else { out->normal = normal; out->inside = 1; }
  out->material = p->material;
  out->uv[0] = out->uv[1] = 0.0;
  if (mat->texture >= 0) { /* Triangle::texcoord, geometry.cpp:86-91 */
    out->uv[0] = u * p->t0[0] + v * p->t1[0] + w * p->t2[0];
    out->uv[1] = u * p->t0[1] + v * p->t1[1] + w * p->t2[1];
  }
  return 1;
}

/* BVH::Node::intersect + intersect_primitives + closest, bvh.cpp:10-45, geometry.cpp:256-271 */
static int node_intersect(const orc_scene* s, int ni, const ray_t* ray, isect_t* out, orc_counters* cnt)
{
  const node_t* n = &s->nodes[ni];
  cnt->box_tests++;
  if (!ray_vs_aabb(ray, n->bbmin, n->bbmax, 0.01, 1e9)) return 0;
  if (n->left < 0) {
    int have = 0;
    for (int i = n->begin; i < n->end; i++) {
      isect_t its;
      if (prim_intersect(s, &s->bvh_prims[i], ray, &its, cnt)) {
        if (have) { if (its.t < out->t) *out = its; }
        else { *out = its; have = 1; }
      }
    }
    return have;
  }
  isect_t l, r;
  int hl = node_intersect(s, n->left, ray, &l, cnt);
  int hr = node_intersect(s, n->right, ray, &r, cnt);
  if (hl && hr) { *out = (l.t < r.t) ? l : r; return 1; }
  if (hl) { *out = l; return 1; }
  if (hr) { *out = r; return 1; }
  return 0;
}

/* Scene::find_intersection, scene.cpp:19 */
static int find_intersection(const orc_scene* s, const ray_t* ray, isect_t* out, orc_counters* cnt)
{
  cnt->queries++;
  out->hit = 0;
  if (s->n_nodes == 0) return 0;
  return node_intersect(s, 0, ray, out, cnt);
}

static void add_counters(orc_counters* dst, const orc_counters* src)
{
  dst->segments += src->segments; dst->queries += src->queries; dst->prim_tests += src->prim_tests;
  dst->box_tests += src->box_tests; dst->camera_rays += src->camera_rays;
}

void orc_intersect_batch(const orc_scene* s, const double* rays, int n, orc_hit* hits, orc_counters* counters)
{
  orc_counters total;
  memset(&total, 0, sizeof total);
#pragma omp parallel
  {
    orc_counters local;
    memset(&local, 0, sizeof local);
#pragma omp for schedule(static)
    for (int i = 0; i < n; i++) {
      ray_t r;
      r.origin = D3(rays[6 * (size_t)i + 0], rays[6 * (size_t)i + 1], rays[6 * (size_t)i + 2]);
      r.direction = D3(rays[6 * (size_t)i + 3], rays[6 * (size_t)i + 4], rays[6 * (size_t)i + 5]);
      isect_t its;
      memset(&its, 0, sizeof its);
      int h = find_intersection(s, &r, &its, &local);
      orc_hit* o = &hits[i];
      memset(o, 0, sizeof *o);
      o->hit = h;
      if (h) {
        o->id = its.id; o->t = its.t;
        o->point[0] = its.point.x; o->point[1] = its.point.y; o->point[2] = its.point.z;
        o->normal[0] = its.normal.x; o->normal[1] = its.normal.y; o->normal[2] = its.normal.z;
        o->uv[0] = its.uv[0]; o->uv[1] = its.uv[1];
        o->material = its.material; o->inside = its.inside;
      }
    }
#pragma omp critical
    add_counters(&total, &local);
  }
  if (counters) add_counters(counters, &total);
}

/* ------------------------------------------------------------------ */
/* images, background                                                  */
/* ------------------------------------------------------------------ */
typedef struct { uint8_t r, g, b; } u8x3;

/* Image::pixel, image.cpp:40-48.  INVALID_COLOR dvec3(1,0,1) becomes u8vec3(1,0,1). */
static u8x3 image_pixel(const image_t* im, int x, int y)
{
  u8x3 p = {1, 0, 1};
  if (!(im->data && 0 < im->w && 0 < im->h && 0 <= x && x < im->w && 0 <= y && y < im->h)) return p;
  size_t i = ((size_t)y * (size_t)im->w + (size_t)x) * 3;
  p.r = im->data[i]; p.g = im->data[i + 1]; p.b = im->data[i + 2];
  return p;
}
/* glm::mix(u8vec3, u8vec3, double): evaluated in double, converted back to u8 (truncation).  For a weight
 * outside [0,1] (texture coordinates below 0 give a negative fraction) the blend can leave 0..255; the
 * reference's x86-64 build then converts through a 32-bit integer and keeps the low byte. */
static u8x3 mix_u8(u8x3 a, u8x3 b, double t)
{
  u8x3 r;
  r.r = (uint8_t)(int32_t)((double)a.r * (1.0 - t) + (double)b.r * t);
  r.g = (uint8_t)(int32_t)((double)a.g * (1.0 - t) + (double)b.g * t);
  r.b = (uint8_t)(int32_t)((double)a.b * (1.0 - t) + (double)b.b * t);
  return r;
}
/* double -> int as the reference's x86-64 build does it (cvttsd2si): out of range and NaN give INT_MIN */
static int to_int(double x)
{
  if (!(x > -2147483649.0 && x < 2147483648.0)) return (-2147483647 - 1);
  return (int)x;
}
/* Image::sample, image.cpp:62-85 (interpolate = true) */
static d3 image_sample(const image_t* im, double u, double v)
{
  double x = u * (im->w - 1);
  double y = v * (im->h - 1);
  int x0 = to_int(x);
  int y0 = to_int(y);
  double x_frac = x - x0;
  double y_frac = y - y0;
  /* x0+1 on INT_MIN stays far out of range; avoid signed overflow at INT_MAX */
  int x1 = x0 == 2147483647 ? x0 : x0 + 1;
  int y1 = y0 == 2147483647 ? y0 : y0 + 1;
  u8x3 tl = image_pixel(im, x0, y0);
  u8x3 tr = image_pixel(im, x1, y0);
  u8x3 bl = image_pixel(im, x0, y1);
  u8x3 br = image_pixel(im, x1, y1);
  u8x3 p = mix_u8(mix_u8(tl, tr, x_frac), mix_u8(bl, br, x_frac), y_frac);
  return D3((double)p.r / 255.0, (double)p.g / 255.0, (double)p.b / 255.0); /* rgb(), util.h:51-61 */
}

/* util.h:142-149 */
static void equirectangular(d3 v, double uv[2])
{
  uv[0] = atan2(v.z, v.x) * 0.1591 + 0.5;
  uv[1] = asin(v.y) * 0.3183 + 0.5;
}

/* Scene::sample_background, scene.cpp:21-25,33-45 */
static d3 sample_background(const orc_scene* s, d3 dir)
{
  if (s->background_texture >= 0) {
    double uv[2];
    equirectangular(dir, uv);
    return vpow(image_sample(&s->textures[s->background_texture], uv[0], uv[1]), 2.2);
  }
  d3 c = s->background_color;
  if (c.x >= 0.0 && c.y >= 0.0 && c.z >= 0.0) return c;
  double a = 0.5 * (dir.y + 1.0);
  return add(muls(D3(1.0, 1.0, 1.0), 1.0 - a), muls(D3(0.5, 0.7, 1.0), a));
}

/* Intersection::albedo, geometry.cpp:18-26 */
static d3 surface_albedo(const orc_scene* s, const isect_t* its)
{
  const material_t* m = &s->materials[its->material];
  if (m->texture >= 0) return vpow(image_sample(&s->textures[m->texture], its->uv[0], its->uv[1]), 2.2);
  return m->albedo;
}

/* ------------------------------------------------------------------ */
/* sampling helpers, util.h:83-120                                     */
/* ------------------------------------------------------------------ */
static d3 spherical_to_cartesian(double theta, double phi)
{
  return D3(cos(phi) * sin(theta), cos(theta), sin(phi) * sin(theta));
}
/* random_unit_vector (util.h:91-96) with its two uniforms passed in */
static d3 unit_vector_from(double r0, double r1)
{
  double phi = 2.0 * ORC_PI * r0;
  double theta = acos(1.0 - 2.0 * r1);
  return spherical_to_cartesian(theta, phi);
}

/* ------------------------------------------------------------------ */
/* BxDF, material.cpp                                                  */
/* ------------------------------------------------------------------ */
static double reflectance(double cosine, double refraction_index)
{ /* material.cpp:100-106 */
  double r0 = (1 - refraction_index) / (1 + refraction_index);
  r0 = r0 * r0;
  return r0 + (1 - r0) * pow((1 - cosine), 5);
}
static double D_Beckmann(double NoH, double roughness)
{ /* :44-55 */
  double alpha = sq(roughness), alpha2 = sq(alpha);
  double NoH2 = sq(NoH), NoH4 = sq(NoH2);
  double sinTheta = sqrt(1.0 - NoH2);
  double tanTheta = sinTheta / NoH;
  return (1.0 / (alpha2 * NoH4)) * exp(-sq(tanTheta) / alpha2);
}
static double PDF_Beckmann(double NoH, double roughness)
{ /* :69-79 */
  double alpha = sq(roughness), alpha2 = sq(alpha);
  double sinTheta = sqrt(1.0 - sq(NoH));
  double tanTheta = sinTheta / NoH;
  double NoH3 = NoH * NoH * NoH;
  return (1.0 / (alpha2 * NoH3 * ORC_PI)) * exp(-sq(tanTheta) / alpha2);
}
static double G1_GGX_Schlick(double NoV, double roughness)
{ /* :81-86 */
  double alpha = roughness * roughness;
  double k = alpha / 2.0;
  return (NoV < 0.001 ? 0.001 : NoV) / (NoV * (1.0 - k) + k);
}

/* BxDF::sample, material.cpp:110-122, with the random numbers passed in:
 * r[0], r[1] feed the two-draw samplers, r[2] the dielectric branch. */
static d3 bxdf_sample_r(const orc_flags* fl, int type, double roughness, double refraction_index, int inside, d3 wo,
                        const double r[3])
{
  if (type == ORC_SPECULAR) { /* :157-162 */
    d3 fuzz = muls(unit_vector_from(r[0], r[1]), roughness);
    return add(reflect(neg(wo), D3(0.0, 1.0, 0.0)), fuzz);
  }
  if (type == ORC_MICROFACET && fl->importance_sample) { /* :57-67, :166-173 */
    double alpha = sq(roughness), alpha2 = sq(alpha);
    double phi = 2.0 * ORC_PI * r[0];
    double theta = atan(sqrt(-alpha2 * log(1.0 - (r[1] / ORC_PI))));
    d3 H = spherical_to_cartesian(theta, phi);
    return reflect(neg(wo), H);
  }
  if (type == ORC_DIELECTRIC) { /* :225-240 */
    double ri = inside ? (1.0 / refraction_index) : refraction_index;
    d3 N = D3(0, 1, 0);
    double cos_theta = wo.y < 1.0 ? wo.y : 1.0;
    double sin_theta = sqrt(1.0 - sq(cos_theta));
    if ((ri * sin_theta > 1.0) || (reflectance(cos_theta, ri) > r[2])) return reflect(neg(wo), N);
    return refract(neg(wo), N, ri);
  }
  /* diffuse (also MIRROR, which the reference's switch never dispatches), :138-144 */
  double phi = 2.0 * ORC_PI * r[0];
  double theta = acos(sqrt(r[1]));
  return spherical_to_cartesian(theta, phi);
}

/* draws random numbers in a fixed order: two for the direction samplers, one for the dielectric branch */
static d3 bxdf_sample(const orc_flags* fl, const material_t* m, int inside, d3 wo, rng_t* g)
{
  double r[3] = {0, 0, 0};
  if (m->type == ORC_DIELECTRIC) r[2] = random_double(g);
  else { r[0] = random_double(g); r[1] = random_double(g); }
  return bxdf_sample_r(fl, m->type, m->roughness, m->refraction_index, inside, wo, r);
}

/* BxDF::eval, material.cpp:124-136 */
static d3 bxdf_eval(const orc_flags* fl, int type, d3 albedo, double roughness, double metallic, d3 V, d3 L)
{
  if (type == ORC_DIELECTRIC) return D3(1, 1, 1); /* :242 */
  if (type == ORC_MICROFACET) { /* :175-215 */
    if (fl->check_hemisphere && (L.y < 0.0 || V.y < 0.0)) return D3(0, 0, 0);
    d3 H = normalize(add(V, L));
    double NoV = V.y, NoL = L.y, NoH = H.y;
    double VoH = fabs(dot(V, H));
    d3 f0 = add(muls(D3(0.04, 0.04, 0.04), 1.0 - metallic), muls(albedo, metallic)); /* glm::mix */
    double p5 = pow(1.0 - VoH, 5.0);
    d3 F = add(f0, muls(sub(D3(1, 1, 1), f0), p5));
    double G = G1_GGX_Schlick(NoV, roughness) * G1_GGX_Schlick(NoL, roughness);
    double D = D_Beckmann(NoH, roughness);
    d3 specular = divs(muls(muls(F, G), D), 4.0 * NoV * NoL);
    double pdf = fl->importance_sample ? PDF_Beckmann(NoH, roughness) : NoL / ORC_PI;
    return divs(muls(specular, NoL), pdf);
  }
  return albedo; /* diffuse :153, specular :164 */
}

/* ------------------------------------------------------------------ */
/* lights, geometry.cpp:228-254                                        */
/* ------------------------------------------------------------------ */
static d3 prim_sample_point(const prim_t* p, d3 point, rng_t* g)
{
  double r1 = random_double(g), r2 = random_double(g);
  if (p->type == PRIM_TRIANGLE) {
    double t = sqrt(r1);
    double u = 1.0 - t;
    double v = r2 * t;
    double w = 1.0 - u - v;
    return add(add(muls(p->v0, u), muls(p->v1, v)), muls(p->v2, w));
  }
  /* the reference draws r1,r2 and then two more inside random_unit_vector; the unused pair only
   * advances its generator, so the sphere case here reuses r1,r2 for the direction instead. */
  d3 normal = normalize(sub(point, p->center));
  d3 v = unit_vector_from(r1, r2); /* random_on_hemisphere, util.h:116-120 */
  if (!(dot(v, normal) > 0.0)) v = neg(v);
  return add(p->center, muls(v, p->radius));
}
static double prim_sample_area(const prim_t* p)
{
  if (p->type == PRIM_TRIANGLE) return 0.5 * length(cross(sub(p->v1, p->v0), sub(p->v2, p->v0)));
  return (4.0 * ORC_PI * sq(p->radius)) / 2.0;
}

/* ------------------------------------------------------------------ */
/* camera, camera.cpp                                                  */
/* ------------------------------------------------------------------ */
static d3 ld3(const double* p) { return D3(p[0], p[1], p[2]); }
static void sd3(double* p, d3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

static void camera_compute(orc_camera* c)
{ /* camera.cpp:78-82 */
  d3 right = cross(ld3(c->forward), ld3(c->world_up));
  d3 up = cross(right, ld3(c->forward));
  sd3(c->right, right); sd3(c->up, up);
}
void orc_camera_init(orc_camera* c, int width, int height, double fov_degrees, double aperture, double focus_distance)
{ /* camera.cpp:6-17 */
  c->width = width; c->height = height;
  sd3(c->world_up, D3(0, 1, 0)); sd3(c->forward, D3(0, 0, -1)); sd3(c->position, D3(0, 0, 0));
  c->fov = fov_degrees * 0.01745329251994329576923690768489; /* glm::radians */
  c->aperture = aperture; c->focus_distance = focus_distance;
  camera_compute(c);
}
void orc_camera_look_at(orc_camera* c, const double position[3], const double target[3])
{ /* camera.cpp:84-89 */
  sd3(c->position, ld3(position));
  sd3(c->forward, normalize(sub(ld3(target), ld3(position))));
  camera_compute(c);
}

/* camera.cpp:19-60 with the random inputs made explicit */
static ray_t camera_ray(const orc_camera* c, int x, int y, double rnd0, double rnd1, double disk_x, double disk_y)
{
  double jx = -0.5 + (rnd0 - 0.0) * (0.5 - -0.5) / (1.0 - 0.0); /* map_range, util.h:39-43 */
  double jy = -0.5 + (rnd1 - 0.0) * (0.5 - -0.5) / (1.0 - 0.0);
  double uvx = (((double)x + jx) / (double)c->width) * 2.0 - 1.0;
  double uvy = (((double)y + jy) / (double)c->height) * 2.0 - 1.0;
  double aspect_ratio = (double)c->width / (double)c->height;
  double half_height = tan(c->fov / 2.0);
  double half_width = half_height * aspect_ratio;
  double width = 2 * half_width;
  double height = 2 * half_height;
  d3 position = ld3(c->position), right = ld3(c->right), up = ld3(c->up);
  d3 target = add(position, ld3(c->forward));
  d3 view_point = sub(add(target, muls(muls(right, width), uvx)), muls(muls(up, height), uvy));
  d3 dir = normalize(sub(view_point, position));
  d3 focus_point = add(position, dir);
  d3 origin = position;
  if (c->aperture > 0 && c->focus_distance > 0) {
    double focus_dist = c->focus_distance;
    focus_point = add(position, muls(dir, focus_dist));
    double defocus_radius = focus_dist * tan((c->aperture / 2.0) * 0.01745329251994329576923690768489);
    d3 defocus_disk_u = muls(up, defocus_radius);
    d3 defocus_disk_v = muls(right, defocus_radius);
    origin = add(origin, add(muls(defocus_disk_v, disk_x), muls(defocus_disk_u, disk_y)));
  }
  ray_t r;
  r.origin = origin;
  r.direction = normalize(sub(focus_point, origin));
  return r;
}

void orc_camera_ray(const orc_camera* c, int x, int y, const double jitter_rnd[2], const double disk[2], double ray[6])
{
  ray_t r = camera_ray(c, x, y, jitter_rnd[0], jitter_rnd[1], disk[0], disk[1]);
  sd3(ray, r.origin); sd3(ray + 3, r.direction);
}

static ray_t camera_get_ray(const orc_camera* c, int x, int y, rng_t* g)
{
  double r0 = random_double(g), r1 = random_double(g);
  double dx = 0, dy = 0;
  if (c->aperture > 0 && c->focus_distance > 0) {
    for (;;) { /* random_in_unit_disk, util.h:98-106 */
      dx = -1.0 + (random_double(g) - 0.0) * (1.0 - -1.0) / (1.0 - 0.0);
      dy = -1.0 + (random_double(g) - 0.0) * (1.0 - -1.0) / (1.0 - 0.0);
      if (dx * dx + dy * dy < 1) break;
    }
  }
  return camera_ray(c, x, y, r0, r1, dx, dy);
}

/* ------------------------------------------------------------------ */
/* the integrator, renderer.cpp:57-171                                 */
/* ------------------------------------------------------------------ */
typedef struct {
  const orc_scene* scene;
  const orc_flags* flags;
  int max_bounce;
  orc_counters* cnt;
  rng_t* rng;
} trace_ctx;

static double luma(d3 c) { return dot(c, D3(0.2126, 0.7152, 0.0722)); } /* renderer.cpp:55 */

/* renderer.cpp:130-171 */
static d3 sample_lights(trace_ctx* cx, const isect_t* surf, d3 incoming)
{
  const orc_scene* s = cx->scene;
  if (s->n_lights == 0) return D3(0, 0, 0);
  /* Scene::random_light, scene.cpp:64-71: uniform over the light list */
  int idx = (int)(random_double(cx->rng) * (double)s->n_lights);
  if (idx > s->n_lights - 1) idx = s->n_lights - 1;
  const prim_t* light = &s->prims[s->lights[idx]];
  d3 result = D3(0, 0, 0);
  d3 point = surf->point;
  d3 point_to_light = sub(prim_sample_point(light, point, cx->rng), point);
  double distance = length(point_to_light);
  point_to_light = normalize(point_to_light);
  ray_t shadow;
  shadow.origin = point; shadow.direction = point_to_light;
  isect_t rec;
  int hit = find_intersection(s, &shadow, &rec, cx->cnt);
  if (hit && rec.id == light->id && surf->id != light->id) {
    d3 normal = rec.normal;
    m3f local2world = local_to_world(normal);
    m3f world2local = m3f_inverse(&local2world);
    d3 wo = to_d3(m3f_mul(&world2local, to_f3(neg(incoming))));
    d3 wi = to_d3(m3f_mul(&world2local, to_f3(point_to_light)));
    double area = prim_sample_area(light);
    double falloff = 1.0 / sq(distance);
    double light_cos_theta = dot(normal, neg(point_to_light));
    if (light_cos_theta < 0.0) light_cos_theta = 0.0;
    double weight = area * falloff * light_cos_theta;
    double light_pdf = 1.0 / (double)s->n_lights;
    d3 emission = s->materials[light->material].emission;
    const material_t* m = &s->materials[surf->material];
    d3 f = bxdf_eval(cx->flags, m->type, surface_albedo(s, surf), m->roughness, m->metallic, wo, wi);
    result = add(result, divs(mulv(muls(emission, weight), f), light_pdf));
  }
  return result;
}

/* renderer.cpp:57-126 */
static d3 trace_ray(trace_ctx* cx, const ray_t* ray, int depth, int perfect_reflection)
{
  if (cx->max_bounce <= depth) return D3(0, 0, 0);
  cx->cnt->segments++;
  const orc_scene* s = cx->scene;
  const orc_flags* fl = cx->flags;
  isect_t surface;
  if (!find_intersection(s, ray, &surface, cx->cnt)) return sample_background(s, ray->direction);
  const material_t* material = &s->materials[surface.material];

  if (fl->debug_normal) /* normal_as_color, renderer.cpp:17 */
    return muls(D3(surface.normal.x + 1, surface.normal.y + 1, surface.normal.z + 1), 0.5);

  double rr_weight = 1;
  if (fl->russian_roulette) {
    int min_depth = 3;
    if (min_depth < depth) {
      double rr_prob = luma(surface_albedo(s, &surface));
      if (random_double(cx->rng) >= rr_prob) return material->emission;
      rr_weight = 1.0 / rr_prob;
    }
  }

  m3f local2world = local_to_world(surface.normal);
  m3f world2local = m3f_inverse(&local2world);
  d3 wo = to_d3(m3f_mul(&world2local, to_f3(neg(ray->direction))));
  d3 wi = bxdf_sample(fl, material, surface.inside, wo, cx->rng);

  d3 radiance = D3(0, 0, 0);
  /* Material::is_perfectly_specular, material.h:30-35 */
  int perfectly_specular = (material->type & ORC_DIELECTRIC) || (material->type & ORC_MIRROR) ||
                           (((material->type & ORC_SPECULAR) || (material->type & ORC_MICROFACET)) &&
                            (material->roughness < 1e-5));

  if (!fl->direct_light_sampling || depth == 0 || perfectly_specular || perfect_reflection)
    radiance = add(radiance, material->emission);

  if (fl->direct_light_sampling && !perfectly_specular)
    radiance = add(radiance, sample_lights(cx, &surface, ray->direction));

  if (fl->indirect_light_sampling) {
    ray_t outgoing;
    outgoing.origin = surface.point;
    outgoing.direction = to_d3(m3f_mul(&local2world, to_f3(wi)));
    d3 li = trace_ray(cx, &outgoing, depth + 1, perfectly_specular);
    d3 f = bxdf_eval(fl, material->type, surface_albedo(s, &surface), material->roughness, material->metallic, wo, wi);
    radiance = add(radiance, mulv(li, f));
  }
  return muls(radiance, rr_weight);
}

/* Renderer::render, renderer.cpp:27-52 */
void orc_render_rows(const orc_scene* s, const orc_camera* cam, const orc_flags* flags, int samples, int max_bounce,
                     uint32_t seed, int total_samples, double* buffer, int y0, int y1, int threads,
                     orc_counters* counters)
{
  orc_counters total;
  memset(&total, 0, sizeof total);
  orc_flags fl = flags ? *flags : orc_default_flags();
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
#pragma omp parallel
  {
    orc_counters local;
    memset(&local, 0, sizeof local);
#pragma omp for schedule(dynamic, 1)
    for (int y = y0; y < y1; y++) {
      for (int x = 0; x < cam->width; x++) {
        int i = y * cam->width + x;
        d3 result = ld3(buffer + 3 * (size_t)i);
        for (int sidx = 0; sidx < samples; sidx++) {
          rng_t g = rng_init(seed, (uint32_t)i, (uint32_t)(total_samples + sidx));
          trace_ctx cx = {s, &fl, max_bounce, &local, &g};
          ray_t ray = camera_get_ray(cam, x, y, &g);
          local.camera_rays++;
          d3 color = trace_ray(&cx, &ray, 0, 0);
          double a = 1.0 / (double)(total_samples + sidx + 1);
          result = add(muls(result, 1.0 - a), muls(color, a)); /* glm::mix */
        }
        sd3(buffer + 3 * (size_t)i, result);
      }
    }
#pragma omp critical
    add_counters(&total, &local);
  }
  if (counters) add_counters(counters, &total);
}

void orc_render(const orc_scene* s, const orc_camera* cam, const orc_flags* flags, int samples, int max_bounce,
                uint32_t seed, int total_samples, double* buffer, int threads, orc_counters* counters)
{
  orc_render_rows(s, cam, flags, samples, max_bounce, seed, total_samples, buffer, 0, cam->height, threads, counters);
}

/* renderer.cpp:10-15,174-192 and util.h:131-134 */
void orc_tonemap_u8(const double* buffer, int n_pixels, uint8_t* out)
{
  const double a = 2.51, b = 0.03, c = 2.43, d = 0.59, e = 0.14;
  for (size_t i = 0; i < (size_t)n_pixels * 3; i++) {
    double x = buffer[i];
    x = (x * (a * x + b)) / (x * (c * x + d) + e);
    x = pow(x, 1.0 / 2.2);
    double cl = x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); /* glm::clamp = min(max(x,lo),hi); NaN -> 0 */
    if (x != x) cl = 0.0;
    out[i] = (uint8_t)(cl * 255.0);
  }
}

/* ------------------------------------------------------------------ */
/* known-answer-test entry points                                      */
/* ------------------------------------------------------------------ */
void orc_local_to_world(const double up[3], float m[9])
{
  m3f r = local_to_world(ld3(up));
  memcpy(m, r.c, sizeof(float) * 9);
}
void orc_mat3_inverse(const float m[9], float out[9])
{
  m3f a, r;
  memcpy(a.c, m, sizeof(float) * 9);
  r = m3f_inverse(&a);
  memcpy(out, r.c, sizeof(float) * 9);
}
void orc_equirectangular(const double v[3], double uv[2]) { equirectangular(ld3(v), uv); }
void orc_image_sample(const orc_scene* s, int texture, double u, double v, double rgb[3])
{
  sd3(rgb, image_sample(&s->textures[texture], u, v));
}
void orc_sample_background(const orc_scene* s, const double dir[3], double rgb[3])
{
  sd3(rgb, sample_background(s, ld3(dir)));
}
double orc_reflectance(double cosine, double refraction_index) { return reflectance(cosine, refraction_index); }
void orc_aces_tone_map(const double x[3], double out[3])
{
  const double a = 2.51, b = 0.03, c = 2.43, d = 0.59, e = 0.14;
  for (int i = 0; i < 3; i++) out[i] = (x[i] * (a * x[i] + b)) / (x[i] * (c * x[i] + d) + e);
}
int orc_ray_vs_triangle(const double ray[6], const double tri[9], double* t)
{
  ray_t r = {ld3(ray), ld3(ray + 3)};
  return ray_vs_triangle(&r, ld3(tri), ld3(tri + 3), ld3(tri + 6), 0.00001, t);
}
int orc_ray_vs_sphere(const double ray[6], const double center[3], double radius, double* t)
{
  ray_t r = {ld3(ray), ld3(ray + 3)};
  return ray_vs_sphere(&r, ld3(center), radius, 0.00001, 1e9, t);
}
int orc_ray_vs_aabb(const double ray[6], const double bmin[3], const double bmax[3], double tmin, double tmax)
{
  ray_t r = {ld3(ray), ld3(ray + 3)};
  return ray_vs_aabb(&r, ld3(bmin), ld3(bmax), tmin, tmax);
}
void orc_bxdf_eval(const orc_flags* fl, int type, const double albedo[3], double roughness, double metallic,
                   const double wo[3], const double wi[3], double out[3])
{
  orc_flags f = fl ? *fl : orc_default_flags();
  sd3(out, bxdf_eval(&f, type, ld3(albedo), roughness, metallic, ld3(wo), ld3(wi)));
}
void orc_bxdf_sample(const orc_flags* fl, int type, double roughness, double refraction_index, int inside,
                     const double wo[3], const double r[3], double wi[3])
{
  orc_flags f = fl ? *fl : orc_default_flags();
  sd3(wi, bxdf_sample_r(&f, type, roughness, refraction_index, inside, ld3(wo), r));
}
void orc_rng_draws(uint32_t seed, uint32_t pixel, uint32_t sample, int n, double* out)
{
  rng_t g = rng_init(seed, pixel, sample);
  for (int i = 0; i < n; i++) out[i] = random_double(&g);
}
