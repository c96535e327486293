/*
 * ptk.h -- C ABI of the MI355X path-tracing kernels (libptk.so).
 *
 * The reference has no plugin or FFI layer: its render path is a C++ class
 * API called from src/main.cpp.  This header is the boundary that API is
 * re-hosted on.  Every entry point names the reference interface it stands
 * in for.  All arguments are plain pointers, sizes and PODs; `int` results
 * are 0 on success and negative on failure, with a message available from
 * ptk_last_error().  Nothing throws or exits across this boundary (the
 * reference calls exit(1) / assert: src/scene.cpp:107, src/scene.cpp:51).
 *
 * Interfaces of the reference that have NO entry point here, because the reference
 * only ever calls them from inside its render loop and that loop now runs on the GPU
 * as a whole: Camera::get_ray (src/camera.h:15), Scene::sample_background and
 * Scene::random_light (src/scene.h:24-27), Image::sample / Image::pixel (src/image.h),
 * BxDF::sample / eval / pdf (src/material.h:38-65), Renderer::trace_ray /
 * sample_lights (src/renderer.h), Primitive::intersect / sample_point
 * (src/geometry.h), BVH and AABB as classes.  Their work is done by device functions
 * of the render kernel; ptk_intersect_batch is the one query a host caller can pose
 * by itself (Scene::find_intersection).  The C++ mirror above this header
 * (pathtracer_amd/host/) leaves the same methods out.
 *
 * A context is bound to one GPU and is not thread-safe; use one context per
 * device (one process per GPU in multi-GPU runs).  Host input arrays are
 * copied during the call; device memory belongs to the context until
 * ptk_destroy().
 */
#ifndef PTK_H
#define PTK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTK_OK 0
#define PTK_ERR_INVALID (-1) /* bad argument or call order */
#define PTK_ERR_HIP (-2)     /* a HIP runtime call failed */
#define PTK_ERR_LIMIT (-3)   /* a documented limit was exceeded */

typedef struct ptk_ctx ptk_ctx;

/* Material::Type, src/material.h:12-18 */
enum {
  PTK_DIFFUSE = 1 << 0,
  PTK_SPECULAR = 1 << 1,
  PTK_DIELECTRIC = 1 << 2,
  PTK_MIRROR = 1 << 3,
  PTK_MICROFACET = 1 << 4
};

/* Material, src/material.h:11-36 */
typedef struct {
  int32_t type;
  int32_t texture; /* index into ptk_scene_desc.textures, -1 = none (Material::texture) */
  double albedo[3];
  double emission[3];
  double refraction_index;
  double roughness;
  double metallic;
} ptk_material;

/* Image as loaded by Image::load, src/image.cpp:25-31: 8-bit RGB, rows bottom-up */
typedef struct {
  const uint8_t* rgb;
  int32_t width, height;
} ptk_texture;

/* What Scene holds after load_obj / add_primitive(s) / add_material / set_background_*
 * (src/scene.h:17-32).  Primitive ids are: triangles 0..n_triangles-1 in array order,
 * then spheres (src/scene.cpp:56-62 numbering with models before spheres, src/main.cpp:117-145). */
typedef struct {
  int32_t n_triangles;
  const double* tri_pos;        /* n*9: v0 v1 v2              (Triangle, src/geometry.h:38-41) */
  const double* tri_nrm;        /* n*9: n0 n1 n2 */
  const double* tri_uv;         /* n*6: t0 t1 t2 */
  const int32_t* tri_material;  /* n */
  int32_t n_spheres;
  const double* sphere_center;  /* n*3                         (Sphere, src/geometry.h:28-32) */
  const double* sphere_radius;  /* n */
  const int32_t* sphere_material;
  int32_t n_materials;
  const ptk_material* materials;
  int32_t n_textures;
  const ptk_texture* textures;
  int32_t background_texture;   /* Scene::set_background_texture; -1 = none */
  double background_color[3];   /* Scene::set_background_color; any component < 0 = sky gradient */
} ptk_scene_desc;

/* Camera state after construction + look_at, src/camera.h:9-26 */
typedef struct {
  int32_t width, height;
  double position[3];
  double forward[3];
  double right[3];        /* cross(forward, world_up), src/camera.cpp:80 */
  double up[3];           /* cross(right, forward),    src/camera.cpp:81 */
  double fov;             /* radians */
  double aperture;        /* defocus angle, degrees */
  double focus_distance;
} ptk_camera;

/* The compile-time switches of src/config.h:3-8, as run-time values */
typedef struct {
  int32_t debug_normal;            /* PT_DEBUG_NORMAL            0 */
  int32_t russian_roulette;        /* PT_RUSSIAN_ROULETTE        1 */
  int32_t direct_light_sampling;   /* PT_DIRECT_LIGHT_SAMPLING   1 */
  int32_t indirect_light_sampling; /* PT_INDIRECT_LIGHT_SAMPLING 1 */
  int32_t importance_sample;       /* PT_IMPORTANCE_SAMPLE       1 */
  int32_t check_hemisphere;        /* PT_CHECK_HEMISPHERE        1 */
} ptk_flags;

/* One Renderer::render call (src/renderer.cpp:27-52) restricted to a share of the frame.
 * Sample indices are global: this call renders indices sample_first + k*sample_stride,
 * k = 0..samples-1, of every pixel in rows [row_begin, row_end).  Splitting a frame by
 * sample index (stride = number of GPUs) or by rows gives the same sample set as one
 * call on one GPU.  Rows can also be shared in interleaved bands: with band_rows > 0 the call
 * renders, of the bands of band_rows rows that the image divides into (band b = rows
 * [b*band_rows, (b+1)*band_rows)), the bands band_first, band_first + band_stride, ... in ONE
 * launch; row_begin / row_end must then be 0. */
typedef struct {
  int32_t samples;
  int32_t max_bounce;     /* Renderer::m_max_bounce */
  uint32_t seed;
  int32_t sample_first;
  int32_t sample_stride;  /* >= 1 */
  int32_t row_begin;      /* 0 */
  int32_t row_end;        /* <= 0 means height */
  int32_t collect_stats;  /* 1: also count box / primitive tests (slower kernel variant; at most 2^20 samples per call) */
  int32_t band_rows;      /* 0: rows [row_begin, row_end); else a multiple of 8 */
  int32_t band_first;     /* >= 0 */
  int32_t band_stride;    /* >= 1 when band_rows > 0 */
} ptk_render_params;

/* Ray, src/ray.h:5-9 */
typedef struct { double origin[3]; double direction[3]; } ptk_ray;

/* Intersection, src/geometry.h:14-24 (material pointer -> index; hit = has_value) */
typedef struct {
  int32_t hit;
  uint32_t id;
  float t;
  float point[3];
  float normal[3];
  float uv[2];
  int32_t material;
  int32_t inside;
} ptk_hit;

/* The reference's two global counters (bounce_counter src/renderer.cpp:8,63 and
 * intersection_test_counter src/geometry.cpp:15,162) plus what the kernels add. */
typedef struct {
  uint64_t camera_rays;
  uint64_t segments;       /* trace_ray bodies = bounce_counter */
  uint64_t queries;        /* find_intersection calls: segments + shadow rays */
  uint64_t box_tests;      /* only with collect_stats */
  uint64_t prim_tests;     /* only with collect_stats = intersection_test_counter */
  double kernel_ms_last;   /* HIP-event time of the last render / intersect launch */
  double kernel_ms_total;  /* sum over launches since ptk_reset_stats */
  uint64_t launches;
} ptk_stats;

/* Scene layout facts after upload (for roofline accounting and tests) */
typedef struct {
  int32_t n_primitives, n_lights, n_nodes, max_depth, max_leaf_size;
  int32_t traversal_kind;  /* which form of the render kernel the scene gets: 0 22-entry LDS stack, 1 27-entry LDS stack,
                              2 whole scene in LDS, 3 22 entries + overflow words, 4 19 entries + material and light
                              tables in LDS, 5 as 4 with overflow words */
  uint64_t node_bytes, prim_bytes, shading_bytes, texture_bytes;
  int32_t builder;         /* PTK_BUILDER_* that built the tree in use */
  int32_t tree_from_cache; /* 1: the host tree was read from the directory PTK_BVH_CACHE names */
  double upload_ms;        /* host wall time of the ptk_upload_scene call (flattening, tree, copies) */
  double device_build_ms;  /* device time of the tree build; 0 for a host-built tree */
} ptk_scene_info;

/* -- lifetime ---------------------------------------------------------------- */
int ptk_create(int device, ptk_ctx** out);
int ptk_destroy(ptk_ctx* ctx);
/* message of the last failing call on ctx (or of the last failing ptk_create when ctx is NULL) */
const char* ptk_last_error(const ptk_ctx* ctx);

/* -- scene: stands in for Scene::compute_bvh (src/scene.cpp:80), which in the reference
 * is the point after which the scene is immutable.  Builds the acceleration structure on
 * the host and uploads everything. */
int ptk_upload_scene(ptk_ctx* ctx, const ptk_scene_desc* scene);
/* Same with the tree builder named (BVH::construct, src/bvh.cpp:54-83, is the reference's only one):
 *   PTK_BUILDER_HOST_SAH     binned surface-area-heuristic build on the host's cores: the tree the
 *                            render rates are quoted on; about a second per million triangles.
 *   PTK_BUILDER_DEVICE_LBVH  linear BVH built on the GPU in milliseconds, for scenes that are loaded or
 *                            changed often; rays visit more nodes in it.  Scenes of fewer than 64
 *                            primitives are built on the host either way.
 * Which tree is used never changes the closest hit of a ray (apart from exact ties in distance).
 * ptk_upload_scene takes the builder from the environment variable PTK_BUILDER (host | device),
 * host when unset.  With PTK_BVH_CACHE naming a directory, host-built trees are kept there, keyed by a
 * hash of the geometry, and read back (after a full validity check) instead of being rebuilt. */
#define PTK_BUILDER_HOST_SAH 0
#define PTK_BUILDER_DEVICE_LBVH 1
int ptk_upload_scene_ex(ptk_ctx* ctx, const ptk_scene_desc* scene, int32_t builder);
int ptk_scene_info_get(const ptk_ctx* ctx, ptk_scene_info* out);

/* -- camera / renderer state: Renderer::Renderer (src/renderer.cpp:19-25) allocates the
 * W*H accumulation buffer for a camera; ptk_set_camera does that and clears it. */
int ptk_set_camera(ptk_ctx* ctx, const ptk_camera* camera);
int ptk_set_flags(ptk_ctx* ctx, const ptk_flags* flags);
/* Use caller-owned device memory (width*height*4 floats: sum r,g,b + sample count) as the
 * accumulation buffer, e.g. a tensor that a collective library reduces across GPUs.
 * NULL returns to the context's own buffer.  The buffer is not cleared. */
int ptk_bind_accum(ptk_ctx* ctx, float* device_rgbw);
int ptk_reset_accum(ptk_ctx* ctx);

/* -- Renderer::render (src/renderer.cpp:27-52): launches the path kernel; asynchronous.
 * The same sequence of calls gives the same bits every time; how a frame is split into calls
 * (rows, bands, sample batches) only changes the fp32 order in which a pixel's samples are summed.
 * May allocate device memory the first time an image size is rendered. */
int ptk_render(ptk_ctx* ctx, const ptk_render_params* params);
/* -- ptk_render for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments, same
 * launches and bit-identical sums, but the per-chunk partial sums of the frame live in a static 128 MiB
 * device array instead of a buffer grown on demand, and no timing events are recorded (ptk_stats'
 * kernel_ms_* do not include these launches).  Calls no hipMalloc / hipFree.  A call whose partial sums
 * (16 bytes x pixels rendered x sample chunks, at most 32) exceed the array fails with PTK_ERR_LIMIT and a
 * message that says so: render fewer rows or samples per call.  All calls on one GPU share that GPU's array:
 * calls of different contexts that overlap in time on one GPU race; calls on one context are ordered by its stream. */
int ptk_render_static(ptk_ctx* ctx, const ptk_render_params* params);
/* -- the buffer Renderer::save_image reads (src/renderer.cpp:186): per-pixel mean radiance,
 * row 0 = top, W*H*3 floats, host memory.  Blocks until rendering has finished.
 * total_samples (may be NULL) receives the largest per-pixel sample count. */
int ptk_read_accum(ptk_ctx* ctx, float* rgb, int32_t* total_samples);
/* -- the per-pixel work of Renderer::save_image (src/renderer.cpp:174-192: mean, ACES curve, gamma
 * 1/2.2, clamp, * 255 truncated to a byte) done on the device: W*H*3 bytes, row 0 = top, host
 * memory.  The same bytes that the reference's loop gives from ptk_read_accum's floats.  Blocks. */
int ptk_read_display_u8(ptk_ctx* ctx, uint8_t* rgb8, int32_t* total_samples);
int ptk_synchronize(ptk_ctx* ctx);

/* -- Scene::find_intersection (src/scene.cpp:19) over a batch of rays.
 * Host arrays; blocks.  The device staging buffers are kept between calls: a batch no larger than an
 * earlier one (a single ray, say) allocates nothing. */
int ptk_intersect_batch(ptk_ctx* ctx, const ptk_ray* rays, int32_t n, ptk_hit* hits);
/* Same with device-resident inputs/outputs: rays as n*6 floats (origin, direction; the array 8-byte
 * aligned); asynchronous. */
int ptk_intersect_batch_device(ptk_ctx* ctx, const float* device_rays, int32_t n, ptk_hit* device_hits);

/* -- counters: print_stats (src/geometry.cpp:273-278) */
int ptk_get_stats(ptk_ctx* ctx, ptk_stats* out); /* blocks */
int ptk_reset_stats(ptk_ctx* ctx);

/* The HIP stream (hipStream_t) the context launches on, for callers that order their own
 * work after it. */
void* ptk_stream(ptk_ctx* ctx);

/* -- several GPUs behind one handle.  The reference's only parallel construct is the OpenMP loop over image
 * rows inside Renderer::render (src/renderer.cpp:31-32); its caller (src/main.cpp:236-255) is one thread that
 * calls render() in a loop and save_image() between batches.  A group keeps that calling pattern and spreads
 * each render call over the GPUs of a node from the one calling thread: one context and one stream per device,
 * no exchange while rendering, and ONE collective (an RCCL sum-reduce of the per-device (sum r, g, b, count)
 * buffers onto device 0, over xGMI) each time the frame is read.  The per-device buffers keep accumulating
 * across render calls (Renderer::total_samples semantics), whatever the split.
 *   PTK_SPLIT_SAMPLES  every device renders every pixel; device i takes the call's samples i, i+n, i+2n, ...
 *                      (global sample indices: the union is exactly what one GPU would render)
 *   PTK_SPLIT_BANDS    device i renders all samples of the 16-row bands i, i+n, i+2n, ... of the image
 * With one device a group is a thin wrapper of that device's context.  `devices` names each GPU once.
 * (Naming ONE device several times is allowed for testing the group logic on a one-GPU machine: those
 * contexts share the GPU and their buffers are added by a kernel instead of a collective.)
 * ptk_group_context gives the context of the index-th device, e.g. for ptk_get_stats, ptk_scene_info_get or
 * ptk_intersect_batch; do not render, bind or change the scene through it.  librccl.so is loaded when the
 * first group of two or more GPUs is created. */
typedef struct ptk_group ptk_group;
#define PTK_SPLIT_SAMPLES 0
#define PTK_SPLIT_BANDS 1
int ptk_group_create(const int* devices, int n_devices, ptk_group** out);
int ptk_group_destroy(ptk_group* group);
/* message of the last failing call on group (or of the last failing ptk_group_create when group is NULL) */
const char* ptk_group_last_error(const ptk_group* group);
int ptk_group_size(const ptk_group* group);
ptk_ctx* ptk_group_context(ptk_group* group, int index);
/* Scene::compute_bvh for every device: the scene is flattened (and a host tree built) once and copied to each
 * GPU; builder as in ptk_upload_scene_ex. */
int ptk_group_upload_scene(ptk_group* group, const ptk_scene_desc* scene, int32_t builder);
int ptk_group_set_camera(ptk_group* group, const ptk_camera* camera);
int ptk_group_set_flags(ptk_group* group, const ptk_flags* flags);
int ptk_group_reset_accum(ptk_group* group);
/* Renderer::render over all devices; asynchronous.  params describes the whole call (samples = the call's total);
 * with the band split leave row_begin, row_end and band_* zero. */
int ptk_group_render(ptk_group* group, const ptk_render_params* params, int32_t split);
int ptk_group_synchronize(ptk_group* group);
/* print_stats (src/geometry.cpp:273-278) reports whole-run totals: the counters of all devices added up (kernel
 * times: the slowest device's, since the devices work side by side).  Blocks. */
int ptk_group_get_stats(ptk_group* group, ptk_stats* out);
int ptk_group_reset_stats(ptk_group* group);
/* ptk_read_accum / ptk_read_display_u8 of the whole frame: reduce onto device 0, then read.  Block. */
int ptk_group_read_accum(ptk_group* group, float* rgb, int32_t* total_samples);
int ptk_group_read_display_u8(ptk_group* group, uint8_t* rgb8, int32_t* total_samples);

#ifdef __cplusplus
}
#endif
#endif
