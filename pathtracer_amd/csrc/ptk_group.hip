// Several GPUs behind one handle (ptk_group_* in include/ptk.h): the reference's one parallel loop
// (#pragma omp parallel for over image rows, src/renderer.cpp:31) spread over the GPUs of a node from ONE host
// thread, so that the reference's own caller (src/main.cpp:236-255: Renderer::render in a loop, save_image
// between batches) can use all of them without knowing.
//
// A group owns one context (one stream) per device.  A render call is cut into one launch per device -- by
// sample index (every device renders every pixel, interleaved GLOBAL sample indices) or by image rows
// (interleaved 16-row bands) -- with no exchange while rendering.  Reading the frame is the one exchange step:
// the devices' (sum r, sum g, sum b, count) buffers are summed onto device 0 by one RCCL reduce over xGMI
// (ncclCommInitAll: one communicator per device in this one process), and device 0 forms the means or the
// display bytes.  The per-device buffers are left as they are, so rendering can go on accumulating.
//
// RCCL is loaded when the first group of more than one distinct device is created (dlopen), so that a process
// that renders on one GPU never maps it.  A device may be named more than once in a group: those contexts
// share the GPU, and their buffers are added by a kernel instead of a collective.  That arrangement exists so
// that the whole group logic can be exercised on a one-GPU machine; it is not a way to render faster.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "ptk_internal.h"

namespace {

constexpr int kBandRows = 16;
constexpr int kMaxGroup = 64;

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclCommInitAll) comm_init_all = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclReduce) reduce = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  std::string why;  // why loading failed
};

Rccl& rccl()
{
  static Rccl r;
  if (r.lib || !r.why.empty()) return r;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
    r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (r.lib) break;
  }
  if (!r.lib) { r.why = std::string("cannot load librccl.so: ") + dlerror(); return r; }
  auto sym = [&](const char* n) -> void* {
    void* p = dlsym(r.lib, n);
    if (!p && r.why.empty()) r.why = std::string("librccl.so lacks ") + n;
    return p;
  };
  r.comm_init_all = (decltype(r.comm_init_all))sym("ncclCommInitAll");
  r.comm_destroy = (decltype(r.comm_destroy))sym("ncclCommDestroy");
  r.group_start = (decltype(r.group_start))sym("ncclGroupStart");
  r.group_end = (decltype(r.group_end))sym("ncclGroupEnd");
  r.reduce = (decltype(r.reduce))sym("ncclReduce");
  r.error_string = (decltype(r.error_string))sym("ncclGetErrorString");
  if (!r.why.empty()) { dlclose(r.lib); r.lib = nullptr; }
  return r;
}

// out[i] = a[i] (first == 1) or out[i] += a[i]: the merge of contexts that share one GPU, in context order
__global__ __launch_bounds__(256) void add_accum_kernel(float4* __restrict__ out, const float4* __restrict__ a, size_t n, int first)
{
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float4 v = a[i];
  if (!first) { float4 o = out[i]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
  out[i] = v;
}

}  // namespace

struct ptk_group {
  std::vector<ptk_ctx*> ctx;
  std::vector<int> device;
  std::vector<ncclComm_t> comm;  // empty: one context, or contexts that share a GPU
  bool shared_device = false;
  float* d_merged = nullptr;     // on device[0]: the sum of all contexts' buffers, formed by the read calls
  size_t merged_pixels = 0;
  std::string err;
};

static std::string g_group_create_error;

static int gfail(ptk_group* g, int code, const std::string& msg)
{
  if (g) g->err = msg; else g_group_create_error = msg;
  return code;
}

// Runs `call` on every context; stops at the first failure and takes over its message.
template <class F>
static int for_each_context(ptk_group* g, const char* what, F call)
{
  for (size_t i = 0; i < g->ctx.size(); i++) {
    int rc = call(g->ctx[i], (int)i);
    if (rc != PTK_OK) return gfail(g, rc, std::string(what) + " (device " + std::to_string(g->device[i]) + "): " + ptk_last_error(g->ctx[i]));
  }
  return PTK_OK;
}

extern "C" {

int ptk_group_create(const int* devices, int n_devices, ptk_group** out)
{
  if (!out) return gfail(nullptr, PTK_ERR_INVALID, "ptk_group_create: out is NULL");
  *out = nullptr;
  if (!devices || n_devices < 1 || n_devices > kMaxGroup) return gfail(nullptr, PTK_ERR_INVALID, "ptk_group_create: 1 to 64 devices");
  ptk_group* g = new (std::nothrow) ptk_group();
  if (!g) return gfail(nullptr, PTK_ERR_INVALID, "ptk_group_create: out of host memory");
  g->device.assign(devices, devices + n_devices);
  std::vector<int> sorted = g->device;
  std::sort(sorted.begin(), sorted.end());
  const bool distinct = std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end();
  if (!distinct && sorted.front() != sorted.back()) {
    delete g;
    return gfail(nullptr, PTK_ERR_INVALID, "ptk_group_create: name every device once (or one device several times, for tests)");
  }
  g->shared_device = !distinct;
  for (int i = 0; i < n_devices; i++) {
    ptk_ctx* c = nullptr;
    int rc = ptk_create(devices[i], &c);
    if (rc != PTK_OK) {
      std::string why = ptk_last_error(nullptr);
      ptk_group_destroy(g);
      return gfail(nullptr, rc, why);
    }
    g->ctx.push_back(c);
  }
  if (n_devices > 1 && distinct) {
    Rccl& r = rccl();
    if (!r.lib) { std::string why = r.why; ptk_group_destroy(g); return gfail(nullptr, PTK_ERR_HIP, "ptk_group_create: " + why); }
    g->comm.assign((size_t)n_devices, nullptr);
    ncclResult_t nr = r.comm_init_all(g->comm.data(), n_devices, g->device.data());
    if (nr != ncclSuccess) {
      g->comm.clear();
      ptk_group_destroy(g);
      return gfail(nullptr, PTK_ERR_HIP, std::string("ptk_group_create: ncclCommInitAll: ") + r.error_string(nr));
    }
  }
  *out = g;
  return PTK_OK;
}

int ptk_group_destroy(ptk_group* g)
{
  if (!g) return PTK_OK;
  for (ptk_ctx* c : g->ctx) if (c) (void)ptk_synchronize(c);
  for (ncclComm_t c : g->comm) if (c) (void)rccl().comm_destroy(c);
  if (g->d_merged) { (void)hipSetDevice(g->device[0]); (void)hipFree(g->d_merged); }
  for (ptk_ctx* c : g->ctx) (void)ptk_destroy(c);
  delete g;
  return PTK_OK;
}

const char* ptk_group_last_error(const ptk_group* g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }

int ptk_group_size(const ptk_group* g) { return g ? (int)g->ctx.size() : 0; }

ptk_ctx* ptk_group_context(ptk_group* g, int index)
{
  if (!g || index < 0 || index >= (int)g->ctx.size()) return nullptr;
  return g->ctx[(size_t)index];
}

int ptk_group_upload_scene(ptk_group* g, const ptk_scene_desc* scene, int32_t builder)
{
  if (!g) return PTK_ERR_INVALID;
  // the scene is flattened (and a host tree built) once, by the first context, and handed to the others
  int rc = ptk::ctx_upload_keep_host(g->ctx[0], scene, builder);
  if (rc != PTK_OK) return gfail(g, rc, std::string("ptk_group_upload_scene: ") + ptk_last_error(g->ctx[0]));
  for (size_t i = 1; i < g->ctx.size() && rc == PTK_OK; i++) {
    rc = ptk::ctx_upload_like(g->ctx[i], g->ctx[0]);
    if (rc != PTK_OK) gfail(g, rc, "ptk_group_upload_scene (device " + std::to_string(g->device[i]) + "): " + ptk_last_error(g->ctx[i]));
  }
  ptk::ctx_release_host_copy(g->ctx[0]);
  // all devices or none: a group with the scene on some of its devices only would render part of a frame
  if (rc != PTK_OK)
    for (ptk_ctx* c : g->ctx) ptk::ctx_drop_scene(c);
  return rc;
}

int ptk_group_set_camera(ptk_group* g, const ptk_camera* camera)
{
  if (!g) return PTK_ERR_INVALID;
  return for_each_context(g, "ptk_group_set_camera", [&](ptk_ctx* c, int) { return ptk_set_camera(c, camera); });
}

int ptk_group_set_flags(ptk_group* g, const ptk_flags* flags)
{
  if (!g) return PTK_ERR_INVALID;
  return for_each_context(g, "ptk_group_set_flags", [&](ptk_ctx* c, int) { return ptk_set_flags(c, flags); });
}

int ptk_group_reset_accum(ptk_group* g)
{
  if (!g) return PTK_ERR_INVALID;
  return for_each_context(g, "ptk_group_reset_accum", [&](ptk_ctx* c, int) { return ptk_reset_accum(c); });
}

int ptk_group_render(ptk_group* g, const ptk_render_params* p, int32_t split)
{
  if (!g) return PTK_ERR_INVALID;
  if (!p) return gfail(g, PTK_ERR_INVALID, "ptk_group_render: params is NULL");
  if (split != PTK_SPLIT_SAMPLES && split != PTK_SPLIT_BANDS) return gfail(g, PTK_ERR_INVALID, "ptk_group_render: unknown split");
  const int n = (int)g->ctx.size();
  if (n == 1) return for_each_context(g, "ptk_group_render", [&](ptk_ctx* c, int) { return ptk_render(c, p); });
  if (split == PTK_SPLIT_BANDS && (p->band_rows != 0 || p->row_begin != 0 || p->row_end > 0))
    return gfail(g, PTK_ERR_INVALID, "ptk_group_render: the band split shares the whole image: leave row_begin, row_end and band_* 0");
  if (p->sample_stride < 1) return gfail(g, PTK_ERR_INVALID, "ptk_group_render: sample_stride must be >= 1");
  return for_each_context(g, "ptk_group_render", [&](ptk_ctx* c, int i) {
    ptk_render_params q = *p;
    if (split == PTK_SPLIT_SAMPLES) {
      // device i takes the call's samples i, i + n, i + 2n, ...: global indices first + (i + k n) * stride
      q.samples = p->samples > i ? (p->samples - i + n - 1) / n : 0;
      q.sample_first = p->sample_first + i * p->sample_stride;
      q.sample_stride = p->sample_stride * n;
    } else {
      q.band_rows = kBandRows;
      q.band_first = i;
      q.band_stride = n;
    }
    return ptk_render(c, &q);
  });
}

int ptk_group_synchronize(ptk_group* g)
{
  if (!g) return PTK_ERR_INVALID;
  return for_each_context(g, "ptk_group_synchronize", [&](ptk_ctx* c, int) { return ptk_synchronize(c); });
}

int ptk_group_get_stats(ptk_group* g, ptk_stats* out)
{
  if (!g) return PTK_ERR_INVALID;
  if (!out) return gfail(g, PTK_ERR_INVALID, "ptk_group_get_stats: out is NULL");
  ptk_stats sum;
  std::memset(&sum, 0, sizeof sum);
  int rc = for_each_context(g, "ptk_group_get_stats", [&](ptk_ctx* c, int) {
    ptk_stats st;
    int r = ptk_get_stats(c, &st);
    if (r != PTK_OK) return r;
    sum.camera_rays += st.camera_rays; sum.segments += st.segments; sum.queries += st.queries;
    sum.box_tests += st.box_tests; sum.prim_tests += st.prim_tests;
    sum.launches += st.launches;
    // the devices work side by side: the slowest one's kernel time is the group's
    sum.kernel_ms_last = std::max(sum.kernel_ms_last, st.kernel_ms_last);
    sum.kernel_ms_total = std::max(sum.kernel_ms_total, st.kernel_ms_total);
    return (int)PTK_OK;
  });
  if (rc == PTK_OK) *out = sum;
  return rc;
}

int ptk_group_reset_stats(ptk_group* g)
{
  if (!g) return PTK_ERR_INVALID;
  return for_each_context(g, "ptk_group_reset_stats", [&](ptk_ctx* c, int) { return ptk_reset_stats(c); });
}

}  // extern "C"

// Sum of all contexts' buffers in g->d_merged (device 0).
static int merge_frame(ptk_group* g)
{
  const size_t pixels = ptk::ctx_accum_pixels(g->ctx[0]);
  if (pixels == 0) return gfail(g, PTK_ERR_INVALID, "ptk_group: set the camera first");
  for (ptk_ctx* c : g->ctx)
    if (ptk::ctx_accum_pixels(c) != pixels) return gfail(g, PTK_ERR_INVALID, "ptk_group: the contexts' image sizes differ (use ptk_group_set_camera)");
  int rc = ptk_group_synchronize(g);
  if (rc != PTK_OK) return rc;
  hipError_t e = hipSetDevice(g->device[0]);
  if (e == hipSuccess && g->merged_pixels != pixels) {
    if (g->d_merged) (void)hipFree(g->d_merged);
    g->d_merged = nullptr;
    g->merged_pixels = 0;
    e = hipMalloc((void**)&g->d_merged, pixels * 4 * sizeof(float));
    if (e == hipSuccess) g->merged_pixels = pixels;
  }
  if (e != hipSuccess) return gfail(g, PTK_ERR_HIP, std::string("ptk_group: ") + hipGetErrorString(e));
  if (!g->comm.empty()) {
    Rccl& r = rccl();
    ncclResult_t nr = r.group_start();
    for (size_t i = 0; i < g->ctx.size() && nr == ncclSuccess; i++)
      nr = r.reduce(ptk::ctx_accum(g->ctx[i]), g->d_merged, pixels * 4, ncclFloat, ncclSum, 0, g->comm[i], (hipStream_t)ptk_stream(g->ctx[i]));
    ncclResult_t ne = r.group_end();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) {
      // whatever part of the collective was queued must have drained before the caller may touch the buffers again
      for (ptk_ctx* c : g->ctx) (void)ptk_synchronize(c);
      return gfail(g, PTK_ERR_HIP, std::string("ptk_group: ncclReduce: ") + r.error_string(nr));
    }
    return ptk_group_synchronize(g);
  }
  hipStream_t s0 = (hipStream_t)ptk_stream(g->ctx[0]);
  for (size_t i = 0; i < g->ctx.size(); i++)
    hipLaunchKernelGGL(add_accum_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, s0, (float4*)g->d_merged,
                       (const float4*)ptk::ctx_accum(g->ctx[i]), pixels, i == 0 ? 1 : 0);
  e = hipGetLastError();
  if (e != hipSuccess) return gfail(g, PTK_ERR_HIP, std::string("ptk_group: ") + hipGetErrorString(e));
  return PTK_OK;
}

// Reads the merged frame through context 0's own read path, with the merged buffer standing in for its own.
template <class F>
static int read_merged(ptk_group* g, const char* what, F read)
{
  ptk_ctx* c0 = g->ctx[0];
  if (g->ctx.size() == 1) {
    int rc = read(c0);
    return rc == PTK_OK ? rc : gfail(g, rc, std::string(what) + ": " + ptk_last_error(c0));
  }
  int rc = merge_frame(g);
  if (rc != PTK_OK) return rc;
  float* own = ptk::ctx_accum(c0);
  rc = ptk_bind_accum(c0, g->d_merged);
  if (rc == PTK_OK) rc = read(c0);
  int rb = ptk_bind_accum(c0, own);
  if (rc == PTK_OK) rc = rb;
  return rc == PTK_OK ? rc : gfail(g, rc, std::string(what) + ": " + ptk_last_error(c0));
}

extern "C" {

int ptk_group_read_accum(ptk_group* g, float* rgb, int32_t* total_samples)
{
  if (!g) return PTK_ERR_INVALID;
  return read_merged(g, "ptk_group_read_accum", [&](ptk_ctx* c) { return ptk_read_accum(c, rgb, total_samples); });
}

int ptk_group_read_display_u8(ptk_group* g, uint8_t* rgb8, int32_t* total_samples)
{
  if (!g) return PTK_ERR_INVALID;
  return read_merged(g, "ptk_group_read_display_u8", [&](ptk_ctx* c) { return ptk_read_display_u8(c, rgb8, total_samples); });
}

}  // extern "C"
