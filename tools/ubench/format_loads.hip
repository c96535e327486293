// Microbenchmark: typed buffer loads (buffer_load_format_xyzw with a 16_16_16_16 USCALED descriptor: the memory
// pipeline converts four 16-bit integers to four floats) against plain 8- and 16-byte gathers, all from a small
// L1-resident table, by how many lanes are active.  Also checks that the conversion is exact.
// Build: hipcc --offload-arch=gfx950 -O3 -o format_loads format_loads.hip     Run: ./format_loads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ v4f buffer_load_format_xyzw(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4f32");
__device__ v2f buffer_load_b64(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ v4f buffer_load_b128(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");

// Buffer resource: base address, no stride, `bytes` addressable, destination select x,y,z,w, and the element format
__device__ v4i make_rsrc(const void* base, uint32_t bytes, uint32_t num_format, uint32_t data_format)
{
  const uint64_t a = (uint64_t)base;
  v4i r;
  r.x = (int)(uint32_t)a;
  r.y = (int)(uint32_t)((a >> 32) & 0xffffu);
  r.z = (int)bytes;
  r.w = (int)(4u | (5u << 3) | (6u << 6) | (7u << 9) | (num_format << 12) | (data_format << 15));
  return r;
}
constexpr uint32_t kNumUscaled = 2, kNumFloat = 7, kData16x4 = 12, kData32 = 4;

__device__ bool lane_on(int mode, int lane)
{
  switch (mode) {
    case 0: return true;
    case 1: return lane < 32;
    case 2: return (lane & 1) == 0;
    case 3: return lane < 16;
    default: return (lane & 3) == 0;
  }
}

// KIND 0: format load of 8 bytes -> 4 floats; 1: plain 8-byte buffer load; 2: plain 16-byte buffer load; 3, 4: global 8- and
// 16-byte loads; 5: a 32-byte record as three format loads and one 8-byte load (four instructions per count)
template <int KIND>
__global__ __launch_bounds__(64) void gather_kernel(const uint16_t* table, int table_mask, int iters, int mode, float* out)
{
  const int lane = threadIdx.x;
  const v4i typed = make_rsrc(table, (uint32_t)(table_mask + 1) * 8u, kNumUscaled, kData16x4);
  const v4i plain = make_rsrc(table, (uint32_t)(table_mask + 1) * 8u, kNumFloat, kData32);
  unsigned idx = (lane * 2654435761u + blockIdx.x * 40503u) & table_mask;
  float acc = 0.f;
  if (lane_on(mode, lane)) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {  // eight independent loads in flight, then a dependent index update
        const unsigned j = (idx + k * 97u) & table_mask;  // entries of 8 bytes
        if (KIND == 0) {
          v4f v = buffer_load_format_xyzw(typed, (int)(j * 8u), 0, 0);
          acc += v.x + v.w;
        } else if (KIND == 1) {
          v2f v = buffer_load_b64(plain, (int)(j * 8u), 0, 0);
          acc += v.x + v.y;
        } else if (KIND == 2) {
          v4f v = buffer_load_b128(plain, (int)((j & ~1u) * 8u), 0, 0);
          acc += v.x + v.w;
        } else if (KIND == 3) {
          v2f v = *reinterpret_cast<const v2f*>(table + j * 4u);
          acc += v.x + v.y;
        } else if (KIND == 4) {
          v4f v = *reinterpret_cast<const v4f*>(table + (j & ~1u) * 4u);
          acc += v.x + v.w;
        } else {  // three typed loads and one plain 8-byte load of one 32-byte record: a tree node
          const int o = (int)((j & ~3u) * 8u);
          v4f x = buffer_load_format_xyzw(typed, o, 0, 0), y = buffer_load_format_xyzw(typed, o + 8, 0, 0), z = buffer_load_format_xyzw(typed, o + 16, 0, 0);
          v2f l = buffer_load_b64(plain, o + 24, 0, 0);
          acc += x.x + y.y + z.z + l.x + x.w + y.w + z.w;
        }
      }
      idx = (idx * 1664525u + 1013904223u + (unsigned)(acc != 123.f)) & table_mask;
    }
  }
  out[blockIdx.x * 64 + lane] = acc;
}

__global__ void check_kernel(const uint16_t* table, int n, int* bad)
{
  const v4i typed = make_rsrc(table, (uint32_t)n * 8u, kNumUscaled, kData16x4);
  for (int j = threadIdx.x + blockIdx.x * blockDim.x; j < n; j += blockDim.x * gridDim.x) {
    v4f v = buffer_load_format_xyzw(typed, j * 8, 0, 0);
    const uint16_t* p = table + j * 4;
    if (v.x != (float)p[0] || v.y != (float)p[1] || v.z != (float)p[2] || v.w != (float)p[3]) atomicAdd(bad, 1);
  }
}

int main()
{
  // exactness: every 16-bit value, in every component
  {
    const int n = 65536;
    std::vector<uint16_t> h((size_t)n * 4);
    for (int i = 0; i < n; i++) { h[4 * i] = (uint16_t)i; h[4 * i + 1] = (uint16_t)(65535 - i); h[4 * i + 2] = (uint16_t)(i * 7); h[4 * i + 3] = (uint16_t)(i ^ 0x5555); }
    uint16_t* d; int* bad; int hbad = -1;
    if (hipMalloc(&d, h.size() * 2) != hipSuccess || hipMalloc(&bad, 4) != hipSuccess) return 1;
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(check_kernel, dim3(64), dim3(256), 0, 0, d, n, bad);
    hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
    printf("format load of 16_16_16_16 USCALED: %d of %d entries differ from (float)u16\n", hbad, n);
    (void)hipFree(d); (void)hipFree(bad);
  }
  const int table_n = 1024;  // 8 KB: stays in every CU's L1
  std::vector<uint16_t> h((size_t)table_n * 4);
  for (size_t i = 0; i < h.size(); i++) h[i] = (uint16_t)(i * 13u);
  uint16_t* d;
  float* out;
  const int blocks = 256 * 20;
  if (hipMalloc(&d, h.size() * 2) != hipSuccess || hipMalloc(&out, blocks * 64 * sizeof(float)) != hipSuccess) return 1;
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 2000;
  const char* names[] = {"64 lanes", "lanes 0-31", "even lanes", "lanes 0-15", "every 4th lane"};
  const char* kinds[] = {"format 8 B -> 4 floats", "buffer  8 B", "buffer 16 B", "global  8 B", "global 16 B", "node: 3 format + 8 B"};
  for (int kind = 0; kind < 6; kind++)
    for (int mode = 0; mode < 5; mode++) {
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        if (kind == 0) hipLaunchKernelGGL(gather_kernel<0>, dim3(blocks), dim3(64), 0, 0, d, table_n - 1, iters, mode, out);
        else if (kind == 1) hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(64), 0, 0, d, table_n - 1, iters, mode, out);
        else if (kind == 2) hipLaunchKernelGGL(gather_kernel<2>, dim3(blocks), dim3(64), 0, 0, d, table_n - 1, iters, mode, out);
        else if (kind == 3) hipLaunchKernelGGL(gather_kernel<3>, dim3(blocks), dim3(64), 0, 0, d, table_n - 1, iters, mode, out);
        else if (kind == 4) hipLaunchKernelGGL(gather_kernel<4>, dim3(blocks), dim3(64), 0, 0, d, table_n - 1, iters, mode, out);
        else hipLaunchKernelGGL(gather_kernel<5>, dim3(blocks), dim3(64), 0, 0, d, table_n - 1, iters, mode, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
      }
      float ms;
      hipEventElapsedTime(&ms, a, b);
      const double per_cu = 20.0 * iters * 8.0;  // wave-instructions per CU: 20 waves x iters x 8
      printf("%-24s %-16s: %8.3f ms, %6.1f cycles per wave-instruction per CU (at 2.4 GHz)\n", kinds[kind], names[mode], ms,
             ms * 1e-3 * 2.4e9 / per_cu);
    }
  return 0;
}
