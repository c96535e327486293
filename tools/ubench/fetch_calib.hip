// Microbenchmark: what the L2's memory-side byte counters (rocprofv3 FETCH_SIZE, TCC_EA0_RDREQ*) report for access
// patterns of known size, so that the render kernel's FETCH_SIZE can be priced without guessing:
//   stream     consecutive lanes read consecutive 16-byte words of a 4 GiB buffer once (bytes known exactly);
//   line16     every lane reads 16 bytes at the start of a random 128-byte line (a 16-byte gather);
//   line2x16   ... and 16 more bytes from the line's other 64-byte half (same number of lines, twice the halves);
//   node32     every lane reads a random 32-byte-aligned record as two 16-byte loads (a tree node);
//   tri48      every lane reads 40 bytes of a random 48-byte record as 16 + 16 + 8 (a triangle; a record can
//              straddle two lines).
// The buffer is far larger than the L2 (4 MB per XCD) and the 256 MB Infinity Cache, so nearly every access misses.
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
// Run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_EA0_RDREQ_32B_sum -- ./fetch_calib     (then a second pass with
//        --pmc TCC_EA0_RDREQ_sum TCC_MISS_sum TCC_REQ_sum TCC_HIT_sum); the program prints the accesses per kernel.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ inline uint32_t mix(uint32_t x)
{
  x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15;
  return x;
}

constexpr int kIters = 64;

__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ buf, size_t n, float* out)
{
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 v = buf[i];
    s += v.x + v.w;
  }
  if (s == 12345.678f) out[0] = s;
}

// MODE 0 line16, 1 line2x16, 2 node32, 3 tri48
template <int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const char* __restrict__ buf, uint32_t records, float* out)
{
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  float s = 0.f;
  for (int it = 0; it < kIters; it++) {
    const uint32_t r = mix(t * 977u + (uint32_t)it * 0x9E3779B9u) % records;
    if (MODE == 0 || MODE == 1) {
      const char* p = buf + (size_t)r * 128;
      const float4 a = *reinterpret_cast<const float4*>(p);
      s += a.x;
      if (MODE == 1) { const float4 b = *reinterpret_cast<const float4*>(p + 64); s += b.y; }
    } else if (MODE == 2) {
      const char* p = buf + (size_t)r * 32;
      const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 16);
      s += a.x + b.w;
    } else {
      const char* p = buf + (size_t)r * 48;
      const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 16);
      const float2 c = *reinterpret_cast<const float2*>(p + 32);
      s += a.x + b.w + c.y;
    }
  }
  if (s == 12345.678f) out[0] = s;
}

int main()
{
  const size_t bytes = (size_t)4 << 30;
  char* buf; float* out;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 256) != hipSuccess) { fprintf(stderr, "allocation failed\n"); return 1; }
  if (hipMemset(buf, 0, bytes) != hipSuccess) return 1;
  const int blocks = 4096;
  const double accesses = (double)blocks * 256 * kIters;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto timed = [&](const char* name, double known_bytes, auto launch) {
    launch();  // warm-up (also the first of the two dispatches the profiler sees per pattern)
    (void)hipEventRecord(e0);
    launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s %8.3f ms  accesses %.0f  bytes asked for %.0f  (%.1f GB/s of asked-for bytes)\n", name, ms,
           name[0] == 's' ? 0.0 : accesses, known_bytes, known_bytes / (ms * 1e6));
  };
  timed("stream", (double)bytes, [&] { stream_kernel<<<8192, 256>>>(reinterpret_cast<const float4*>(buf), bytes / 16, out); });
  timed("line16", accesses * 16, [&] { gather_kernel<0><<<blocks, 256>>>(buf, (uint32_t)(bytes / 128), out); });
  timed("line2x16", accesses * 32, [&] { gather_kernel<1><<<blocks, 256>>>(buf, (uint32_t)(bytes / 128), out); });
  timed("node32", accesses * 32, [&] { gather_kernel<2><<<blocks, 256>>>(buf, (uint32_t)(bytes / 32), out); });
  timed("tri48", accesses * 40, [&] { gather_kernel<3><<<blocks, 256>>>(buf, (uint32_t)(bytes / 48), out); });
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  (void)hipFree(buf); (void)hipFree(out);
  return 0;
}
