// Microbenchmark: what a wave-level 16-byte gather costs on the L1 path as a function of how many lanes are
// active (and which), with every CU's waves issuing such loads back to back from a small, L1-resident table.
// Build: hipcc --offload-arch=gfx950 -O3 -o l1_lanes l1_lanes.hip     Run: ./l1_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct alignas(16) f4 { float x, y, z, w; };

// mode: 0 all lanes, 1 lanes 0-31, 2 even lanes, 3 lanes 0-15, 4 every 4th lane, 5 lanes 0-7, 6 one lane per quad of 16
__device__ bool lane_on(int mode, int lane)
{
  switch (mode) {
    case 0: return true;
    case 1: return lane < 32;
    case 2: return (lane & 1) == 0;
    case 3: return lane < 16;
    case 4: return (lane & 3) == 0;
    case 5: return lane < 8;
    default: return (lane & 15) == 0;
  }
}

template <int WIDTH>  // bytes per lane: 16, 8, 4
__global__ __launch_bounds__(64) void gather_kernel(const f4* table, int table_mask, int iters, int mode, float* out, int use_lds)
{
  __shared__ f4 lds[256];  // 4 KB
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) lds[i] = table[i];
  __syncthreads();
  unsigned idx = (lane * 2654435761u + blockIdx.x * 40503u) & table_mask;
  float acc = 0.f;
  if (lane_on(mode, lane)) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {  // eight independent loads in flight, then a dependent index update
        unsigned j = (idx + k * 97u) & table_mask;
        if (use_lds) {
          f4 v = lds[j & 255];
          acc += v.x + v.w;
        } else if (WIDTH == 16) {
          f4 v = table[j];
          acc += v.x + v.w;
        } else if (WIDTH == 8) {
          float2 v = *reinterpret_cast<const float2*>(&table[j]);
          acc += v.x + v.y;
        } else {
          acc += table[j].x;
        }
      }
      idx = (idx * 1664525u + 1013904223u + (unsigned)(acc != 123.f)) & table_mask;
    }
  }
  out[blockIdx.x * 64 + lane] = acc;
}

int main()
{
  const int table_n = 512;  // 8 KB: stays in every CU's L1
  std::vector<f4> h(table_n);
  for (int i = 0; i < table_n; i++) h[i] = f4{(float)i, 1.f, 2.f, 3.f};
  f4* d;
  float* out;
  const int blocks = 256 * 20;
  hipMalloc(&d, table_n * sizeof(f4));
  hipMalloc(&out, blocks * 64 * sizeof(float));
  hipMemcpy(d, h.data(), table_n * sizeof(f4), hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 2000;
  const char* names[] = {"64 lanes", "lanes 0-31", "even lanes", "lanes 0-15", "every 4th lane", "lanes 0-7", "every 16th lane"};
  for (int lds = 0; lds < 2; lds++)
    for (int width : {16, 8, 4}) {
      if (lds && width != 16) continue;
      for (int mode = 0; mode < 7; mode++) {
        for (int rep = 0; rep < 2; rep++) {
          hipEventRecord(a);
          if (width == 16) hipLaunchKernelGGL(gather_kernel<16>, dim3(blocks), dim3(64), 0, 0, d, table_n - 1, iters, mode, out, lds);
          else if (width == 8) hipLaunchKernelGGL(gather_kernel<8>, dim3(blocks), dim3(64), 0, 0, d, table_n - 1, iters, mode, out, lds);
          else hipLaunchKernelGGL(gather_kernel<4>, dim3(blocks), dim3(64), 0, 0, d, table_n - 1, iters, mode, out, lds);
          hipEventRecord(b);
          hipEventSynchronize(b);
        }
        float ms;
        hipEventElapsedTime(&ms, a, b);
        // wave-instructions per CU: 20 waves x iters x 8
        double per_cu = 20.0 * iters * 8.0;
        double cycles = ms * 1e-3 * 2.4e9 / per_cu;
        printf("%s %2d B/lane, %-16s: %8.3f ms, %6.1f cycles per wave-instruction per CU (at 2.4 GHz)\n", lds ? "LDS" : "L1 ", width,
               names[mode], ms, cycles);
      }
    }
  return 0;
}
