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

struct f3 { float x, y, z; };

template <int WIDTH>  // bytes per lane: 16, 8, 4; 12 (one dwordx3); 32 (two 16-byte loads of one 32-byte record); 28 (16 + 12 of a 28-byte record)
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
          acc += (v.x + v.y) + (v.z + v.w);
        } else if (WIDTH == 16) {
          f4 v = table[j];
          acc += (v.x + v.y) + (v.z + v.w);  // every component is used: the compiler must not narrow the load
        } else if (WIDTH == 32) {
          const f4* r = table + 2 * (j >> 1);
          f4 v = r[0], w = r[1];
          acc += ((v.x + v.y) + (v.z + v.w)) + ((w.x + w.y) + (w.z + w.w));
        } else if (WIDTH == 28) {
          const char* r = reinterpret_cast<const char*>(table) + 28u * (j >> 1);  // records of 28 bytes, 4-byte aligned
          f4 v = *reinterpret_cast<const f4*>(r);
          f3 w = *reinterpret_cast<const f3*>(r + 16);
          acc += ((v.x + v.y) + (v.z + v.w)) + ((w.x + w.y) + w.z);
        } else if (WIDTH == 12) {
          f3 v = *reinterpret_cast<const f3*>(&table[j]);
          acc += (v.x + v.y) + v.z;
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
    for (int width : {16, 8, 4, 12, 32, 28}) {
      if (lds && width != 16) continue;
      for (int mode = 0; mode < 7; mode++) {
        for (int rep = 0; rep < 2; rep++) {
          hipEventRecord(a);
          if (width == 16) hipLaunchKernelGGL(gather_kernel<16>, dim3(blocks), dim3(64), 0, 0, d, table_n - 1, iters, mode, out, lds);
          else if (width == 12) hipLaunchKernelGGL(gather_kernel<12>, dim3(blocks), dim3(64), 0, 0, d, table_n - 1, iters, mode, out, lds);
          else if (width == 32) hipLaunchKernelGGL(gather_kernel<32>, dim3(blocks), dim3(64), 0, 0, d, table_n - 1, iters, mode, out, lds);
          else if (width == 28) hipLaunchKernelGGL(gather_kernel<28>, dim3(blocks), dim3(64), 0, 0, d, table_n - 1, iters, mode, out, lds);
          else if (width == 8) hipLaunchKernelGGL(gather_kernel<8>, dim3(blocks), dim3(64), 0, 0, d, table_n - 1, iters, mode, out, lds);
          else hipLaunchKernelGGL(gather_kernel<4>, dim3(blocks), dim3(64), 0, 0, d, table_n - 1, iters, mode, out, lds);
          hipEventRecord(b);
          hipEventSynchronize(b);
        }
        float ms;
        hipEventElapsedTime(&ms, a, b);
        // wave-instructions per CU: 20 waves x iters x 8 (the 32- and 28-byte records: per record, two instructions)
        double per_cu = 20.0 * iters * 8.0;
        double cycles = ms * 1e-3 * 2.4e9 / per_cu;
        printf("%s %2d B/lane, %-16s: %8.3f ms, %6.1f cycles per wave-instruction per CU (at 2.4 GHz)\n", lds ? "LDS" : "L1 ", width,
               names[mode], ms, cycles);
      }
    }
  return 0;
}
