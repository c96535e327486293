// Microbenchmark: how many cycles a SIMD needs per vector instruction of a wave, by instruction, by the waves
// resident on the SIMD and by how many lanes are active.  Every wave runs 16 independent dependency chains, so a
// single wave can keep its SIMD's pipe full if the hardware lets it.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip     Run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// mode: 0 all lanes, 1 lanes 0-31, 2 even lanes, 3 lanes 0-15
__device__ bool lane_on(int mode, int lane)
{
  switch (mode) {
    case 0: return true;
    case 1: return lane < 32;
    case 2: return (lane & 1) == 0;
    default: return lane < 16;
  }
}

constexpr int kChains = 16, kUnroll = 4, kOps = 33;
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(64) void valu_kernel(int iters, int mode, float seed, float* out, unsigned long long* clocks)
{
  const int lane = threadIdx.x;
  float a[kChains];
  for (int k = 0; k < kChains; k++) a[k] = seed + (float)(lane * kChains + k) * 1e-3f;
  const float m = 1.0f + seed * 1e-7f, c = seed * 1e-9f;
  unsigned long long mask = __ballot(lane & 1);
  if (lane_on(mode, lane)) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int u = 0; u < kUnroll; u++) {
#pragma unroll
        for (int k = 0; k < kChains; k++) {
          if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 3) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 4) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 5) asm volatile("v_mov_b32 %0, %1" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 6) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 7) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 8) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 9) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 10) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 11) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 12) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 13) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 14) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 15) asm volatile("v_mul_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 16) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 17) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 18) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 19) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 20) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 21) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 22) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 23) asm volatile("v_cndmask_b32 %0, %0, %1, %3" : "+v"(a[k]) : "v"(m), "v"(c), "s"(mask));
          if (OP == 24) asm volatile("v_cmp_le_f32 %1, %0, %2" : "+v"(a[k]), "=s"(mask) : "v"(m) : );
          if (OP == 25) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 26) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 27) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 30) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 31) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(a[k]) : "v"(m), "v"(c));
          if (OP == 32) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        }
        if (OP == 28 || OP == 29) {
#pragma unroll
          for (int k = 0; k < kChains; k += 2) {
            f2 v = {a[k], a[k + 1]}, mm = {m, m}, cc = {c, c};
            if (OP == 28) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(mm), "v"(cc));
            else asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(mm));
            a[k] = v.x; a[k + 1] = v.y;
          }
        }
      }
    }
  }
  float s = (float)(mask & 1);
  for (int k = 0; k < kChains; k++) s += a[k];
  out[blockIdx.x * 64 + lane] = s;
}

int main()
{
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
  const int cus = prop.multiProcessorCount;
  float* out; unsigned long long* clocks;
  const int max_waves = cus * 4 * 8;
  if (hipMalloc(&out, (size_t)max_waves * 64 * sizeof(float)) != hipSuccess) return 1;
  if (hipMalloc(&clocks, (size_t)max_waves * sizeof(unsigned long long)) != hipSuccess) return 1;
  const int iters = 8000;
  const char* op_names[kOps] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_fmac_f32", "v_mov_b32", "v_min_f32", "v_max_f32", "v_min3_f32", "v_med3_f32", "v_cvt_f32_u32", "v_cvt_f32_u32_sdwa", "v_cvt_f32_ubyte1", "v_cvt_f32_f16", "v_fma_mix_f32", "v_mul_f32_sdwa", "v_perm_b32", "v_and_b32", "v_and_or_b32", "v_lshlrev_b32", "v_add_u32", "v_mul_lo_u32", "v_mad_u32_u24", "v_cndmask_b32 (sgpr mask)", "v_cmp_le_f32 (to sgpr)", "v_rcp_f32", "v_sqrt_f32", "v_ldexp_f32", "v_pk_fma_f32 (2 fma)", "v_pk_mul_f32 (2 mul)", "v_pk_fma_f16", "v_pk_min_f16", "v_dot2c_f32_f16"};
  const char* mode_names[] = {"64 lanes", "lanes 0-31", "even lanes", "lanes 0-15"};
  for (int op = 0; op < kOps; op++)
    for (int mode = 0; mode < (op < 1 ? 4 : 1); mode++)
      for (int waves_per_simd : {1, 2, 4, 8}) {
        if (op >= 1 && waves_per_simd != 4) continue;
        const int n = cus * 4 * waves_per_simd;  // one wave per workgroup; the dispatcher spreads them over the SIMDs
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
          hipEventRecord(e0);
          switch (op) {
            case 0: valu_kernel<0><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 1: valu_kernel<1><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 2: valu_kernel<2><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 3: valu_kernel<3><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 4: valu_kernel<4><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 5: valu_kernel<5><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 6: valu_kernel<6><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 7: valu_kernel<7><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 8: valu_kernel<8><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 9: valu_kernel<9><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 10: valu_kernel<10><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 11: valu_kernel<11><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 12: valu_kernel<12><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 13: valu_kernel<13><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 14: valu_kernel<14><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 15: valu_kernel<15><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 16: valu_kernel<16><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 17: valu_kernel<17><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 18: valu_kernel<18><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 19: valu_kernel<19><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 20: valu_kernel<20><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 21: valu_kernel<21><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 22: valu_kernel<22><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 23: valu_kernel<23><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 24: valu_kernel<24><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 25: valu_kernel<25><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 26: valu_kernel<26><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 27: valu_kernel<27><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 28: valu_kernel<28><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 29: valu_kernel<29><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 30: valu_kernel<30><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 31: valu_kernel<31><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
            case 32: valu_kernel<32><<<n, 64>>>(iters, mode, 1.0f, out, clocks); break;
          }
          hipEventRecord(e1);
          hipEventSynchronize(e1);
        }
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const bool packed = op == 28 || op == 29;
        const double instr_per_wave = (double)iters * kChains * kUnroll / (packed ? 2 : 1);
        const double cycles_at_2p4 = ms * 1e-3 * 2.4e9;
        printf("%-26s %-10s %d waves/SIMD: %8.3f ms, %6.2f cycles per wave-instruction per SIMD at 2.4 GHz\n",
               op_names[op], mode_names[mode], waves_per_simd, ms, cycles_at_2p4 / (instr_per_wave * waves_per_simd));
        hipEventDestroy(e0); hipEventDestroy(e1);
      }
  (void)hipFree(out); (void)hipFree(clocks);
  return 0;
}
