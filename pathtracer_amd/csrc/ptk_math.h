// Small float vector helpers shared by the HIP kernels and by host code that
// prepares their inputs.  Single precision throughout: the kernels trade the
// reference's double arithmetic (glm::dvec3 everywhere) for fp32, which keeps
// hit distances within 1e-5 relative of the reference's.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PTK_HD __host__ __device__ __forceinline__
#else
#define PTK_HD inline
#endif

// Analysis aid: -DPTK_MARKERS puts named comments into the compiler's assembly output (they constrain its scheduling a
// little, so normal builds leave them out); tools/asm_regions.py counts the instructions between them.
#if defined(PTK_MARKERS) && defined(__HIP_DEVICE_COMPILE__)
#define PTK_MARK(name) asm volatile("; PTK_MARK " name)
#else
#define PTK_MARK(name)
#endif

namespace ptk {

struct v3 {
  float x, y, z;
};

struct alignas(16) f4 {
  float x, y, z, w;
};

PTK_HD v3 V3(float x, float y, float z) { return v3{x, y, z}; }
PTK_HD v3 operator+(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PTK_HD v3 operator-(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PTK_HD v3 operator*(v3 a, v3 b) { return v3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PTK_HD v3 operator*(v3 a, float s) { return v3{a.x * s, a.y * s, a.z * s}; }
PTK_HD v3 operator*(float s, v3 a) { return v3{a.x * s, a.y * s, a.z * s}; }
PTK_HD v3 operator-(v3 a) { return v3{-a.x, -a.y, -a.z}; }
PTK_HD v3& operator+=(v3& a, v3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
PTK_HD v3& operator*=(v3& a, v3 b) { a.x *= b.x; a.y *= b.y; a.z *= b.z; return a; }
PTK_HD v3& operator*=(v3& a, float s) { a.x *= s; a.y *= s; a.z *= s; return a; }
PTK_HD float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PTK_HD v3 cross(v3 a, v3 b) { return v3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
PTK_HD float sqf(float x) { return x * x; }

// 1/x and 1/sqrt(x): on the GPU the hardware's 1-ulp reciprocal instructions, one instruction each.
PTK_HD float frcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}
PTK_HD float frsq(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rsqf(x);
#else
  return 1.0f / sqrtf(x);
#endif
}
PTK_HD float length(v3 a) { return sqrtf(dot(a, a)); }
PTK_HD v3 normalize(v3 a) { return a * frsq(dot(a, a)); }

// x^y for x >= 0 and (sin, cos) of an angle given in turns.  On the GPU these map to the hardware's
// log2 / exp2 / sin / cos instructions (about 1e-6 absolute error), a few instructions instead of
// the hundreds of the correctly rounded library routines; they only feed shading arithmetic
// (texture gamma, sampled directions), never an intersection test.
PTK_HD float fast_pow(float x, float y)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
#else
  return powf(x, y);
#endif
}
PTK_HD void sincos_turns(float turns, float* s, float* c)
{
#if defined(__HIP_DEVICE_COMPILE__)
  *s = __builtin_amdgcn_sinf(turns);
  *c = __builtin_amdgcn_cosf(turns);
#else
  *s = sinf(turns * 6.28318530718f);
  *c = cosf(turns * 6.28318530718f);
#endif
}

// One 16-byte fetch of a record's chunk.  On the device it is written as a vector load, which the compiler
// keeps as ONE instruction (a struct copy is taken apart into single words and only sometimes put together
// again; each separate load costs the memory pipeline what a 16-byte one does).
PTK_HD f4 load16(const void* p)
{
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float vf4 __attribute__((ext_vector_type(4)));
  const vf4 v = *reinterpret_cast<const vf4*>(p);
  return f4{v.x, v.y, v.z, v.w};
#else
  return *reinterpret_cast<const f4*>(p);
#endif
}

PTK_HD float as_float(uint32_t u)
{
  union { uint32_t u; float f; } c;
  c.u = u;
  return c.f;
}
PTK_HD uint32_t as_uint(float f)
{
  union { uint32_t u; float f; } c;
  c.f = f;
  return c.u;
}

}  // namespace ptk
