// Shared device/host helpers for the far3d_hip C-ABI library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define FAR3D_OK 0
#define FAR3D_ERR_ARG (-1)      // bad argument (null pointer, unsupported size/dtype)
#define FAR3D_ERR_LAUNCH (-2)   // hipLaunch / runtime error
#define FAR3D_ERR_UNSUPPORTED (-3)

#define FAR3D_DT_F32 0
#define FAR3D_DT_BF16 1
#define FAR3D_DT_F32_BF16X3 2   // far3d_conv2d_nhwc weight code: fp32 data, two-term bf16 split products (include/far3d_hip.h)
#define FAR3D_SUMS_MAX_PARTS 32

void far3d_set_error(const char* fmt, ...);

#define FAR3D_CHECK_ARG(cond, ...)            \
  do {                                        \
    if (!(cond)) {                            \
      far3d_set_error(__VA_ARGS__);           \
      return FAR3D_ERR_ARG;                   \
    }                                         \
  } while (0)

#define FAR3D_CHECK_LAUNCH(name)                                            \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      far3d_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return FAR3D_ERR_LAUNCH;                                              \
    }                                                                       \
  } while (0)

typedef uint16_t bf16_t;  // raw bfloat16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
// round-to-nearest-even, NaN preserved (same rounding torch uses for .to(bfloat16))
// gfx950 converts in hardware (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __bf16 hw_bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float hw_f32x2_t __attribute__((ext_vector_type(2)));
  const hw_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2_t));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

template <typename T> struct LoadCvt;
template <> struct LoadCvt<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
};
template <> struct LoadCvt<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
};

// 4 consecutive channels -> float4
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const bf16_t* p) {
  uint2 r = *reinterpret_cast<const uint2*>(p);
  float4 o;
  o.x = __uint_as_float(r.x << 16);
  o.y = __uint_as_float(r.x & 0xffff0000u);
  o.z = __uint_as_float(r.y << 16);
  o.w = __uint_as_float(r.y & 0xffff0000u);
  return o;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
