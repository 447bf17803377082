// Shared device/host helpers for the far3d_hip C-ABI library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <atomic>

#define FAR3D_OK 0
#define FAR3D_ERR_ARG (-1)      // bad argument (null pointer, unsupported size/dtype)
#define FAR3D_ERR_LAUNCH (-2)   // hipLaunch / runtime error
#define FAR3D_ERR_UNSUPPORTED (-3)

#define FAR3D_DT_F32 0
#define FAR3D_DT_BF16 1
#define FAR3D_DT_F32_BF16X3 2   // far3d_conv2d_nhwc weight code: fp32 data, two-term bf16 split products (include/far3d_hip.h)
#define FAR3D_DT_BF16_PAIR 3    // activation storage: fp32 values kept as [32 hi | 32 lo] bf16 per 32-channel block (include/far3d_hip.h)
#define FAR3D_SUMS_MAX_PARTS 32
#define FAR3D_SUMS_FRAC_BITS 18      // fixed-point channel sums of far3d_conv2d_nhwc (include/far3d_hip.h)

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

// Dynamic LDS above 64 KiB has to be allowed per kernel AND per device (hipFuncSetAttribute acts on the current device): `mask` is the
// launcher's function-local set of devices already done (one bit per device ordinal; a duplicate call from a racing thread is harmless).
// A failure is an error of the call, not something to launch through.
static inline int far3d_allow_lds(const void* fn, int bytes, std::atomic<unsigned long long>& mask, const char* who) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const unsigned long long bit = 1ull << (dev & 63);
  if (mask.load(std::memory_order_relaxed) & bit) return FAR3D_OK;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    far3d_set_error("%s: hipFuncSetAttribute(max dynamic LDS = %d bytes) failed on device %d: %s", who, bytes, dev, hipGetErrorString(e));
    return FAR3D_ERR_LAUNCH;
  }
  mask.fetch_or(bit, std::memory_order_relaxed);
  return FAR3D_OK;
}

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

// ---- "pair" storage (FAR3D_DT_BF16_PAIR): a tensor of C logical channels (C % 32 == 0) is stored as 2C bf16 per pixel; every
// 32-channel block is 32 hi = bf16(x) followed by 32 lo = bf16(x - hi).  hi + lo is exact in fp32 (both are pieces of x's own
// mantissa), so the format carries 16 significant bits at fp32's byte size and its 64-byte hi / lo runs are what the LDS-DMA
// conv kernels stream.  Pointers / strides of pair tensors count bf16 elements; chan_off maps a logical channel to its hi.
struct pair_t { uint16_t v; };
template <typename T> __device__ __forceinline__ int chan_off(int c) { return c; }
template <> __device__ __forceinline__ int chan_off<pair_t>(int c) { return ((c >> 5) << 6) | (c & 31); }
template <typename T> struct ChanScale { static constexpr int v = 1; };
template <> struct ChanScale<pair_t> { static constexpr int v = 2; };      // stored elements per logical channel
__device__ __forceinline__ float4 load4(const pair_t* p) {   // p -> hi of 4 consecutive channels (same 32-block); lo 32 elements on
  const uint2 h = *reinterpret_cast<const uint2*>(p);
  const uint2 l = *reinterpret_cast<const uint2*>(p + 32);
  float4 o;
  o.x = __uint_as_float(h.x << 16) + __uint_as_float(l.x << 16);
  o.y = __uint_as_float(h.x & 0xffff0000u) + __uint_as_float(l.x & 0xffff0000u);
  o.z = __uint_as_float(h.y << 16) + __uint_as_float(l.y << 16);
  o.w = __uint_as_float(h.y & 0xffff0000u) + __uint_as_float(l.y & 0xffff0000u);
  return o;
}
// 4 floats -> (4 hi bf16, 4 lo bf16)
__device__ __forceinline__ void split4f(float x0, float x1, float x2, float x3, uint2& h, uint2& l) {
  h.x = pack_bf16x2(x0, x1); h.y = pack_bf16x2(x2, x3);
  l.x = pack_bf16x2(x0 - __uint_as_float(h.x << 16), x1 - __uint_as_float(h.x & 0xffff0000u));
  l.y = pack_bf16x2(x2 - __uint_as_float(h.y << 16), x3 - __uint_as_float(h.y & 0xffff0000u));
}
__device__ __forceinline__ void store4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, const float4& v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
}
__device__ __forceinline__ void store4(pair_t* p, const float4& v) {
  uint2 h, l;
  split4f(v.x, v.y, v.z, v.w, h, l);
  *reinterpret_cast<uint2*>(p) = h;
  *reinterpret_cast<uint2*>(p + 32) = l;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
