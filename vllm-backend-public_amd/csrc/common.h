// Shared device helpers and host-side argument checking for libmi355_decode.so.
// gfx950 only: wave = 64 lanes, MFMA 16x16x32 for bf16 / fp16.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "mi355_decode.h"

namespace mi355 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- per-dtype traits: raw 16-bit word <-> float, MFMA on 8-element fragments
struct BF16 {
  static __device__ __forceinline__ float to_f(uint16_t u) {
    return __uint_as_float(static_cast<uint32_t>(u) << 16);
  }
  static __device__ __forceinline__ uint16_t from_f(float f) {
    __bf16 b = static_cast<__bf16>(f);  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(uint16_t, b);
  }
  // two floats -> two packed 16-bit values, one conversion instruction (RNE, as from_f)
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    const b2 r = __builtin_convertvector(f2{a, b}, b2);
    return __builtin_bit_cast(uint32_t, r);
  }
  static __device__ __forceinline__ f32x4 mfma(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  // two fp8 (e4m3fn) of the low / high half of `src` -> two packed 16-bit values (exact)
  static __device__ __forceinline__ uint32_t fp8x2(uint32_t src, bool high_half) {
    typedef __attribute__((ext_vector_type(2))) __bf16 v2;
    const v2 r = high_half ? __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(src, 1.0f, true)
                           : __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(src, 1.0f, false);
    return __builtin_bit_cast(uint32_t, r);
  }
};

struct F16 {
  static __device__ __forceinline__ float to_f(uint16_t u) {
    return static_cast<float>(__builtin_bit_cast(_Float16, u));
  }
  static __device__ __forceinline__ uint16_t from_f(float f) {
    _Float16 h = static_cast<_Float16>(f);
    return __builtin_bit_cast(uint16_t, h);
  }
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const h2 r = __builtin_convertvector(f2{a, b}, h2);
    return __builtin_bit_cast(uint32_t, r);
  }
  static __device__ __forceinline__ f32x4 mfma(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t fp8x2(uint32_t src, bool high_half) {
    typedef __attribute__((ext_vector_type(2))) _Float16 v2;
    const v2 r = high_half ? __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(src, 1.0f, true)
                           : __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(src, 1.0f, false);
    return __builtin_bit_cast(uint32_t, r);
  }
};

template <typename T>
__device__ __forceinline__ float round_f(float f) {
  return T::to_f(T::from_f(f));
}

// An fp32 value that is about to be rounded to 16 bits, pinned as an fp32 register value: without
// this the backend may fold the conversion into the instruction that produces the value (a
// mixed-precision fma rounds the exact result once, not to fp32 first) in one kernel and not in
// another, and the two disagree where the fp32 value is a tie on the 16-bit grid.
__device__ __forceinline__ float pin_f32(float v) {
  asm volatile("" : "+v"(v));
  return v;
}

// SwiGLU gate on values already on the 16-bit grid: round(silu(g)) * u (the caller rounds the
// product).  One definition for every kernel; contraction off and the quotient pinned, so that
// the "1 +" is not fused with the last multiply inside the inlined expf and the rounding of
// silu(g) is not folded into the last step of the division.
template <typename T>
__device__ __forceinline__ float silu_mul(float g, float u) {
#pragma clang fp contract(off)
  const float e = expf(-g);
  const float d = 1.0f + e;
  const float s = round_f<T>(pin_f32(g / d));
  return s * u;
}

// 8 packed 16-bit words <-> 8 floats
template <typename T>
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = T::to_f(static_cast<uint16_t>(w[i] & 0xFFFFu));
    f[2 * i + 1] = T::to_f(static_cast<uint16_t>(w[i] >> 16));
  }
}

template <typename T>
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(T::pack2(f[0], f[1]), T::pack2(f[2], f[3]), T::pack2(f[4], f[5]), T::pack2(f[6], f[7]));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- host side ------------------------------------------------------------
int32_t fail(int32_t code, const char* fmt, ...);

#define MI355_REQUIRE(cond, code, ...)                  \
  do {                                                  \
    if (!(cond)) return ::mi355::fail((code), __VA_ARGS__); \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Call after a launch: turns a HIP launch error into MI355_ERR_LAUNCH.
int32_t check_launch(const char* what);

}  // namespace mi355
