// Shared device helpers and host-side argument checking for libmi355_decode.so.
// gfx950 only: wave = 64 lanes, MFMA 16x16x32 for bf16 / fp16.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stddef.h>

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

// ---- global-memory accesses of the kernels ----------------------------------
// MI355_LD(V, base, p)        load a V from p
// MI355_LDNT(V, base, p)      the same, non-temporal
// MI355_ST(V, base, p, v)     store v to p
// MI355_OK_AT(base, p, bytes) guard of an access of another kind (an atomic): true = go ahead
// `base` is the kernel argument (the buffer) the address was computed from.  In the library as
// shipped these are the plain accesses and `base` is not looked at.  Built with -DMI355_BOUNDS
// (make bounds -> libmi355_decode_bounds.so, a debugging aid, many times slower) every access is
// first compared with the allocation its base pointer lies in: the host hands the library the
// list of live device allocations (mi355_bounds_set_ranges), an access that leaves the allocation
// of its base is NOT performed (a load yields zeros), and the first such access -- source line,
// base, address, size -- is kept for mi355_bounds_report.  Nothing faults, so one run names the
// access instead of killing the process.  Before any list is set every access passes; an access
// whose base the list does not hold (memory allocated after the list was set) is performed and
// counted.
#ifdef MI355_BOUNDS
struct BoundsRange {
  uint64_t lo, hi;            // [lo, hi)
};
constexpr unsigned kBoundsMaxRanges = 4096;
struct BoundsFirst {
  unsigned hits, line;        // accesses refused since the last report; line of the first one
  uint64_t base, addr, bytes;
  unsigned long long unknown; // accesses whose base lies in no range of the list (memory allocated
                              // after the list was set): performed, only counted
};
struct BoundsState {
  unsigned count, pad;        // ranges in use: sorted by lo, disjoint
  BoundsFirst first;
  BoundsRange r[kBoundsMaxRanges];
};
// one copy per translation unit (the library is not built with relocatable device code); every
// unit registers the two host functions below, and the entry points in abi.hip walk the list
static __device__ BoundsState g_bounds;

struct BoundsUnit {
  const char* file;
  hipError_t (*upload)(const BoundsRange* r, unsigned count);
  hipError_t (*collect)(BoundsFirst* out);        // reads and clears
};
void bounds_register_unit(const BoundsUnit& u);    // abi.hip

static hipError_t bounds_upload(const BoundsRange* r, unsigned count) {
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_bounds), r, sizeof(BoundsRange) * count, offsetof(BoundsState, r));
  if (e != hipSuccess) return e;
  const unsigned head[2] = {count, 0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_bounds), head, sizeof(head), 0);
}
static hipError_t bounds_collect(BoundsFirst* out) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bounds), sizeof(BoundsFirst), offsetof(BoundsState, first));
  if (e != hipSuccess) return e;
  const BoundsFirst zero = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_bounds), &zero, sizeof(zero), offsetof(BoundsState, first));
}
static const struct BoundsAuto {
  BoundsAuto() { bounds_register_unit(BoundsUnit{__BASE_FILE__, bounds_upload, bounds_collect}); }
} g_bounds_auto;

__device__ __noinline__ bool bounds_ok(const void* base, const void* p, unsigned bytes, unsigned line) {
  const unsigned n = g_bounds.count;
  if (n == 0) return true;
  const uint64_t b = reinterpret_cast<uint64_t>(base), a = reinterpret_cast<uint64_t>(p);
  unsigned lo = 0, hi = n;                          // first range whose end lies beyond the base
  while (lo < hi) {
    const unsigned mid = (lo + hi) >> 1;
    if (g_bounds.r[mid].hi <= b) lo = mid + 1; else hi = mid;
  }
  if (lo >= n || g_bounds.r[lo].lo > b) {           // a base the list does not know
    atomicAdd(&g_bounds.first.unknown, 1ull);
    return true;
  }
  if (a >= g_bounds.r[lo].lo && a + bytes <= g_bounds.r[lo].hi) return true;
  if (atomicAdd(&g_bounds.first.hits, 1u) == 0u) {
    g_bounds.first.line = line;
    g_bounds.first.base = b;
    g_bounds.first.addr = a;
    g_bounds.first.bytes = bytes;
  }
  return false;
}
template <typename V>
__device__ __forceinline__ V bounds_load(const void* base, const void* p, unsigned line) {
  if (!bounds_ok(base, p, sizeof(V), line)) return V{};
  return *reinterpret_cast<const V*>(p);
}
template <typename V, typename W>
__device__ __forceinline__ void bounds_store(const void* base, void* p, const W& v, unsigned line) {
  if (bounds_ok(base, p, sizeof(V), line)) *reinterpret_cast<V*>(p) = v;
}
#define MI355_LD(V, base, p) (::mi355::bounds_load<V>((base), (p), __LINE__))
#define MI355_LDNT(V, base, p) (::mi355::bounds_load<V>((base), (p), __LINE__))
#define MI355_ST(V, base, p, v) (::mi355::bounds_store<V>((base), (p), (v), __LINE__))
#define MI355_OK_AT(base, p, bytes) (::mi355::bounds_ok((base), (p), (bytes), __LINE__))
#else
#define MI355_LD(V, base, p) (*reinterpret_cast<const V*>(p))
#define MI355_LDNT(V, base, p) (__builtin_nontemporal_load(reinterpret_cast<const V*>(p)))
#define MI355_ST(V, base, p, v) (*reinterpret_cast<V*>(p) = (v))
#define MI355_OK_AT(base, p, bytes) true
#endif

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
