// RMSNorm and residual-add + RMSNorm.
//
// One 256-thread workgroup per token row.  Each thread owns 16-byte vectors
// (8 elements) of the row, keeps them in registers across the reduction (rows up
// to 8192 wide; wider rows are re-read, an L2 hit), reduces the sum of squares
// with wave64 shuffles + one LDS exchange, and writes the result with 16-byte
// stores.  HBM-bound: 2 bytes in + 2 bytes out per element (plus the residual
// read/write in the fused form).
//
// The PART form takes its x from the fp32 partial-sum slabs a K-split GEMM left
// behind ([slab][token][hidden]): x = round(slab_0 + slab_1 + ... + bias), summed
// in slab order, so the GEMM's reduction, the residual add and the norm are one
// pass over the row.
#include "common.h"

namespace mi355 {

constexpr int kThreads = 256;
constexpr int kMaxVec = 4;  // vectors cached per thread: 256 * 4 * 8 = 8192 elements

template <typename T, bool ADD, bool CACHED, bool PART>
__global__ __launch_bounds__(kThreads) void rms_norm_kernel(
    uint16_t* out, const uint16_t* x, uint16_t* residual,  // out may alias x
    const uint16_t* __restrict__ weight, int hidden, int64_t x_stride, int64_t out_stride,
    int64_t res_stride, float eps, const float* __restrict__ partial, int num_slabs, int64_t slab_stride,
    const uint16_t* __restrict__ bias) {
  const int64_t row = blockIdx.x;
  const int nvec = hidden >> 3;
  const uint4* xr = PART ? nullptr : reinterpret_cast<const uint4*>(x + row * x_stride);
  const float* pr = PART ? partial + row * hidden : nullptr;
  uint4* rr = ADD ? reinterpret_cast<uint4*>(residual + row * res_stride) : nullptr;
  uint4* orow = reinterpret_cast<uint4*>(out + row * out_stride);
  const uint4* wv = reinterpret_cast<const uint4*>(weight);

  float cache[CACHED ? kMaxVec : 1][8];
  float ss = 0.f;

  auto load_row_vec = [&](int v, float (&f)[8]) {
    if (PART) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = 0.f;
      for (int s = 0; s < num_slabs; ++s) {
        const float4* p = reinterpret_cast<const float4*>(pr + s * slab_stride + 8 * v);
        const float4 a = MI355_LD(float4, partial, p), b = MI355_LD(float4, partial, p + 1);
        f[0] += a.x; f[1] += a.y; f[2] += a.z; f[3] += a.w;
        f[4] += b.x; f[5] += b.y; f[6] += b.z; f[7] += b.w;
      }
      if (bias != nullptr) {
        float bb[8];
        unpack8<T>(MI355_LD(uint4, bias, reinterpret_cast<const uint4*>(bias) + v), bb);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] += bb[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = round_f<T>(f[j]);
    } else {
      unpack8<T>(MI355_LD(uint4, x, xr + v), f);
    }
    if (ADD) {
      float r[8];
      unpack8<T>(MI355_LD(uint4, residual, rr + v), r);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = round_f<T>(f[j] + r[j]);
      MI355_ST(uint4, residual, rr + v, pack8<T>(f));
    }
  };

  if (CACHED) {
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
      const int v = threadIdx.x + i * kThreads;
      if (v < nvec) {
        load_row_vec(v, cache[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += cache[i][j] * cache[i][j];
      }
    }
  } else {
    for (int v = threadIdx.x; v < nvec; v += kThreads) {
      float f[8];
      load_row_vec(v, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
    }
  }

  __shared__ float part[kThreads / 64];
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
  __syncthreads();
  float total = 0.f;
#pragma unroll
  for (int i = 0; i < kThreads / 64; ++i) total += part[i];
  const float inv = 1.0f / sqrtf(total / static_cast<float>(hidden) + eps);

  auto finish = [&](int v, const float (&f)[8]) {
    float w[8], y[8];
    unpack8<T>(MI355_LD(uint4, weight, wv + v), w);
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = round_f<T>(f[j] * inv) * w[j];
    MI355_ST(uint4, out, orow + v, pack8<T>(y));
  };

  if (CACHED) {
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
      const int v = threadIdx.x + i * kThreads;
      if (v < nvec) finish(v, cache[i]);
    }
  } else {
    // The same thread wrote residual[v] above, so its own re-read sees it.
    const uint4* src = ADD ? rr : xr;
    for (int v = threadIdx.x; v < nvec; v += kThreads) {
      float f[8];
      unpack8<T>(MI355_LD(uint4, ADD ? residual : x, src + v), f);
      finish(v, f);
    }
  }
}

// The PART + ADD form for the slab counts and widths of the decode step, with every load of the
// row -- SLABS x NV slab vectors, the residual, the bias, the norm weight -- issued before the
// first value is used.  The generic kernel walks the slabs in a run-time loop, one dependent
// round trip to memory per slab; this one pays one.  A thread owns NV 8-element vectors of the
// row (vector indices beyond the row are clamped: loaded, never stored).  Same arithmetic, in the
// same order, as the generic kernel.
// SLABS == 0: x is a 16-bit tensor (the plain fused add + norm), loaded the same way.
template <typename T, int SLABS, int NV>
__global__ __launch_bounds__(kThreads) void rms_norm_slabs_kernel(
    uint16_t* out, uint16_t* residual, const uint16_t* __restrict__ weight, int hidden,
    int64_t out_stride, int64_t res_stride, float eps, const float* __restrict__ partial, int64_t slab_stride,
    const uint16_t* __restrict__ bias, const uint16_t* x16, int64_t x_stride) {
  const int64_t row = blockIdx.x;
  const int nvec = hidden >> 3;
  const float* pr = SLABS > 0 ? partial + row * hidden : nullptr;
  const uint4* xr = SLABS > 0 ? nullptr : reinterpret_cast<const uint4*>(x16 + row * x_stride);
  uint4* rr = reinterpret_cast<uint4*>(residual + row * res_stride);
  uint4* orow = reinterpret_cast<uint4*>(out + row * out_stride);
  const uint4* wv = reinterpret_cast<const uint4*>(weight);
  // a null bias reads the norm weight instead and is multiplied away: no branch around a load
  const uint4* bv = reinterpret_cast<const uint4*>(bias != nullptr ? bias : weight);
  const float has_bias = bias != nullptr ? 1.f : 0.f;

  float4 a[NV][SLABS > 0 ? SLABS : 1][2];
  uint4 res[NV], wr[NV], br[NV], xv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = min(static_cast<int>(threadIdx.x) + i * kThreads, nvec - 1);
#pragma unroll
    for (int s = 0; s < SLABS; ++s) {
      const float4* p = reinterpret_cast<const float4*>(pr + s * slab_stride + 8 * v);
      a[i][s][0] = MI355_LD(float4, partial, p);
      a[i][s][1] = MI355_LD(float4, partial, p + 1);
    }
    if (SLABS == 0) xv[i] = MI355_LD(uint4, x16, xr + v);
    res[i] = MI355_LD(uint4, residual, rr + v);
    br[i] = MI355_LD(uint4, bias != nullptr ? bias : weight, bv + v);
    wr[i] = MI355_LD(uint4, weight, wv + v);
  }
  __builtin_amdgcn_sched_barrier(0);

  float f[NV][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = static_cast<int>(threadIdx.x) + i * kThreads;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[i][j] = 0.f;
#pragma unroll
    for (int s = 0; s < SLABS; ++s) {
      f[i][0] += a[i][s][0].x; f[i][1] += a[i][s][0].y; f[i][2] += a[i][s][0].z; f[i][3] += a[i][s][0].w;
      f[i][4] += a[i][s][1].x; f[i][5] += a[i][s][1].y; f[i][6] += a[i][s][1].z; f[i][7] += a[i][s][1].w;
    }
    float bb[8], r[8];
    unpack8<T>(br[i], bb);
    unpack8<T>(res[i], r);
    if (SLABS == 0) {
      unpack8<T>(xv[i], f[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[i][j] = round_f<T>(f[i][j] + r[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (has_bias != 0.f) f[i][j] += bb[j];
        f[i][j] = round_f<T>(round_f<T>(f[i][j]) + r[j]);
      }
    }
    if (v < nvec) {
      MI355_ST(uint4, residual, rr + v, pack8<T>(f[i]));
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += f[i][j] * f[i][j];
    }
  }

  __shared__ float part[kThreads / 64];
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
  __syncthreads();
  float total = 0.f;
#pragma unroll
  for (int i = 0; i < kThreads / 64; ++i) total += part[i];
  const float inv = 1.0f / sqrtf(total / static_cast<float>(hidden) + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = static_cast<int>(threadIdx.x) + i * kThreads;
    float w[8], y[8];
    unpack8<T>(wr[i], w);
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = round_f<T>(f[i][j] * inv) * w[j];
    if (v < nvec) MI355_ST(uint4, out, orow + v, pack8<T>(y));
  }
}

// First kernel of a decode step: the embedding gather and the first layer's RMSNorm in one
// launch.  Row b of the residual stream is embedding row ids[b] (copied as is), x is its norm.
// An id outside [0, vocab) reads row 0 (the engine validates ids on admission; a padded row of a
// batch bucket may hold anything).
template <typename T>
__global__ __launch_bounds__(kThreads) void embed_rms_norm_kernel(
    uint16_t* __restrict__ out, uint16_t* __restrict__ residual, const uint16_t* __restrict__ embed,
    const int64_t* __restrict__ ids, int64_t vocab, const uint16_t* __restrict__ weight, int hidden,
    int64_t embed_stride, int64_t out_stride, int64_t res_stride, float eps) {
  const int64_t row = blockIdx.x;
  int64_t id = MI355_LD(int64_t, ids, ids + row);
  id = (id < 0 || id >= vocab) ? 0 : id;
  const int nvec = hidden >> 3;
  const uint4* er = reinterpret_cast<const uint4*>(embed + id * embed_stride);
  uint4* rr = reinterpret_cast<uint4*>(residual + row * res_stride);
  uint4* orow = reinterpret_cast<uint4*>(out + row * out_stride);
  const uint4* wv = reinterpret_cast<const uint4*>(weight);
  float ss = 0.f;
  for (int v = threadIdx.x; v < nvec; v += kThreads) {
    const uint4 raw = MI355_LD(uint4, embed, er + v);
    MI355_ST(uint4, residual, rr + v, raw);
    float f[8];
    unpack8<T>(raw, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
  }
  __shared__ float part[kThreads / 64];
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
  __syncthreads();
  float total = 0.f;
#pragma unroll
  for (int i = 0; i < kThreads / 64; ++i) total += part[i];
  const float inv = 1.0f / sqrtf(total / static_cast<float>(hidden) + eps);
  for (int v = threadIdx.x; v < nvec; v += kThreads) {
    float f[8], w[8], y[8];
    unpack8<T>(MI355_LD(uint4, embed, er + v), f);                 // an L2 hit: this thread just read it
    unpack8<T>(MI355_LD(uint4, weight, wv + v), w);
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = round_f<T>(f[j] * inv) * w[j];
    MI355_ST(uint4, out, orow + v, pack8<T>(y));
  }
}

template <bool ADD, bool PART = false>
static int32_t launch(void* out, const void* x, void* residual, const void* weight,
                      int64_t num_tokens, int32_t hidden, int64_t x_stride, int64_t out_stride,
                      int64_t res_stride, float eps, int32_t dtype, void* stream,
                      const float* partial = nullptr, int num_slabs = 0, int64_t slab_stride = 0,
                      const void* bias = nullptr) {
  const char* name = PART ? "mi355_fused_add_rms_norm_partial" : ADD ? "mi355_fused_add_rms_norm" : "mi355_rms_norm";
  MI355_REQUIRE(out && x && weight && (!ADD || residual), MI355_ERR_INVALID, "%s: null pointer", name);
  if (PART) {
    MI355_REQUIRE(num_slabs >= 1 && num_slabs <= 64, MI355_ERR_INVALID, "%s: num_slabs %d outside [1, 64]", name,
                  num_slabs);
    MI355_REQUIRE(slab_stride >= num_tokens * hidden && slab_stride % 4 == 0, MI355_ERR_INVALID,
                  "%s: slab_stride %lld must be >= num_tokens * hidden and a multiple of 4", name,
                  (long long)slab_stride);
    MI355_REQUIRE(bias == nullptr || aligned16(bias), MI355_ERR_INVALID, "%s: bias must be 16-byte aligned", name);
  }
  MI355_REQUIRE(num_tokens >= 0 && num_tokens <= 0x7FFFFFFF, MI355_ERR_INVALID,
                "%s: num_tokens %lld out of range", name, (long long)num_tokens);
  MI355_REQUIRE(hidden > 0 && hidden % 8 == 0, MI355_ERR_UNSUPPORTED,
                "%s: hidden %d must be a positive multiple of 8", name, hidden);
  MI355_REQUIRE(x_stride >= hidden && out_stride >= hidden && x_stride % 8 == 0 && out_stride % 8 == 0 &&
                    (!ADD || (res_stride >= hidden && res_stride % 8 == 0)),
                MI355_ERR_INVALID, "%s: row strides must be >= hidden and multiples of 8", name);
  MI355_REQUIRE(aligned16(out) && aligned16(x) && aligned16(weight) && (!ADD || aligned16(residual)),
                MI355_ERR_INVALID, "%s: pointers must be 16-byte aligned", name);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "%s: dtype %d not supported (bf16/fp16 only)", name, dtype);
  if (num_tokens == 0) return MI355_OK;

  const bool cached = (hidden >> 3) <= kThreads * kMaxVec;
  dim3 grid(static_cast<unsigned>(num_tokens)), block(kThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto* o = static_cast<uint16_t*>(out);
  auto* xi = static_cast<const uint16_t*>(x);
  auto* r = static_cast<uint16_t*>(residual);
  auto* w = static_cast<const uint16_t*>(weight);
  auto* bp = static_cast<const uint16_t*>(bias);
  // out == x is allowed: a thread reads its vectors of x before it writes the same vectors of out
  if (ADD && (hidden >> 3) <= 4 * kThreads &&
      (!PART || num_slabs == 1 || num_slabs == 2 || num_slabs == 4 || num_slabs == 8)) {
    const int nv = ((hidden >> 3) + kThreads - 1) / kThreads;   // 1..4 vectors per thread
    if (!PART) num_slabs = 0;
#define MI355_RMS_SLABS(TT, SS, NN)                                                                          \
  hipLaunchKernelGGL((rms_norm_slabs_kernel<TT, SS, NN>), grid, block, 0, s, o, r, w, hidden, out_stride,     \
                     res_stride, eps, partial, slab_stride, bp, PART ? nullptr : xi, x_stride)
#define MI355_RMS_SLABS_NV(TT, SS)                                                                           \
  do {                                                                                                       \
    if (nv == 1) MI355_RMS_SLABS(TT, SS, 1);                                                                 \
    else if (nv == 2) MI355_RMS_SLABS(TT, SS, 2);                                                            \
    else if (nv == 3) MI355_RMS_SLABS(TT, SS, 3);                                                            \
    else MI355_RMS_SLABS(TT, SS, 4);                                                                         \
  } while (0)
#define MI355_RMS_SLABS_T(TT)                                                                                \
  do {                                                                                                       \
    if (num_slabs == 0) MI355_RMS_SLABS_NV(TT, 0);                                                           \
    else if (num_slabs == 1) MI355_RMS_SLABS_NV(TT, 1);                                                      \
    else if (num_slabs == 2) MI355_RMS_SLABS_NV(TT, 2);                                                      \
    else if (num_slabs == 4) MI355_RMS_SLABS_NV(TT, 4);                                                      \
    else MI355_RMS_SLABS_NV(TT, 8);                                                                          \
  } while (0)
    if (dtype == MI355_BF16) MI355_RMS_SLABS_T(BF16); else MI355_RMS_SLABS_T(F16);
#undef MI355_RMS_SLABS_T
#undef MI355_RMS_SLABS_NV
#undef MI355_RMS_SLABS
    return check_launch(name);
  }
#define MI355_RMS_LAUNCH(TT, CC)                                                                         \
  hipLaunchKernelGGL((rms_norm_kernel<TT, ADD, CC, PART>), grid, block, 0, s, o, xi, r, w, hidden, x_stride, \
                     out_stride, res_stride, eps, partial, num_slabs, slab_stride, bp)
  if (dtype == MI355_BF16) {
    if (cached) MI355_RMS_LAUNCH(BF16, true); else MI355_RMS_LAUNCH(BF16, false);
  } else {
    if (cached) MI355_RMS_LAUNCH(F16, true); else MI355_RMS_LAUNCH(F16, false);
  }
#undef MI355_RMS_LAUNCH
  return check_launch(name);
}

}  // namespace mi355

extern "C" int32_t mi355_rms_norm(void* out, const void* x, const void* weight, int64_t num_tokens,
                                  int32_t hidden, int64_t x_stride, int64_t out_stride, float eps,
                                  int32_t dtype, void* stream) {
  return mi355::launch<false>(out, x, nullptr, weight, num_tokens, hidden, x_stride, out_stride, 0, eps,
                              dtype, stream);
}

extern "C" int32_t mi355_fused_add_rms_norm(void* out, const void* x, void* residual, const void* weight,
                                            int64_t num_tokens, int32_t hidden, int64_t x_stride,
                                            int64_t out_stride, int64_t residual_stride, float eps,
                                            int32_t dtype, void* stream) {
  return mi355::launch<true>(out, x, residual, weight, num_tokens, hidden, x_stride, out_stride,
                             residual_stride, eps, dtype, stream);
}

extern "C" int32_t mi355_fused_add_rms_norm_partial(void* out, const float* partial, int32_t num_slabs,
                                                    int64_t slab_stride, const void* bias, void* residual,
                                                    const void* weight, int64_t num_tokens, int32_t hidden,
                                                    int64_t out_stride, int64_t residual_stride, float eps,
                                                    int32_t dtype, void* stream) {
  // `partial` stands in for x in the shared argument checks (16-byte alignment, non-null)
  return mi355::launch<true, true>(out, partial, residual, weight, num_tokens, hidden, hidden, out_stride,
                                   residual_stride, eps, dtype, stream, partial, num_slabs, slab_stride, bias);
}

extern "C" int32_t mi355_embed_rms_norm(void* out, void* residual, const void* embed, const int64_t* token_ids,
                                        int64_t vocab, const void* weight, int64_t num_tokens, int32_t hidden,
                                        int64_t embed_stride, int64_t out_stride, int64_t residual_stride, float eps,
                                        int32_t dtype, void* stream) {
  using namespace mi355;
  const char* name = "mi355_embed_rms_norm";
  MI355_REQUIRE(out && residual && embed && token_ids && weight, MI355_ERR_INVALID, "%s: null pointer", name);
  MI355_REQUIRE(num_tokens >= 0 && num_tokens <= 0x7FFFFFFF && vocab > 0, MI355_ERR_INVALID,
                "%s: num_tokens %lld / vocab %lld out of range", name, (long long)num_tokens, (long long)vocab);
  MI355_REQUIRE(hidden > 0 && hidden % 8 == 0, MI355_ERR_UNSUPPORTED, "%s: hidden %d must be a positive multiple of 8",
                name, hidden);
  MI355_REQUIRE(embed_stride >= hidden && out_stride >= hidden && residual_stride >= hidden && embed_stride % 8 == 0 &&
                    out_stride % 8 == 0 && residual_stride % 8 == 0,
                MI355_ERR_INVALID, "%s: row strides must be >= hidden and multiples of 8", name);
  MI355_REQUIRE(aligned16(out) && aligned16(residual) && aligned16(embed) && aligned16(weight), MI355_ERR_INVALID,
                "%s: pointers must be 16-byte aligned", name);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "%s: dtype %d not supported (bf16/fp16 only)", name, dtype);
  if (num_tokens == 0) return MI355_OK;
  dim3 grid(static_cast<unsigned>(num_tokens)), block(kThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == MI355_BF16)
    hipLaunchKernelGGL((embed_rms_norm_kernel<BF16>), grid, block, 0, s, static_cast<uint16_t*>(out),
                       static_cast<uint16_t*>(residual), static_cast<const uint16_t*>(embed), token_ids, vocab,
                       static_cast<const uint16_t*>(weight), hidden, embed_stride, out_stride, residual_stride, eps);
  else
    hipLaunchKernelGGL((embed_rms_norm_kernel<F16>), grid, block, 0, s, static_cast<uint16_t*>(out),
                       static_cast<uint16_t*>(residual), static_cast<const uint16_t*>(embed), token_ids, vocab,
                       static_cast<const uint16_t*>(weight), hidden, embed_stride, out_stride, residual_stride, eps);
  return check_launch(name);
}
