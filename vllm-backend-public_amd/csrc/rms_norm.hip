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
        const float4 a = p[0], b = p[1];
        f[0] += a.x; f[1] += a.y; f[2] += a.z; f[3] += a.w;
        f[4] += b.x; f[5] += b.y; f[6] += b.z; f[7] += b.w;
      }
      if (bias != nullptr) {
        float bb[8];
        unpack8<T>(reinterpret_cast<const uint4*>(bias)[v], bb);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] += bb[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = round_f<T>(f[j]);
    } else {
      unpack8<T>(xr[v], f);
    }
    if (ADD) {
      float r[8];
      unpack8<T>(rr[v], r);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = round_f<T>(f[j] + r[j]);
      rr[v] = pack8<T>(f);
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
    unpack8<T>(wv[v], w);
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = round_f<T>(f[j] * inv) * w[j];
    orow[v] = pack8<T>(y);
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
      unpack8<T>(src[v], f);
      finish(v, f);
    }
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
