// LayerNorm with weight and bias (the norm of OPT-style decoders), alone, after a residual add,
// and after the token-embedding + learned-position sum that starts such a model's step; ReLU.
//
//   y = round(((x - mean) * rsqrt(var + eps)) * w + b),  var = mean((x - mean)^2)
//
// in fp32 with one rounding at the end (what a 16-bit torch.nn.LayerNorm computes).  One
// 256-thread workgroup per token row; a thread owns 16-byte vectors (8 elements) of the row and
// keeps them in registers across both reductions (rows up to 8192 wide; wider rows are re-read,
// L2 hits).  Two passes over the registers -- the mean first, then the sum of squared deviations
// from it -- so that rows with a large mean lose nothing to cancellation.  Reductions: wave64
// shuffles + one LDS exchange.  HBM-bound: 2 bytes in + 2 bytes out per element (+ residual
// read/write in the fused forms).
#include "common.h"

namespace mi355 {

constexpr int kLnThreads = 256;
constexpr int kLnMaxVec = 4;  // vectors cached per thread: 256 * 4 * 8 = 8192 elements

enum { LN_PLAIN = 0, LN_ADD = 1, LN_EMBED = 2 };

struct LnEmbedArgs {            // LN_EMBED: the row is round(embed[id] + pos_embed[position + offset])
  const uint16_t* embed;
  const int64_t* ids;
  int64_t vocab, embed_stride;
  const uint16_t* pos_embed;
  const int64_t* positions;
  int64_t num_positions, position_offset, pos_stride;
};

__device__ __forceinline__ float block_sum(float v, float* part) {
  v = wave_sum(v);
  __syncthreads();                                   // `part` may still be read from the previous sum
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  float total = 0.f;
#pragma unroll
  for (int i = 0; i < kLnThreads / 64; ++i) total += part[i];
  return total;
}

template <typename T, int MODE, bool CACHED>
__global__ __launch_bounds__(kLnThreads) void layer_norm_kernel(
    uint16_t* out, const uint16_t* x, uint16_t* residual,  // out may alias x
    const uint16_t* __restrict__ weight, const uint16_t* __restrict__ bias, int hidden, int64_t x_stride,
    int64_t out_stride, int64_t res_stride, float eps, LnEmbedArgs ea) {
  const int64_t row = blockIdx.x;
  const int nvec = hidden >> 3;
  const uint4* xr = MODE == LN_EMBED ? nullptr : reinterpret_cast<const uint4*>(x + row * x_stride);
  uint4* rr = MODE != LN_PLAIN ? reinterpret_cast<uint4*>(residual + row * res_stride) : nullptr;
  uint4* orow = reinterpret_cast<uint4*>(out + row * out_stride);
  const uint4* wv = reinterpret_cast<const uint4*>(weight);
  const uint4* bv = reinterpret_cast<const uint4*>(bias);
  const uint4 *er = nullptr, *pr = nullptr;
  if (MODE == LN_EMBED) {
    // ids and positions outside their tables read row 0 / the last row (the engine validates both
    // on admission; a padded row of a batch bucket may hold anything)
    int64_t id = MI355_LD(int64_t, ea.ids, ea.ids + row);
    id = (id < 0 || id >= ea.vocab) ? 0 : id;
    int64_t pos = MI355_LD(int64_t, ea.positions, ea.positions + row) + ea.position_offset;
    pos = pos < 0 ? 0 : (pos >= ea.num_positions ? ea.num_positions - 1 : pos);
    er = reinterpret_cast<const uint4*>(ea.embed + id * ea.embed_stride);
    pr = reinterpret_cast<const uint4*>(ea.pos_embed + pos * ea.pos_stride);
  }
  __shared__ float part[kLnThreads / 64];

  // the row's values as the norm sees them; the fused forms also write them back as the residual
  auto load_row_vec = [&](int v, float (&f)[8]) {
    if (MODE == LN_EMBED) {
      float p[8];
      unpack8<T>(MI355_LD(uint4, ea.embed, er + v), f);
      unpack8<T>(MI355_LD(uint4, ea.pos_embed, pr + v), p);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = round_f<T>(f[j] + p[j]);
      MI355_ST(uint4, residual, rr + v, pack8<T>(f));
    } else {
      unpack8<T>(MI355_LD(uint4, x, xr + v), f);
      if (MODE == LN_ADD) {
        float r[8];
        unpack8<T>(MI355_LD(uint4, residual, rr + v), r);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = round_f<T>(f[j] + r[j]);
        MI355_ST(uint4, residual, rr + v, pack8<T>(f));
      }
    }
  };
  // second and third pass of rows too wide for the registers: the same thread wrote residual[v]
  auto reload_row_vec = [&](int v, float (&f)[8]) {
    if (MODE == LN_PLAIN) unpack8<T>(MI355_LD(uint4, x, xr + v), f);
    else unpack8<T>(MI355_LD(uint4, residual, rr + v), f);
  };

  float cache[CACHED ? kLnMaxVec : 1][8];
  float s = 0.f;
  if (CACHED) {
#pragma unroll
    for (int i = 0; i < kLnMaxVec; ++i) {
      const int v = threadIdx.x + i * kLnThreads;
      if (v < nvec) {
        load_row_vec(v, cache[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += cache[i][j];
      }
    }
  } else {
    for (int v = threadIdx.x; v < nvec; v += kLnThreads) {
      float f[8];
      load_row_vec(v, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += f[j];
    }
  }
  const float mean = block_sum(s, part) / static_cast<float>(hidden);

  float ss = 0.f;
  if (CACHED) {
#pragma unroll
    for (int i = 0; i < kLnMaxVec; ++i) {
      const int v = threadIdx.x + i * kLnThreads;
      if (v < nvec) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float dlt = cache[i][j] - mean;
          ss += dlt * dlt;
        }
      }
    }
  } else {
    for (int v = threadIdx.x; v < nvec; v += kLnThreads) {
      float f[8];
      reload_row_vec(v, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dlt = f[j] - mean;
        ss += dlt * dlt;
      }
    }
  }
  const float inv = 1.0f / sqrtf(block_sum(ss, part) / static_cast<float>(hidden) + eps);

  auto finish = [&](int v, const float (&f)[8]) {
#pragma clang fp contract(off)
    float w[8], b[8], y[8];
    unpack8<T>(MI355_LD(uint4, weight, wv + v), w);
    unpack8<T>(MI355_LD(uint4, bias, bv + v), b);
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = ((f[j] - mean) * inv) * w[j] + b[j];
    MI355_ST(uint4, out, orow + v, pack8<T>(y));
  };
  if (CACHED) {
#pragma unroll
    for (int i = 0; i < kLnMaxVec; ++i) {
      const int v = threadIdx.x + i * kLnThreads;
      if (v < nvec) finish(v, cache[i]);
    }
  } else {
    for (int v = threadIdx.x; v < nvec; v += kLnThreads) {
      float f[8];
      reload_row_vec(v, f);
      finish(v, f);
    }
  }
}

// x < 0 ? 0 : x on 16-bit words (NaN and -0.0 pass through, as torch.relu leaves them)
template <typename T>
__global__ __launch_bounds__(256) void relu_kernel(uint16_t* out, const uint16_t* x, int64_t nvec) {
  for (int64_t v = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; v < nvec;
       v += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    float f[8];
    const uint4 raw = MI355_LD(uint4, x, reinterpret_cast<const uint4*>(x) + v);
    unpack8<T>(raw, f);
    uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (f[j] < 0.f) w[j >> 1] &= (j & 1) ? 0x0000FFFFu : 0xFFFF0000u;
    MI355_ST(uint4, out, reinterpret_cast<uint4*>(out) + v, make_uint4(w[0], w[1], w[2], w[3]));
  }
}

template <int MODE>
static int32_t launch_ln(const char* name, void* out, const void* x, void* residual, const void* weight,
                         const void* bias, int64_t num_tokens, int32_t hidden, int64_t x_stride, int64_t out_stride,
                         int64_t res_stride, float eps, int32_t dtype, void* stream, const LnEmbedArgs& ea) {
  MI355_REQUIRE(out && weight && bias && (MODE == LN_EMBED || x) && (MODE == LN_PLAIN || residual), MI355_ERR_INVALID,
                "%s: null pointer", name);
  MI355_REQUIRE(num_tokens >= 0 && num_tokens <= 0x7FFFFFFF, MI355_ERR_INVALID, "%s: num_tokens %lld out of range",
                name, (long long)num_tokens);
  MI355_REQUIRE(hidden > 0 && hidden % 8 == 0, MI355_ERR_UNSUPPORTED, "%s: hidden %d must be a positive multiple of 8",
                name, hidden);
  MI355_REQUIRE(out_stride >= hidden && out_stride % 8 == 0 && (MODE == LN_EMBED || (x_stride >= hidden && x_stride % 8 == 0)) &&
                    (MODE == LN_PLAIN || (res_stride >= hidden && res_stride % 8 == 0)),
                MI355_ERR_INVALID, "%s: row strides must be >= hidden and multiples of 8", name);
  MI355_REQUIRE(aligned16(out) && aligned16(weight) && aligned16(bias) && (MODE == LN_EMBED || aligned16(x)) &&
                    (MODE == LN_PLAIN || aligned16(residual)),
                MI355_ERR_INVALID, "%s: pointers must be 16-byte aligned", name);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "%s: dtype %d not supported (bf16/fp16 only)", name, dtype);
  if (MODE == LN_EMBED) {
    MI355_REQUIRE(ea.embed && ea.ids && ea.pos_embed && ea.positions && aligned16(ea.embed) && aligned16(ea.pos_embed),
                  MI355_ERR_INVALID, "%s: embedding tables / ids / positions missing or misaligned", name);
    MI355_REQUIRE(ea.vocab > 0 && ea.num_positions > 0 && ea.embed_stride >= hidden && ea.embed_stride % 8 == 0 &&
                      ea.pos_stride >= hidden && ea.pos_stride % 8 == 0,
                  MI355_ERR_INVALID, "%s: vocab / num_positions must be positive, table strides >= hidden and "
                  "multiples of 8", name);
  }
  if (num_tokens == 0) return MI355_OK;
  const bool cached = (hidden >> 3) <= kLnThreads * kLnMaxVec;
  dim3 grid(static_cast<unsigned>(num_tokens)), block(kLnThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define MI355_LN(TT, CC)                                                                                          \
  hipLaunchKernelGGL((layer_norm_kernel<TT, MODE, CC>), grid, block, 0, s, static_cast<uint16_t*>(out),            \
                     static_cast<const uint16_t*>(x), static_cast<uint16_t*>(residual),                             \
                     static_cast<const uint16_t*>(weight), static_cast<const uint16_t*>(bias), hidden, x_stride,    \
                     out_stride, res_stride, eps, ea)
  if (dtype == MI355_BF16) {
    if (cached) MI355_LN(BF16, true); else MI355_LN(BF16, false);
  } else {
    if (cached) MI355_LN(F16, true); else MI355_LN(F16, false);
  }
#undef MI355_LN
  return check_launch(name);
}

}  // namespace mi355

extern "C" int32_t mi355_layer_norm(void* out, const void* x, const void* weight, const void* bias, int64_t num_tokens,
                                    int32_t hidden, int64_t x_stride, int64_t out_stride, float eps, int32_t dtype,
                                    void* stream) {
  return mi355::launch_ln<mi355::LN_PLAIN>("mi355_layer_norm", out, x, nullptr, weight, bias, num_tokens, hidden,
                                           x_stride, out_stride, 0, eps, dtype, stream, mi355::LnEmbedArgs{});
}

extern "C" int32_t mi355_fused_add_layer_norm(void* out, const void* x, void* residual, const void* weight,
                                              const void* bias, int64_t num_tokens, int32_t hidden, int64_t x_stride,
                                              int64_t out_stride, int64_t residual_stride, float eps, int32_t dtype,
                                              void* stream) {
  return mi355::launch_ln<mi355::LN_ADD>("mi355_fused_add_layer_norm", out, x, residual, weight, bias, num_tokens,
                                         hidden, x_stride, out_stride, residual_stride, eps, dtype, stream,
                                         mi355::LnEmbedArgs{});
}

extern "C" int32_t mi355_embed_pos_layer_norm(void* out, void* residual, const void* embed, const int64_t* token_ids,
                                              int64_t vocab, int64_t embed_stride, const void* pos_embed,
                                              const int64_t* positions, int64_t num_positions, int64_t position_offset,
                                              int64_t pos_stride, const void* weight, const void* bias,
                                              int64_t num_tokens, int32_t hidden, int64_t out_stride,
                                              int64_t residual_stride, float eps, int32_t dtype, void* stream) {
  const mi355::LnEmbedArgs ea{static_cast<const uint16_t*>(embed), token_ids, vocab, embed_stride,
                              static_cast<const uint16_t*>(pos_embed), positions, num_positions, position_offset,
                              pos_stride};
  return mi355::launch_ln<mi355::LN_EMBED>("mi355_embed_pos_layer_norm", out, nullptr, residual, weight, bias,
                                           num_tokens, hidden, 0, out_stride, residual_stride, eps, dtype, stream, ea);
}

extern "C" int32_t mi355_relu(void* out, const void* x, int64_t num_elements, int32_t dtype, void* stream) {
  using namespace mi355;
  const char* name = "mi355_relu";
  MI355_REQUIRE(out && x, MI355_ERR_INVALID, "%s: null pointer", name);
  MI355_REQUIRE(num_elements >= 0 && num_elements % 8 == 0, MI355_ERR_UNSUPPORTED,
                "%s: the element count %lld must be a non-negative multiple of 8", name, (long long)num_elements);
  MI355_REQUIRE(aligned16(out) && aligned16(x), MI355_ERR_INVALID, "%s: pointers must be 16-byte aligned", name);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "%s: dtype %d not supported (bf16/fp16 only)", name, dtype);
  if (num_elements == 0) return MI355_OK;
  const int64_t nvec = num_elements >> 3;
  const int64_t blocks = (nvec + 255) / 256;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(blocks > 4096 ? 4096 : blocks)), block(256);
  if (dtype == MI355_BF16)
    hipLaunchKernelGGL((relu_kernel<BF16>), grid, block, 0, s, static_cast<uint16_t*>(out),
                       static_cast<const uint16_t*>(x), nvec);
  else
    hipLaunchKernelGGL((relu_kernel<F16>), grid, block, 0, s, static_cast<uint16_t*>(out),
                       static_cast<const uint16_t*>(x), nvec);
  return check_launch(name);
}
