// Greedy sampling: per-row argmax over the vocabulary.
//
// One 1024-thread workgroup per row streams the row with 16-byte loads and
// keeps, per thread, a single 64-bit key
//     (order-preserving image of the logit) << 32  |  (0xFFFFFFFF - index)
// whose maximum is the largest logit and, among equal logits, the LOWEST index.
// NaN maps to the top key (torch.argmax treats NaN as the maximum) and -0.0 is
// folded onto +0.0.  Keys are max-reduced with wave64 shuffles and one LDS
// exchange.  HBM-bound single pass: vocab * sizeof(logit) bytes per row.
#include "common.h"

namespace mi355 {

constexpr int kArgmaxThreads = 1024;

__device__ __forceinline__ unsigned long long argmax_key(float v, uint32_t idx) {
  uint32_t u = __float_as_uint(v);
  uint32_t k;
  if (v != v) {
    k = 0xFFFFFFFFu;
  } else {
    if (v == 0.0f) u = 0u;
    k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  }
  return (static_cast<unsigned long long>(k) << 32) | static_cast<unsigned long long>(0xFFFFFFFFu - idx);
}

__device__ __forceinline__ unsigned long long umax64(unsigned long long a, unsigned long long b) {
  return a > b ? a : b;
}

template <int LDT>
__global__ __launch_bounds__(kArgmaxThreads) void argmax_kernel(int64_t* __restrict__ out,
                                                                const void* __restrict__ logits,
                                                                int64_t stride, int vocab) {
  const int row = blockIdx.x;
  unsigned long long best = 0ull;
  if (LDT == MI355_F32) {
    const float* p = static_cast<const float*>(logits) + row * stride;
    const int nvec = vocab >> 2;
    for (int v = threadIdx.x; v < nvec; v += kArgmaxThreads) {
      const float4 f = *reinterpret_cast<const float4*>(p + 4 * v);
      best = umax64(best, argmax_key(f.x, 4 * v));
      best = umax64(best, argmax_key(f.y, 4 * v + 1));
      best = umax64(best, argmax_key(f.z, 4 * v + 2));
      best = umax64(best, argmax_key(f.w, 4 * v + 3));
    }
    for (int i = (nvec << 2) + threadIdx.x; i < vocab; i += kArgmaxThreads)
      best = umax64(best, argmax_key(p[i], i));
  } else {
    const uint16_t* p = static_cast<const uint16_t*>(logits) + row * stride;
    const int nvec = vocab >> 3;
    for (int v = threadIdx.x; v < nvec; v += kArgmaxThreads) {
      float f[8];
      const uint4 raw = *reinterpret_cast<const uint4*>(p + 8 * v);
      if (LDT == MI355_BF16) unpack8<BF16>(raw, f); else unpack8<F16>(raw, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) best = umax64(best, argmax_key(f[j], 8 * v + j));
    }
    for (int i = (nvec << 3) + threadIdx.x; i < vocab; i += kArgmaxThreads) {
      const float f = (LDT == MI355_BF16) ? BF16::to_f(p[i]) : F16::to_f(p[i]);
      best = umax64(best, argmax_key(f, i));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) best = umax64(best, __shfl_xor(best, off, 64));
  __shared__ unsigned long long part[kArgmaxThreads / 64];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long b = part[0];
#pragma unroll
    for (int i = 1; i < kArgmaxThreads / 64; ++i) b = umax64(b, part[i]);
    out[row] = static_cast<int64_t>(0xFFFFFFFFu - static_cast<uint32_t>(b & 0xFFFFFFFFull));
  }
}

}  // namespace mi355

extern "C" int32_t mi355_argmax(int64_t* out_ids, const void* logits, int64_t logits_stride,
                                int32_t num_rows, int32_t vocab, int32_t logits_dtype, void* stream) {
  using namespace mi355;
  const char* name = "mi355_argmax";
  MI355_REQUIRE(out_ids && logits, MI355_ERR_INVALID, "%s: null pointer", name);
  MI355_REQUIRE(num_rows >= 0 && vocab > 0, MI355_ERR_INVALID, "%s: num_rows/vocab out of range", name);
  MI355_REQUIRE(logits_dtype == MI355_BF16 || logits_dtype == MI355_F16 || logits_dtype == MI355_F32,
                MI355_ERR_UNSUPPORTED, "%s: logits dtype %d not supported", name, logits_dtype);
  const int per16 = logits_dtype == MI355_F32 ? 4 : 8;
  MI355_REQUIRE(logits_stride >= vocab && logits_stride % per16 == 0 && aligned16(logits),
                MI355_ERR_INVALID, "%s: logits rows must be 16-byte aligned (stride %lld)", name,
                (long long)logits_stride);
  if (num_rows == 0) return MI355_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid(num_rows), block(kArgmaxThreads);
  if (logits_dtype == MI355_F32)
    hipLaunchKernelGGL((argmax_kernel<MI355_F32>), grid, block, 0, s, out_ids, logits, logits_stride, vocab);
  else if (logits_dtype == MI355_BF16)
    hipLaunchKernelGGL((argmax_kernel<MI355_BF16>), grid, block, 0, s, out_ids, logits, logits_stride, vocab);
  else
    hipLaunchKernelGGL((argmax_kernel<MI355_F16>), grid, block, 0, s, out_ids, logits, logits_stride, vocab);
  return check_launch(name);
}
