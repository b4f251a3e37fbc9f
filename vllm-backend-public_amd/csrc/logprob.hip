// Log-probabilities of chosen tokens:  out[b] = logits[b, ids[b]] - logsumexp(logits[b, :]),
// and the row's logsumexp itself (so a caller can turn any other logit of the row into a
// log-probability by one subtraction).
//
// One 1024-thread workgroup per row, two passes over the row (the second is an L2 hit for
// rows up to a few MB): the row maximum, then the sum of exp(x - max) in fp32.  Every thread
// accumulates its strided elements in index order and the partial sums are added in a fixed
// tree (wave shuffles, then the 16 wave sums in order), so a row always gives the same bits.
// HBM-bound and small: vocab * sizeof(logit) bytes per row.
#include "common.h"

#include <math.h>

namespace mi355 {

constexpr int kLseThreads = 1024;

template <int LDT>
__device__ __forceinline__ float logit_at(const void* row, int i) {
  if (LDT == MI355_F32) return MI355_LD(float, row, static_cast<const float*>(row) + i);
  const uint16_t u = MI355_LD(uint16_t, row, static_cast<const uint16_t*>(row) + i);
  return LDT == MI355_BF16 ? BF16::to_f(u) : F16::to_f(u);
}

template <int LDT>
__global__ __launch_bounds__(kLseThreads) void token_logprob_kernel(float* __restrict__ out_logprob,
                                                                    float* __restrict__ out_lse,
                                                                    const void* __restrict__ logits, int64_t stride,
                                                                    int vocab, const int64_t* __restrict__ ids) {
  const int row = blockIdx.x;
  const char* base = static_cast<const char*>(logits) + row * stride * (LDT == MI355_F32 ? 4 : 2);
  __shared__ float part[kLseThreads / 64];
  __shared__ float bcast;

  auto block_reduce = [&](float v, bool is_max) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float o = __shfl_xor(v, off, 64);
      v = is_max ? fmaxf(v, o) : v + o;
    }
    __syncthreads();                       // part[] may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      float r = part[0];
#pragma unroll
      for (int i = 1; i < kLseThreads / 64; ++i) r = is_max ? fmaxf(r, part[i]) : r + part[i];
      bcast = r;
    }
    __syncthreads();
    return bcast;
  };

  float mx = -INFINITY;
  for (int i = threadIdx.x; i < vocab; i += kLseThreads) mx = fmaxf(mx, logit_at<LDT>(base, i));
  mx = block_reduce(mx, true);
  float sum = 0.f;
  for (int i = threadIdx.x; i < vocab; i += kLseThreads) sum += expf(logit_at<LDT>(base, i) - mx);
  sum = block_reduce(sum, false);
  if (threadIdx.x == 0) {
    const float lse = mx + logf(sum);
    if (out_lse != nullptr) MI355_ST(float, out_lse, out_lse + row, lse);
    if (out_logprob != nullptr) {
      const int64_t id = MI355_LD(int64_t, ids, ids + row);
      MI355_ST(float, out_logprob, out_logprob + row,
               (id >= 0 && id < vocab) ? logit_at<LDT>(base, static_cast<int>(id)) - lse : -INFINITY);
    }
  }
}

}  // namespace mi355

extern "C" int32_t mi355_token_logprobs(float* out_logprob, float* out_lse, const void* logits,
                                        int64_t logits_stride, int32_t num_rows, int32_t vocab,
                                        int32_t logits_dtype, const int64_t* token_ids, void* stream) {
  using namespace mi355;
  const char* name = "mi355_token_logprobs";
  MI355_REQUIRE(logits && (out_logprob || out_lse) && (!out_logprob || token_ids), MI355_ERR_INVALID,
                "%s: null pointer", name);
  MI355_REQUIRE(num_rows >= 0 && vocab > 0 && logits_stride >= vocab, MI355_ERR_INVALID,
                "%s: num_rows/vocab/logits_stride out of range", name);
  MI355_REQUIRE(logits_dtype == MI355_BF16 || logits_dtype == MI355_F16 || logits_dtype == MI355_F32,
                MI355_ERR_UNSUPPORTED, "%s: logits dtype %d not supported", name, logits_dtype);
  if (num_rows == 0) return MI355_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid(num_rows), block(kLseThreads);
  if (logits_dtype == MI355_F32)
    hipLaunchKernelGGL((token_logprob_kernel<MI355_F32>), grid, block, 0, s, out_logprob, out_lse, logits,
                       logits_stride, vocab, token_ids);
  else if (logits_dtype == MI355_BF16)
    hipLaunchKernelGGL((token_logprob_kernel<MI355_BF16>), grid, block, 0, s, out_logprob, out_lse, logits,
                       logits_stride, vocab, token_ids);
  else
    hipLaunchKernelGGL((token_logprob_kernel<MI355_F16>), grid, block, 0, s, out_logprob, out_lse, logits,
                       logits_stride, vocab, token_ids);
  return check_launch(name);
}
