// SwiGLU gate: out = round(round(silu(gate)) * up) for rows holding gate | up.
// Elementwise and HBM-bound (4 bytes in, 2 out per output element): 16-byte
// vector loads/stores, grid-stride over 8-element vectors.
#include "common.h"

namespace mi355 {

template <typename T>
__global__ __launch_bounds__(256) void silu_and_mul_kernel(uint16_t* __restrict__ out,
                                                           const uint16_t* __restrict__ x, int64_t x_stride,
                                                           int64_t num_tokens, int inter) {
  const int vec_per_row = inter >> 3;
  const int64_t total = num_tokens * vec_per_row;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t row = i / vec_per_row;
    const int v = static_cast<int>(i % vec_per_row);
    const uint16_t* xr = x + row * x_stride;
    float g[8], u[8], y[8];
    unpack8<T>(MI355_LD(uint4, x, xr + 8 * v), g);
    unpack8<T>(MI355_LD(uint4, x, xr + inter + 8 * v), u);
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = silu_mul<T>(g[j], u[j]);
    MI355_ST(uint4, out, out + row * inter + 8 * v, pack8<T>(y));
  }
}

}  // namespace mi355

extern "C" int32_t mi355_silu_and_mul(void* out, const void* x, int64_t x_stride, int64_t num_tokens,
                                      int32_t inter, int32_t dtype, void* stream) {
  using namespace mi355;
  const char* name = "mi355_silu_and_mul";
  MI355_REQUIRE(out && x, MI355_ERR_INVALID, "%s: null pointer", name);
  MI355_REQUIRE(num_tokens >= 0 && num_tokens <= 0x7FFFFFFF, MI355_ERR_INVALID,
                "%s: num_tokens %lld out of range", name, (long long)num_tokens);
  MI355_REQUIRE(inter > 0 && inter % 8 == 0, MI355_ERR_UNSUPPORTED,
                "%s: intermediate size %d must be a positive multiple of 8", name, inter);
  MI355_REQUIRE(x_stride >= 2LL * inter && x_stride % 8 == 0, MI355_ERR_INVALID,
                "%s: x_stride %lld must be >= 2*inter and a multiple of 8", name, (long long)x_stride);
  MI355_REQUIRE(aligned16(out) && aligned16(x), MI355_ERR_INVALID, "%s: pointers must be 16-byte aligned",
                name);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "%s: dtype %d not supported (bf16/fp16 only)", name, dtype);
  if (num_tokens == 0) return MI355_OK;
  int64_t blocks = (num_tokens * (inter >> 3) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == MI355_BF16)
    hipLaunchKernelGGL((silu_and_mul_kernel<BF16>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s,
                       static_cast<uint16_t*>(out), static_cast<const uint16_t*>(x), x_stride, num_tokens,
                       inter);
  else
    hipLaunchKernelGGL((silu_and_mul_kernel<F16>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s,
                       static_cast<uint16_t*>(out), static_cast<const uint16_t*>(x), x_stride, num_tokens,
                       inter);
  return check_launch(name);
}
