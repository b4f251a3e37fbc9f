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

// ---- consumers of the fp32 partial-sum slabs [num_slabs][M][N] a K-split GEMM leaves behind, for the
// call sites whose next step needs a 16-bit tensor: the all-reduce of a tensor-parallel rank's
// o / down projection, and the SwiGLU gate between gate_up and down.  The slabs are added in slab
// order (a fixed order: the result does not depend on timing), every load of a thread issued
// before the first use.  HBM/L2-bound: num_slabs * 4 bytes in, 2 (SwiGLU: 1) out per element.
template <typename T, int S>
__global__ __launch_bounds__(256) void sum_slabs_kernel(uint16_t* __restrict__ y, int64_t y_stride,
                                                        const float* __restrict__ partial,
                                                        const uint16_t* __restrict__ bias, int M, int N) {
  const int vec_per_row = N >> 3;
  const int64_t total = static_cast<int64_t>(M) * vec_per_row;
  const int64_t slab = static_cast<int64_t>(M) * N;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t row = i / vec_per_row;
    const int v = static_cast<int>(i % vec_per_row);
    const float* p = partial + row * N + 8 * v;
    float4 a[S], b[S];
#pragma unroll
    for (int g = 0; g < S; ++g) {
      a[g] = MI355_LD(float4, partial, p + g * slab);
      b[g] = MI355_LD(float4, partial, p + g * slab + 4);
    }
    float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < S; ++g) {
      f[0] += a[g].x, f[1] += a[g].y, f[2] += a[g].z, f[3] += a[g].w;
      f[4] += b[g].x, f[5] += b[g].y, f[6] += b[g].z, f[7] += b[g].w;
    }
    if (bias != nullptr) {
      float bb[8];
      unpack8<T>(MI355_LD(uint4, bias, bias + 8 * v), bb);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += bb[j];
    }
    MI355_ST(uint4, y, y + row * y_stride + 8 * v, pack8<T>(f));
  }
}

// SwiGLU from slabs of a gate_up projection whose 16-column tiles alternate gate / up (the order
// mi355_tile_weights builds with interleave_halves): slab columns [32u, 32u + 16) are the gate of
// output columns [16u, 16u + 16), [32u + 16, 32u + 32) their up values.
//   out = round(round(silu(round(sum gate))) * round(sum up))     -- what the GEMM's own SwiGLU epilogue computes
template <typename T, int S>
__global__ __launch_bounds__(256) void silu_and_mul_slabs_kernel(uint16_t* __restrict__ out, int64_t out_stride,
                                                                 const float* __restrict__ partial, int M, int N) {
  const int F = N >> 1;
  const int vec_per_row = F >> 3;
  const int64_t total = static_cast<int64_t>(M) * vec_per_row;
  const int64_t slab = static_cast<int64_t>(M) * N;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t row = i / vec_per_row;
    const int v = static_cast<int>(i % vec_per_row);          // output columns [8v, 8v + 8)
    const float* pg = partial + row * N + (v >> 1) * 32 + (v & 1) * 8;
    float4 g0[S], g1[S], u0[S], u1[S];
#pragma unroll
    for (int g = 0; g < S; ++g) {
      g0[g] = MI355_LD(float4, partial, pg + g * slab);
      g1[g] = MI355_LD(float4, partial, pg + g * slab + 4);
      u0[g] = MI355_LD(float4, partial, pg + g * slab + 16);
      u1[g] = MI355_LD(float4, partial, pg + g * slab + 20);
    }
    float gs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, us[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < S; ++g) {
      gs[0] += g0[g].x, gs[1] += g0[g].y, gs[2] += g0[g].z, gs[3] += g0[g].w;
      gs[4] += g1[g].x, gs[5] += g1[g].y, gs[6] += g1[g].z, gs[7] += g1[g].w;
      us[0] += u0[g].x, us[1] += u0[g].y, us[2] += u0[g].z, us[3] += u0[g].w;
      us[4] += u1[g].x, us[5] += u1[g].y, us[6] += u1[g].z, us[7] += u1[g].w;
    }
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = silu_mul<T>(round_f<T>(gs[j]), round_f<T>(us[j]));
    MI355_ST(uint4, out, out + row * out_stride + 8 * v, pack8<T>(o));
  }
}

}  // namespace mi355

#define MI355_SLAB_CASES(KERNEL, ...)                                                                              \
  switch (num_slabs) {                                                                                             \
    case 1: hipLaunchKernelGGL((KERNEL<TT, 1>), grid, dim3(256), 0, s, __VA_ARGS__); break;                        \
    case 2: hipLaunchKernelGGL((KERNEL<TT, 2>), grid, dim3(256), 0, s, __VA_ARGS__); break;                        \
    case 3: hipLaunchKernelGGL((KERNEL<TT, 3>), grid, dim3(256), 0, s, __VA_ARGS__); break;                        \
    case 4: hipLaunchKernelGGL((KERNEL<TT, 4>), grid, dim3(256), 0, s, __VA_ARGS__); break;                        \
    case 5: hipLaunchKernelGGL((KERNEL<TT, 5>), grid, dim3(256), 0, s, __VA_ARGS__); break;                        \
    case 6: hipLaunchKernelGGL((KERNEL<TT, 6>), grid, dim3(256), 0, s, __VA_ARGS__); break;                        \
    case 7: hipLaunchKernelGGL((KERNEL<TT, 7>), grid, dim3(256), 0, s, __VA_ARGS__); break;                        \
    default: hipLaunchKernelGGL((KERNEL<TT, 8>), grid, dim3(256), 0, s, __VA_ARGS__); break;                       \
  }

extern "C" int32_t mi355_sum_slabs(void* y, int64_t y_stride, const float* partial, int32_t num_slabs, const void* bias,
                                   int32_t M, int32_t N, int32_t dtype, void* stream) {
  using namespace mi355;
  const char* name = "mi355_sum_slabs";
  MI355_REQUIRE(y && partial, MI355_ERR_INVALID, "%s: null pointer", name);
  MI355_REQUIRE(num_slabs >= 1 && num_slabs <= 8, MI355_ERR_UNSUPPORTED, "%s: num_slabs %d outside [1, 8]", name, num_slabs);
  MI355_REQUIRE(M >= 0 && N > 0 && N % 8 == 0, MI355_ERR_UNSUPPORTED, "%s: N %d must be a positive multiple of 8 (M %d)", name, N, M);
  MI355_REQUIRE(y_stride >= N && y_stride % 8 == 0 && aligned16(y) && aligned16(partial) && (bias == nullptr || aligned16(bias)),
                MI355_ERR_INVALID, "%s: pointers must be 16-byte aligned, y_stride >= N and a multiple of 8", name);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED, "%s: dtype %d not supported (bf16/fp16 only)",
                name, dtype);
  if (M == 0) return MI355_OK;
  int64_t blocks = (static_cast<int64_t>(M) * (N >> 3) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  const dim3 grid(static_cast<unsigned>(blocks));
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto* yp = static_cast<uint16_t*>(y);
  auto* bp = static_cast<const uint16_t*>(bias);
  if (dtype == MI355_BF16) {
    using TT = BF16;
    MI355_SLAB_CASES(sum_slabs_kernel, yp, y_stride, partial, bp, M, N)
  } else {
    using TT = F16;
    MI355_SLAB_CASES(sum_slabs_kernel, yp, y_stride, partial, bp, M, N)
  }
  return check_launch(name);
}

extern "C" int32_t mi355_silu_and_mul_slabs(void* out, int64_t out_stride, const float* partial, int32_t num_slabs,
                                            int32_t M, int32_t N, int32_t dtype, void* stream) {
  using namespace mi355;
  const char* name = "mi355_silu_and_mul_slabs";
  MI355_REQUIRE(out && partial, MI355_ERR_INVALID, "%s: null pointer", name);
  MI355_REQUIRE(num_slabs >= 1 && num_slabs <= 8, MI355_ERR_UNSUPPORTED, "%s: num_slabs %d outside [1, 8]", name, num_slabs);
  MI355_REQUIRE(M >= 0 && N > 0 && N % 32 == 0, MI355_ERR_UNSUPPORTED,
                "%s: N %d must be a positive multiple of 32 (gate / up tiles of 16 columns alternate; M %d)", name, N, M);
  MI355_REQUIRE(out_stride >= N / 2 && out_stride % 8 == 0 && aligned16(out) && aligned16(partial), MI355_ERR_INVALID,
                "%s: pointers must be 16-byte aligned, out_stride >= N / 2 and a multiple of 8", name);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED, "%s: dtype %d not supported (bf16/fp16 only)",
                name, dtype);
  if (M == 0) return MI355_OK;
  int64_t blocks = (static_cast<int64_t>(M) * (N >> 4) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  const dim3 grid(static_cast<unsigned>(blocks));
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto* op = static_cast<uint16_t*>(out);
  if (dtype == MI355_BF16) {
    using TT = BF16;
    MI355_SLAB_CASES(silu_and_mul_slabs_kernel, op, out_stride, partial, M, N)
  } else {
    using TT = F16;
    MI355_SLAB_CASES(silu_and_mul_slabs_kernel, op, out_stride, partial, M, N)
  }
  return check_launch(name);
}

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
