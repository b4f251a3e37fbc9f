// Experiment: throughput of fp8 -> bf16 operand conversion per wave (what a W8A16 GEMM pays per k-step).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) float f2;

__device__ __forceinline__ uint32_t bf16pair(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  bf2 v = {static_cast<__bf16>(a), static_cast<__bf16>(b)};
  return __builtin_bit_cast(uint32_t, v);
}

// mode 0: v_cvt_pk_f32_fp8 + v_cvt_pk_bf16_f32; mode 1: integer bit placement (normals only)
template <int MODE>
__global__ __launch_bounds__(512) void cvt_kernel(uint32_t* out, const uint32_t* in, int iters) {
  uint32_t lo = in[threadIdx.x], hi = in[threadIdx.x + 512];
  uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      const f2 a = __builtin_amdgcn_cvt_pk_f32_fp8(lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(lo, true);
      const f2 c = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
      acc0 ^= bf16pair(a[0], a[1]); acc1 ^= bf16pair(b[0], b[1]);
      acc2 ^= bf16pair(c[0], c[1]); acc3 ^= bf16pair(d[0], d[1]);
    } else {
      // bytes b0 b1 of a 16-bit half -> two bf16: ((b & 0x7f) << 4 | (b & 0x80) << 8) + 0x3c00
      const uint32_t p0 = __builtin_amdgcn_perm(0u, lo, 0x0c010c00u), p1 = __builtin_amdgcn_perm(0u, lo, 0x0c030c02u);
      const uint32_t p2 = __builtin_amdgcn_perm(0u, hi, 0x0c010c00u), p3 = __builtin_amdgcn_perm(0u, hi, 0x0c030c02u);
      acc0 ^= (((p0 & 0x007f007fu) << 4) | ((p0 & 0x00800080u) << 8)) + 0x3c003c00u;
      acc1 ^= (((p1 & 0x007f007fu) << 4) | ((p1 & 0x00800080u) << 8)) + 0x3c003c00u;
      acc2 ^= (((p2 & 0x007f007fu) << 4) | ((p2 & 0x00800080u) << 8)) + 0x3c003c00u;
      acc3 ^= (((p3 & 0x007f007fu) << 4) | ((p3 & 0x00800080u) << 8)) + 0x3c003c00u;
    }
    lo = lo * 1664525u + 1013904223u;   // new data every iteration (2 VALU ops of overhead)
    hi ^= lo;
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc0 ^ acc1 ^ acc2 ^ acc3;
}

extern "C" int exp_cvt(void* out, const void* in, int iters, int mode, void* stream) {
  if (mode == 0) hipLaunchKernelGGL(cvt_kernel<0>, dim3(256), dim3(512), 0, static_cast<hipStream_t>(stream), static_cast<uint32_t*>(out), static_cast<const uint32_t*>(in), iters);
  else hipLaunchKernelGGL(cvt_kernel<1>, dim3(256), dim3(512), 0, static_cast<hipStream_t>(stream), static_cast<uint32_t*>(out), static_cast<const uint32_t*>(in), iters);
  return static_cast<int>(hipGetLastError());
}
