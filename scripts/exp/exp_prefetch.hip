// Experiment: does a read pass over a weight matrix leave it in the memory-side
// cache so that the GEMM that follows streams it faster than from HBM?
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void prefetch_kernel(const u32x4* __restrict__ p, int64_t n16, unsigned* sink) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const u32x4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
    acc ^= a ^ b ^ c ^ d;
  }
  for (; i < n16; i += stride) acc ^= p[i];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) *sink = 1;  // practically never; keeps the loads alive
}

extern "C" int exp_prefetch(const void* p, int64_t bytes, int blocks, void* sink, void* stream) {
  hipLaunchKernelGGL(prefetch_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const u32x4*>(p), bytes / 16, static_cast<unsigned*>(sink));
  return static_cast<int>(hipGetLastError());
}
