// Experiment: latency of a grid barrier over 256 resident workgroups.
//   mode 0: one device-scope counter + device-scope flag polled by everybody
//   mode 1: two levels: a counter/flag pair per XCD in that XCD's L2 (plain L2 atomics, polled from L2),
//           the last arriver of each XCD takes part in the device-scope step
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xF;
}

// bar layout (uint32): [0] global count, [1] global flag, [64 + 32*x] XCD x count, [64 + 32*x + 16] XCD x flag
__global__ __launch_bounds__(256) void barrier_kernel(unsigned* bar, int iters, int mode, int* abort_flag, unsigned* xcd_hist,
                                                      unsigned long long* cycles) {
  const unsigned nwg = gridDim.x;
  const unsigned xcd = xcc_id();
  __shared__ unsigned per_xcd;
  if (threadIdx.x == 0) {
    atomicAdd(&xcd_hist[xcd], 1u);
    per_xcd = 0;
  }
  __syncthreads();
  unsigned gen = 0;
  const uint64_t t_begin = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    if (threadIdx.x == 0) {
      gen += 1;
      bool give_up = false;
      if (mode == 0) {
        const unsigned a = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        if (a == gen * nwg) {
          __hip_atomic_store(&bar[1], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          const uint64_t t0 = wall_clock64();
          while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
            if (wall_clock64() - t0 > 2000000ull) { give_up = true; break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
      } else {
        unsigned* xc = bar + 64 + 32 * xcd;
        unsigned* xf = xc + 16;
        // arrivals inside the XCD: atomics resolved in this XCD's L2 (workgroup scope = no cache bypass)
        const unsigned a = __hip_atomic_fetch_add(xc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1;
        const unsigned members = nwg / 8;   // workgroups are dealt round-robin to the 8 XCDs
        const uint64_t t0 = wall_clock64();
        if (a == gen * members) {
          const unsigned g = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
          if (g == gen * 8) {
            __hip_atomic_store(&bar[1], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
              if (wall_clock64() - t0 > 2000000ull) { give_up = true; break; }
            }
          }
          __hip_atomic_store(xf, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          while (__hip_atomic_load(xf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < gen) {
            if (wall_clock64() - t0 > 2000000ull) { give_up = true; break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
      }
      if (give_up) *abort_flag = 1;
    }
    __syncthreads();
    if (*reinterpret_cast<volatile int*>(abort_flag)) break;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *cycles = wall_clock64() - t_begin;
}

extern "C" int exp_barrier(void* bar, int iters, int mode, void* abort_flag, void* hist, void* cycles, int wgs, void* stream) {
  hipLaunchKernelGGL(barrier_kernel, dim3(wgs), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<unsigned*>(bar),
                     iters, mode, static_cast<int*>(abort_flag), static_cast<unsigned*>(hist),
                     static_cast<unsigned long long*>(cycles));
  return static_cast<int>(hipGetLastError());
}
