// Experiment: one persistent grid (a workgroup per CU) walking the four GEMM phases of a
// decoder layer with grid barriers between them, versus the same phases without barriers,
// and with the next phase's weight ring issued BEFORE the barrier wait.  Timing only: x is
// whatever the buffer holds, results go to a scratch slab.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Phase {
  int tiles;        // 16-row tiles of W
  int kgroups;      // K split
  int slices;       // 1024-wide slices per workgroup task
  int ksteps;       // K / 32 of the whole matrix
  int64_t w_off;    // element offset of the matrix inside a layer's weights
};

constexpr int NW = 8, NT = 512, KSL = 1024, MT = 2, ROWS = 32, U = 4, D = 4, G = KSL / 32 / U;
constexpr int ROWB = KSL * 2 + 32, XBUF = ROWS * ROWB;

__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned& gen, unsigned nwg, int* abort_flag) {
  __syncthreads();
  if (threadIdx.x == 0) {
    gen += 1;
    const unsigned arrived = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    if (arrived == gen * nwg) {
      __hip_atomic_store(&bar[1], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const uint64_t t0 = wall_clock64();
      while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        if (wall_clock64() - t0 > 2000000ull) {  // 20 ms at 100 MHz: give up, everybody leaves
          __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
  }
  __syncthreads();
}

// mode 0: barrier after every phase; 1: barrier + the next phase's ring is issued before the wait; 2: no barriers
__global__ __launch_bounds__(NT) void persist_kernel(const Phase* __restrict__ phases, int nphases, int layers,
                                                     const uint16_t* __restrict__ weights, int64_t layer_elems,
                                                     int pool_layers, const uint16_t* __restrict__ x, int64_t x_stride,
                                                     float* __restrict__ out, unsigned* bar, int* abort_flag, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * XBUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, h = lane >> 4;
  const unsigned nwg = gridDim.x;
  unsigned gen = 0;
  constexpr int NX = ROWS * (KSL / 8) / NT;

  uint4 ring[D][U];
  f32x4 acc[MT];

  // task of this workgroup in phase ph of layer l: (tile group, k group) or none
  auto task = [&](int l, int ph, const uint16_t*& wbase, int& groups_total, int& slice0, int& slices, bool& has) {
    const Phase p = phases[ph];
    const int tgs = (p.tiles + NW - 1) / NW;
    const int vb = blockIdx.x;
    has = vb < tgs * p.kgroups;
    const int tg = has ? vb % tgs : 0, kg = has ? vb / tgs : 0;
    const int tile = min(tg * NW + wave, p.tiles - 1);
    slices = p.slices;
    slice0 = kg * p.slices;
    wbase = weights + static_cast<int64_t>(l % pool_layers) * layer_elems + p.w_off +
            (static_cast<int64_t>(tile) * p.ksteps + static_cast<int64_t>(slice0) * (KSL / 32)) * 512 + lane * 8;
    groups_total = p.slices * G;
  };
  auto load_group = [&](uint4 (&wd)[U], const uint16_t* wbase, int g, int groups_total) {
    const uint16_t* p = wbase + static_cast<int64_t>(min(g, groups_total - 1)) * (U * 512);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + u * 512));
      wd[u] = make_uint4(v.x, v.y, v.z, v.w);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto x_issue = [&](u32x4 (&xs)[NX], int slice) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int v = i * NT + tid;
      const int row = v / (KSL / 8), col = v % (KSL / 8);
      xs[i] = *reinterpret_cast<const u32x4*>(x + static_cast<int64_t>(row) * x_stride + static_cast<int64_t>(slice) * KSL + col * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto x_commit = [&](u32x4 (&xs)[NX], int buf) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int v = i * NT + tid;
      const int row = v / (KSL / 8), col = v % (KSL / 8);
      *reinterpret_cast<u32x4*>(&lds[buf * XBUF + row * ROWB + col * 16]) = xs[i];
    }
  };
  auto compute = [&](uint4 (&wd)[U], int buf, int g) {
    const unsigned char* xb = &lds[buf * XBUF + c * ROWB + g * (U * 64) + 16 * h];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int m = 0; m < MT; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
            __builtin_bit_cast(bf16x8, wd[u]), __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xb + u * 64 + m * 16 * ROWB)),
            acc[m], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };

  const uint16_t* wbase;
  int groups_total, slice0, slices;
  bool has;
  task(0, 0, wbase, groups_total, slice0, slices, has);
  if (has) {
#pragma unroll
    for (int j = 0; j < D - 1; ++j) load_group(ring[j], wbase, j, groups_total);
  }
  const int total = layers * nphases;
  for (int it = 0; it < total; ++it) {
    const int l = it / nphases, ph = it % nphases;
    if (has) {
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      {
        u32x4 xs0[NX];
        x_issue(xs0, slice0);
        x_commit(xs0, 0);
      }
      __syncthreads();
      for (int p = 0; p < slices; ++p) {
        const bool more = p + 1 < slices;
        u32x4 xs[NX];
        x_issue(xs, slice0 + (more ? p + 1 : p));
#pragma unroll
        for (int g = 0; g < G; ++g) {
          load_group(ring[(g + D - 1) % D], wbase, p * G + g + D - 1, groups_total);
          compute(ring[g % D], p & 1, g);
        }
        if (more) {
          x_commit(xs, (p + 1) & 1);
          __syncthreads();
        }
      }
      // partial-sum store (scratch slab, one row of 16 floats per lane pair; timing only)
      float* pp = out + (static_cast<int64_t>(blockIdx.x) * NT + tid) * 8;
      *reinterpret_cast<float4*>(pp) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
      *reinterpret_cast<float4*>(pp + 4) = make_float4(acc[1][0], acc[1][1], acc[1][2], acc[1][3]);
    }
    __syncthreads();   // the x buffers are free again
    if (it + 1 < total) {
      if (mode == 1) {  // next task's ring goes out before we wait
        task((it + 1) / nphases, (it + 1) % nphases, wbase, groups_total, slice0, slices, has);
        if (has) {
#pragma unroll
          for (int j = 0; j < D - 1; ++j) load_group(ring[j], wbase, j, groups_total);
        }
        grid_barrier(bar, gen, nwg, abort_flag);
      } else {
        if (mode == 0) grid_barrier(bar, gen, nwg, abort_flag);
        task((it + 1) / nphases, (it + 1) % nphases, wbase, groups_total, slice0, slices, has);
        if (has) {
#pragma unroll
          for (int j = 0; j < D - 1; ++j) load_group(ring[j], wbase, j, groups_total);
        }
      }
      if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    }
  }
}

extern "C" int exp_persist(const void* phases, int nphases, int layers, const void* weights, int64_t layer_elems,
                           int pool_layers, const void* x, int64_t x_stride, void* out, void* bar, void* abort_flag,
                           int mode, int wgs, void* stream) {
  hipLaunchKernelGGL(persist_kernel, dim3(wgs), dim3(NT), 0, static_cast<hipStream_t>(stream),
                     static_cast<const Phase*>(phases), nphases, layers, static_cast<const uint16_t*>(weights),
                     layer_elems, pool_layers, static_cast<const uint16_t*>(x), x_stride, static_cast<float*>(out),
                     static_cast<unsigned*>(bar), static_cast<int*>(abort_flag), mode);
  return static_cast<int>(hipGetLastError());
}
