// Decode-time linear layer  y[M, N] = x[M, K] . W[N, K]^T (+ bias),  M <= 128.
//
// At decode the weight matrix is read once and never reused, so the layer is a
// pure HBM stream of W (N*K*2 bytes); x (M*K*2 bytes) is re-read from L2 by every
// wave.  Each WAVE owns one 16-row tile of W and one slice of K and never
// synchronises in its main loop:
//   D[n, m] += W[n, k] . x[m, k]      v_mfma_f32_16x16x32
//     A = W tile: lane (i = lane & 15, h = lane >> 4) loads the 16 bytes of row
//         n0 + i at k + 8h straight from HBM into the operand registers (one wave
//         instruction = 16 rows x 64 contiguous bytes), non-temporal so the stream
//         does not evict x from L2; no LDS staging (nothing is shared)
//     B = x^T tile(s): lane (m = lane & 15, h) loads x[16t + m][k + 8h ..], an L2 hit
//   Loads of the next group of k-steps are issued before the MFMAs of the current
//   one (register double buffer), and several waves per SIMD keep >64 KB per CU in
//   flight.
// A workgroup is 4 waves: KS of them split K for the same row tile (partial sums
// meet in LDS, summed in a fixed order -> bit-reproducible), 4/KS row tiles per
// workgroup.  Algorithmic bytes per launch: (N*K + M*K + M*N) * 2.
#include "common.h"

namespace mi355 {

constexpr int kLinThreads = 256;
constexpr int kLinU = 4;  // k-steps (of 32) per register group

template <typename T, int MT, int KS>
__global__ __launch_bounds__(kLinThreads) void linear_stream_kernel(
    uint16_t* __restrict__ y, const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
    const uint16_t* __restrict__ bias, int M, int N, int K, int64_t x_stride, int64_t y_stride) {
  constexpr int TILES = 4 / KS;  // row tiles per workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, h = lane >> 4;
  const int tile = blockIdx.x * TILES + wave / KS;
  const int slice = wave % KS;
  const int ntiles = (N + 15) >> 4;
  const bool active = tile < ntiles;

  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (active) {
    const int row = min(tile * 16 + c, N - 1);
    const uint16_t* wp = w + static_cast<int64_t>(row) * K + 8 * h;
    const uint16_t* xp[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) xp[t] = x + static_cast<int64_t>(min(16 * t + c, M - 1)) * x_stride + 8 * h;

    const int steps = K >> 5;
    const int s0 = static_cast<int>(static_cast<int64_t>(slice) * steps / KS);
    const int s1 = static_cast<int>(static_cast<int64_t>(slice + 1) * steps / KS);
    const int groups = (s1 - s0) / kLinU;

    auto load_w = [&](int s) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + 32 * s));
      return make_uint4(v.x, v.y, v.z, v.w);
    };
    auto load_x = [&](int t, int s) { return *reinterpret_cast<const uint4*>(xp[t] + 32 * s); };

    // Two register groups in ping-pong; the prefetch is unconditional (the last
    // iteration re-loads its own group) so that the compiler can count its waits
    // instead of draining the queue at a branch join.
    uint4 wa[kLinU], xa[kLinU][MT], wb[kLinU], xb[kLinU][MT];
    auto load_group = [&](uint4 (&wd)[kLinU], uint4 (&xd)[kLinU][MT], int g) {
#pragma unroll
      for (int u = 0; u < kLinU; ++u) {
        wd[u] = load_w(s0 + g * kLinU + u);
#pragma unroll
        for (int t = 0; t < MT; ++t) xd[u][t] = load_x(t, s0 + g * kLinU + u);
      }
    };
    auto compute = [&](uint4 (&wd)[kLinU], uint4 (&xd)[kLinU][MT]) {
#pragma unroll
      for (int u = 0; u < kLinU; ++u)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = T::mfma(wd[u], xd[u][t], acc[t]);
    };
    if (groups > 0) {
      load_group(wa, xa, 0);
      for (int g = 0; g + 1 < groups; g += 2) {
        load_group(wb, xb, g + 1);
        compute(wa, xa);
        load_group(wa, xa, min(g + 2, groups - 1));
        compute(wb, xb);
      }
      if (groups & 1) compute(wa, xa);
    }
    for (int s = s0 + groups * kLinU; s < s1; ++s) {  // remainder steps
      const uint4 a = load_w(s);
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = T::mfma(a, load_x(t, s), acc[t]);
    }
  }

  // lane holds D[n = 4h + r][m = c] of each m-tile
  __shared__ float red[KS > 1 ? 4 : 1][MT][4][64];
  if (KS > 1) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][t][r][lane] = acc[t][r];
    __syncthreads();
    if (slice != 0) return;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sum = acc[t][r];
#pragma unroll
        for (int k = 1; k < KS; ++k) sum += red[wave + k][t][r][lane];
        acc[t][r] = sum;
      }
  }
  if (!active) return;
  const int n0 = tile * 16 + 4 * h;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias != nullptr) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n0 + r < N) bv[r] = T::to_f(bias[n0 + r]);
  }
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = 16 * t + c;
    if (m >= M) continue;
    uint16_t* yp = y + static_cast<int64_t>(m) * y_stride + n0;
    if (n0 + 3 < N) {
      const uint32_t w0 = static_cast<uint32_t>(T::from_f(acc[t][0] + bv[0])) |
                          (static_cast<uint32_t>(T::from_f(acc[t][1] + bv[1])) << 16);
      const uint32_t w1 = static_cast<uint32_t>(T::from_f(acc[t][2] + bv[2])) |
                          (static_cast<uint32_t>(T::from_f(acc[t][3] + bv[3])) << 16);
      *reinterpret_cast<uint2*>(yp) = make_uint2(w0, w1);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n0 + r < N) yp[r] = T::from_f(acc[t][r] + bv[r]);
    }
  }
}


// ---------------------------------------------------------------------------
// Variant for wide layers: one LOADER wave per workgroup stages x through LDS.
//
// Re-reading x from L2 by every wave costs the weight stream dearly (measured:
// two fragment-shaped L2 loads per weight load halve the stream rate).  Here a
// workgroup is CW compute waves + 1 loader wave.  Compute wave w owns the 16-row
// tile blockIdx.x * CW + w over the workgroup's K range and issues ONLY weight
// loads, so its in-order memory counter never waits on anything else; the loader
// wave copies x in 256-wide K chunks (full 128-byte lines) into a double-buffered
// LDS image one chunk ahead; one workgroup barrier per chunk.  B operands are
// ds_read_b128 from rows padded by 32 bytes (conflict-free for the 16-lane
// groups).  x traffic from L2 drops to M / (16 * CW) of the weight bytes.
// gridDim.y > 1 splits K across workgroups; each writes an fp32 partial slab
// [split][M][N] that the consumer kernel sums in a fixed order.
// ---------------------------------------------------------------------------
constexpr int kLdsKC = 256;                 // K elements per chunk
constexpr int kLdsRowB = kLdsKC * 2 + 32;   // padded LDS row, bytes

template <typename T, int MT, int CW>
__global__ __launch_bounds__((CW + 1) * 64) void linear_lds_kernel(
    uint16_t* __restrict__ y, float* __restrict__ partial, const uint16_t* __restrict__ x,
    const uint16_t* __restrict__ w, const uint16_t* __restrict__ bias, int M, int N, int K,
    int64_t x_stride, int64_t y_stride) {
  constexpr int ROWS = MT * 16;
  constexpr int U = kLdsKC / 64;  // k-steps per register group (two groups per chunk)
  __shared__ __attribute__((aligned(16))) unsigned char xl[2][ROWS * kLdsRowB];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, h = lane >> 4;
  const int splits = gridDim.y;
  const int chunks_total = K / kLdsKC;
  const int ch0 = static_cast<int>(static_cast<int64_t>(blockIdx.y) * chunks_total / splits);
  const int ch1 = static_cast<int>(static_cast<int64_t>(blockIdx.y + 1) * chunks_total / splits);
  const int chunks = ch1 - ch0;
  // Workgroups walk K in rotated order (start chunk = blockIdx.x mod chunks): at any
  // moment the chip then touches all of x and all K offsets of W, not one hot column.
  const int rot = chunks > 0 ? static_cast<int>(blockIdx.x % chunks) : 0;
  auto chunk_at = [&](int i) { const int j = i + rot; return j >= chunks ? j - chunks : j; };

  if (wave == CW) {
    // ---- loader wave: chunk i -> xl[i & 1], one chunk ahead of the compute waves
    constexpr int VEC = ROWS * (kLdsKC / 8) / 64;  // 16-byte vectors per lane per chunk
    constexpr int PASS = VEC > 16 ? 16 : VEC;      // at most 64 staging registers
    auto copy_chunk = [&](int ch, int buf) {
#pragma unroll
      for (int p0 = 0; p0 < VEC; p0 += PASS) {
        uint4 st[PASS];
#pragma unroll
        for (int i = 0; i < PASS; ++i) {
          const int v = (p0 + i) * 64 + lane;
          const int row = v / (kLdsKC / 8), col = v % (kLdsKC / 8);
          st[i] = *reinterpret_cast<const uint4*>(x + static_cast<int64_t>(min(row, M - 1)) * x_stride +
                                                  static_cast<int64_t>(ch) * kLdsKC + col * 8);
        }
#pragma unroll
        for (int i = 0; i < PASS; ++i) {
          const int v = (p0 + i) * 64 + lane;
          const int row = v / (kLdsKC / 8), col = v % (kLdsKC / 8);
          *reinterpret_cast<uint4*>(&xl[buf][row * kLdsRowB + col * 16]) = st[i];
        }
      }
    };
    copy_chunk(ch0 + chunk_at(0), 0);
    __syncthreads();
    for (int i = 0; i < chunks; ++i) {
      if (i + 1 < chunks) copy_chunk(ch0 + chunk_at(i + 1), (i + 1) & 1);
      __syncthreads();
    }
    return;
  }

  // ---- compute waves
  const int tile = blockIdx.x * CW + wave;
  const int ntiles = (N + 15) >> 4;
  const bool active = tile < ntiles;
  const int row = min(tile * 16 + c, N - 1);
  const uint16_t* wp = w + static_cast<int64_t>(row) * K + static_cast<int64_t>(ch0) * kLdsKC + 8 * h;
  auto load_w = [&](int s) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + 32 * s));
    return make_uint4(v.x, v.y, v.z, v.w);
  };
  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 wa[U], wb[U];
  auto load_group = [&](uint4 (&wd)[U], int s0) {
#pragma unroll
    for (int u = 0; u < U; ++u) wd[u] = load_w(s0 + u);
  };
  auto compute = [&](uint4 (&wd)[U], int buf, int koff) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned char* base = &xl[buf][c * kLdsRowB + (koff + u) * 64 + 16 * h];
#pragma unroll
      for (int t = 0; t < MT; ++t)
        acc[t] = T::mfma(wd[u], *reinterpret_cast<const uint4*>(base + t * 16 * kLdsRowB), acc[t]);
    }
  };
  load_group(wa, chunk_at(0) * 2 * U);
  load_group(wb, chunk_at(0) * 2 * U + U);
  __syncthreads();
  for (int i = 0; i < chunks; ++i) {
    const int nxt = chunk_at(min(i + 1, chunks - 1)) * 2 * U;  // last iteration re-loads its own chunk
    compute(wa, i & 1, 0);
    load_group(wa, nxt);
    compute(wb, i & 1, U);
    load_group(wb, nxt + U);
    __syncthreads();
  }
  if (!active) return;
  const int n0 = tile * 16 + 4 * h;
  if (splits > 1) {
    float* pp = partial + static_cast<int64_t>(blockIdx.y) * M * N;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int m = 16 * t + c;
      if (m >= M) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n0 + r < N) pp[static_cast<int64_t>(m) * N + n0 + r] = acc[t][r];
    }
    return;
  }
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias != nullptr) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n0 + r < N) bv[r] = T::to_f(bias[n0 + r]);
  }
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = 16 * t + c;
    if (m >= M) continue;
    uint16_t* yp = y + static_cast<int64_t>(m) * y_stride + n0;
    if (n0 + 3 < N) {
      const uint32_t w0 = static_cast<uint32_t>(T::from_f(acc[t][0] + bv[0])) |
                          (static_cast<uint32_t>(T::from_f(acc[t][1] + bv[1])) << 16);
      const uint32_t w1 = static_cast<uint32_t>(T::from_f(acc[t][2] + bv[2])) |
                          (static_cast<uint32_t>(T::from_f(acc[t][3] + bv[3])) << 16);
      *reinterpret_cast<uint2*>(yp) = make_uint2(w0, w1);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n0 + r < N) yp[r] = T::from_f(acc[t][r] + bv[r]);
    }
  }
}

template <typename T, int MT>
static void launch_lds(int cw, dim3 grid, hipStream_t s, uint16_t* y, float* partial, const uint16_t* x,
                       const uint16_t* w, const uint16_t* bias, int M, int N, int K, int64_t xs, int64_t ys) {
  if (cw == 8)
    hipLaunchKernelGGL((linear_lds_kernel<T, MT, 8>), grid, dim3(9 * 64), 0, s, y, partial, x, w, bias, M, N, K, xs, ys);
  else
    hipLaunchKernelGGL((linear_lds_kernel<T, MT, 4>), grid, dim3(5 * 64), 0, s, y, partial, x, w, bias, M, N, K, xs, ys);
}

template <typename T>
static void launch_lds_mt(int mt, int cw, dim3 grid, hipStream_t s, uint16_t* y, float* partial,
                          const uint16_t* x, const uint16_t* w, const uint16_t* bias, int M, int N, int K,
                          int64_t xs, int64_t ys) {
  switch (mt) {
    case 1: launch_lds<T, 1>(cw, grid, s, y, partial, x, w, bias, M, N, K, xs, ys); break;
    case 2: launch_lds<T, 2>(cw, grid, s, y, partial, x, w, bias, M, N, K, xs, ys); break;
    case 3: case 4: launch_lds<T, 4>(cw, grid, s, y, partial, x, w, bias, M, N, K, xs, ys); break;
    default: launch_lds<T, 8>(cw, grid, s, y, partial, x, w, bias, M, N, K, xs, ys); break;
  }
}

template <typename T, int MT>
static void launch_ks(int ks, dim3 grid, hipStream_t s, uint16_t* y, const uint16_t* x, const uint16_t* w,
                      const uint16_t* bias, int M, int N, int K, int64_t xs, int64_t ys) {
  dim3 block(kLinThreads);
  if (ks == 1)
    hipLaunchKernelGGL((linear_stream_kernel<T, MT, 1>), grid, block, 0, s, y, x, w, bias, M, N, K, xs, ys);
  else if (ks == 2)
    hipLaunchKernelGGL((linear_stream_kernel<T, MT, 2>), grid, block, 0, s, y, x, w, bias, M, N, K, xs, ys);
  else
    hipLaunchKernelGGL((linear_stream_kernel<T, MT, 4>), grid, block, 0, s, y, x, w, bias, M, N, K, xs, ys);
}

template <typename T>
static void launch_mt(int mt, int ks, dim3 grid, hipStream_t s, uint16_t* y, const uint16_t* x,
                      const uint16_t* w, const uint16_t* bias, int M, int N, int K, int64_t xs, int64_t ys) {
  switch (mt) {
    case 1: launch_ks<T, 1>(ks, grid, s, y, x, w, bias, M, N, K, xs, ys); break;
    case 2: launch_ks<T, 2>(ks, grid, s, y, x, w, bias, M, N, K, xs, ys); break;
    case 3: case 4: launch_ks<T, 4>(ks, grid, s, y, x, w, bias, M, N, K, xs, ys); break;
    default: launch_ks<T, 8>(ks, grid, s, y, x, w, bias, M, N, K, xs, ys); break;
  }
}

}  // namespace mi355

extern "C" int32_t mi355_linear_decode(void* y, int64_t y_stride, const void* x, int64_t x_stride,
                                       const void* weight, const void* bias, int32_t M, int32_t N,
                                       int32_t K, int32_t k_slices, int32_t dtype, void* stream) {
  using namespace mi355;
  const char* name = "mi355_linear_decode";
  MI355_REQUIRE(y && x && weight, MI355_ERR_INVALID, "%s: null pointer", name);
  MI355_REQUIRE(M >= 0 && M <= 128, MI355_ERR_UNSUPPORTED, "%s: M %d outside [0, 128] (decode batches only)",
                name, M);
  MI355_REQUIRE(N > 0 && K > 0 && K % 32 == 0, MI355_ERR_UNSUPPORTED,
                "%s: K %d must be a positive multiple of 32 (N %d)", name, K, N);
  MI355_REQUIRE(x_stride >= K && x_stride % 8 == 0 && y_stride >= N && y_stride % 4 == 0, MI355_ERR_INVALID,
                "%s: x_stride/y_stride invalid", name);
  MI355_REQUIRE(aligned16(x) && aligned16(weight) && (reinterpret_cast<uintptr_t>(y) & 7u) == 0,
                MI355_ERR_INVALID, "%s: x/weight must be 16-byte aligned, y 8-byte aligned", name);
  MI355_REQUIRE(k_slices == 0 || k_slices == 1 || k_slices == 2 || k_slices == 4 || k_slices == 101 ||
                    k_slices == 102,
                MI355_ERR_INVALID, "%s: k_slices must be 0 (auto), 1, 2, 4, or 101/102 (LDS-staged x, 4/8 compute waves)",
                name);
  MI355_REQUIRE(k_slices < 100 || K % 256 == 0, MI355_ERR_UNSUPPORTED,
                "%s: the LDS-staged variant needs K %% 256 == 0 (K %d)", name, K);
  if (M == 0) return MI355_OK;
  const int ntiles = (N + 15) / 16;
  const int mt = (M + 15) / 16;
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto* yp = static_cast<uint16_t*>(y);
  auto* xp = static_cast<const uint16_t*>(x);
  auto* wp = static_cast<const uint16_t*>(weight);
  auto* bp = static_cast<const uint16_t*>(bias);
  int ks = k_slices;
  if (ks >= 100) {
    const int cw = ks == 102 ? 8 : 4;
    dim3 grid((ntiles + cw - 1) / cw, 1);
    if (dtype == MI355_BF16)
      launch_lds_mt<BF16>(mt, cw, grid, s, yp, nullptr, xp, wp, bp, M, N, K, x_stride, y_stride);
    else
      launch_lds_mt<F16>(mt, cw, grid, s, yp, nullptr, xp, wp, bp, M, N, K, x_stride, y_stride);
    return check_launch(name);
  }
  if (ks == 0) {  // aim at >= 16 waves per CU (256 CUs), but keep >= 8 k-steps per wave
    ks = 1;
    while (ks < 4 && ntiles * ks < 4096 && (K / 32) / (ks * 2) >= 8) ks *= 2;
  }
  const int tiles_per_wg = 4 / ks;
  dim3 grid((ntiles + tiles_per_wg - 1) / tiles_per_wg);
  if (dtype == MI355_BF16)
    launch_mt<BF16>(mt, ks, grid, s, yp, xp, wp, bp, M, N, K, x_stride, y_stride);
  else
    launch_mt<F16>(mt, ks, grid, s, yp, xp, wp, bp, M, N, K, x_stride, y_stride);
  return check_launch(name);
}
