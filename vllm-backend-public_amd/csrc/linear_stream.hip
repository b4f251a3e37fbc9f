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
      const u32x4 v = MI355_LDNT(u32x4, w, wp + 32 * s);
      return make_uint4(v.x, v.y, v.z, v.w);
    };
    auto load_x = [&](int t, int s) { return MI355_LD(uint4, x, xp[t] + 32 * s); };

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
      if (n0 + r < N) bv[r] = T::to_f(MI355_LD(uint16_t, bias, bias + (n0 + r)));
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
      MI355_ST(uint2, y, yp, make_uint2(w0, w1));
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n0 + r < N) MI355_ST(uint16_t, y, yp + r, T::from_f(acc[t][r] + bv[r]));
    }
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
  MI355_REQUIRE(k_slices == 0 || k_slices == 1 || k_slices == 2 || k_slices == 4, MI355_ERR_INVALID,
                "%s: k_slices must be 0 (auto), 1, 2 or 4", name);
  if (M == 0) return MI355_OK;
  const int ntiles = (N + 15) / 16;
  const int mt = (M + 15) / 16;
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto* yp = static_cast<uint16_t*>(y);
  auto* xp = static_cast<const uint16_t*>(x);
  auto* wp = static_cast<const uint16_t*>(weight);
  auto* bp = static_cast<const uint16_t*>(bias);
  int ks = k_slices;
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
