// Decode-time linear layer on FRAGMENT-TILED weights:  y[M, N] = x[M, K] . W[N, K]^T, M <= 32.
//
// Weight layout (built once at load time, mi355_tile_weights): W is cut into
// blocks of 16 rows x 32 k; block (tile, kstep) is 1 KB stored as
// [lane 0..63][8 elements] with lane (i = lane & 15, h = lane >> 4) holding
// W[16*tile + i][32*kstep + 8h .. +8], blocks ordered [tile][kstep].  One wave
// load instruction therefore reads 1 KB of CONTIGUOUS memory that is already the
// v_mfma_f32_16x16x32 A operand, and a wave walking k reads one contiguous
// stream -- the access shape that reaches the HBM copy rate on this chip
// (fragment loads from row-major weights touch 16 rows x 64 B per instruction
// and stop ~15 % lower).
//
// Work decomposition.  A workgroup is NW waves (4 or 8) and owns NW consecutive
// 16-row tiles x one K group; WAVE w streams tile w as one contiguous byte stream
// (32 KB per 1024 of K), so nothing is ever summed across waves.  x is staged
// through LDS one KSL-wide slice (a "phase") at a time, double buffered: the
// global loads of slice p+1 are issued at the start of phase p and written to the
// other LDS buffer at its end, one workgroup barrier per phase.  Weight loads run
// through a ring of 4 register groups of 4 k-steps (4 KB each), three groups
// ahead of the MFMAs; scheduling barriers pin loads where they are written (the
// compiler otherwise sinks every load down to its use and the wave ends up
// waiting for one load at a time).  B operands are ds_read_b128 from LDS rows
// padded by 32 bytes (conflict-free).  Epilogues:
//   EPI_STORE     y = round(sum + bias)
//   EPI_SILU_MUL  tiles alternate gate/up rows of the same 16 columns (waves 2u,
//                 2u+1): act = round(round(silu(round(gate))) * round(up))
//   EPI_PARTIAL   fp32 partial sums to slab [kgroup][M][N] (K split over
//                 workgroups; the consumer kernel adds the slabs in order)
// HBM bytes: N*K*2 once; x is read from L2 once per workgroup.
#include "common.h"

namespace mi355 {

constexpr int kTU = 4;  // k-steps per register group

enum { EPI_STORE = 0, EPI_SILU_MUL = 1, EPI_PARTIAL = 2 };

template <typename T, int MT, int NW, int KSL, int EPI>
__global__ __launch_bounds__(NW * 64) void linear_tiled_kernel(
    uint16_t* __restrict__ y, float* __restrict__ partial, const uint16_t* __restrict__ x,
    const uint16_t* __restrict__ wt, const uint16_t* __restrict__ bias, int M, int N, int K,
    int64_t x_stride, int64_t y_stride, int tiles_total, int phases_per_group) {
  constexpr int ROWS = MT * 16;
  constexpr int U = kTU;
  constexpr int NT = NW * 64;
  constexpr int ROWB = KSL * 2 + 32;       // padded LDS row (bytes)
  constexpr int XBUF = ROWS * ROWB;
  constexpr int G = KSL / 32 / U;          // register groups per phase (8 or 4)
  constexpr int D = 4;                     // ring depth
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * XBUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, h = lane >> 4;
  const int tile_raw = blockIdx.x * NW + wave;
  const int tile = min(tile_raw, tiles_total - 1);
  const int kgroup = blockIdx.y;
  const int P = phases_per_group;
  const int p0 = kgroup * P;               // first phase of this workgroup

  // ---- x slice staging: 16-byte vectors, whole rows, coalesced
  constexpr int NX = ROWS * (KSL / 8) / NT;
  typedef unsigned int xvec __attribute__((ext_vector_type(4)));   // native vector: stays in registers
  auto x_issue = [&](xvec (&xs)[NX], int p) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int v = i * NT + tid;
      const int row = v / (KSL / 8), col = v % (KSL / 8);
      xs[i] = *reinterpret_cast<const xvec*>(x + static_cast<int64_t>(min(row, M - 1)) * x_stride +
                                             static_cast<int64_t>(p) * KSL + col * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto x_commit = [&](xvec (&xs)[NX], int buf) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int v = i * NT + tid;
      const int row = v / (KSL / 8), col = v % (KSL / 8);
      *reinterpret_cast<xvec*>(&lds[buf * XBUF + row * ROWB + col * 16]) = xs[i];
    }
  };

  // ---- weight stream of this wave: blocks [tile][kstep], 512 elements each
  const uint16_t* wbase = wt + (static_cast<int64_t>(tile) * (K >> 5) + static_cast<int64_t>(p0) * (KSL / 32)) * 512 + lane * 8;
  const int groups_total = P * G;
  auto load_group = [&](uint4 (&wd)[U], int g) {  // g clamped: the tail re-loads the last group (never used)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const uint16_t* p = wbase + static_cast<int64_t>(min(g, groups_total - 1)) * (U * 512);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + u * 512));
      wd[u] = make_uint4(v.x, v.y, v.z, v.w);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  f32x4 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](uint4 (&wd)[U], int buf, int g) {  // g = group index inside the phase
    const unsigned char* xb = &lds[buf * XBUF + c * ROWB + g * (U * 64) + 16 * h];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int m = 0; m < MT; ++m)
        acc[m] = T::mfma(wd[u], *reinterpret_cast<const uint4*>(xb + u * 64 + m * 16 * ROWB), acc[m]);
    __builtin_amdgcn_sched_barrier(0);
  };

  uint4 ring[D][U];
#pragma unroll
  for (int j = 0; j < D - 1; ++j) load_group(ring[j], j);
  {
    xvec xs0[NX];
    x_issue(xs0, p0);
    x_commit(xs0, 0);
  }
  __syncthreads();
  for (int p = 0; p < P; ++p) {
    const bool more = p + 1 < P;
    xvec xs[NX];                           // lives inside one iteration only
    x_issue(xs, p0 + (more ? p + 1 : p));  // unconditional (the last phase re-loads its own slice, unused)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      load_group(ring[(g + D - 1) % D], p * G + g + D - 1);
      compute(ring[g % D], p & 1, g);
    }
    if (more) {
      x_commit(xs, (p + 1) & 1);
      __syncthreads();
    }
  }

  // ---- epilogue: lane holds D[n = 4h + r][m = c] of each m-tile
  const int n0 = tile_raw * 16 + 4 * h;
  const bool live = tile_raw < tiles_total;
  if (EPI == EPI_SILU_MUL) {
    __syncthreads();                                  // x buffers are dead: reuse them for the gate/up exchange
    float* ex = reinterpret_cast<float*>(lds);        // [pair][mt][r][lane]
    if (wave & 1) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) ex[(((wave >> 1) * MT + m) * 4 + r) * 64 + lane] = acc[m][r];
    }
    __syncthreads();
    if ((wave & 1) || !live) return;
    const int F = N >> 1;
    const int j0 = (tile_raw >> 1) * 16 + 4 * h;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int mm = 16 * m + c;
      if (mm >= M) continue;
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float g = round_f<T>(acc[m][r]);
        const float u = round_f<T>(ex[(((wave >> 1) * MT + m) * 4 + r) * 64 + lane]);
        o[r] = round_f<T>(g / (1.0f + expf(-g))) * u;
      }
      uint16_t* yp = y + static_cast<int64_t>(mm) * y_stride + j0;
      if (j0 + 3 < F) {
        *reinterpret_cast<uint2*>(yp) = make_uint2(
            static_cast<uint32_t>(T::from_f(o[0])) | (static_cast<uint32_t>(T::from_f(o[1])) << 16),
            static_cast<uint32_t>(T::from_f(o[2])) | (static_cast<uint32_t>(T::from_f(o[3])) << 16));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (j0 + r < F) yp[r] = T::from_f(o[r]);
      }
    }
    return;
  }
  if (!live) return;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (EPI == EPI_STORE && bias != nullptr) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n0 + r < N) bv[r] = T::to_f(bias[n0 + r]);
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int mm = 16 * m + c;
    if (mm >= M) continue;
    if (EPI == EPI_PARTIAL) {
      float* pp = partial + (static_cast<int64_t>(kgroup) * M + mm) * N + n0;
      if (n0 + 3 < N && (N & 3) == 0) {
        *reinterpret_cast<float4*>(pp) = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n0 + r < N) pp[r] = acc[m][r];
      }
    } else {
      uint16_t* yp = y + static_cast<int64_t>(mm) * y_stride + n0;
      if (n0 + 3 < N && (y_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0) {
        *reinterpret_cast<uint2*>(yp) = make_uint2(
            static_cast<uint32_t>(T::from_f(acc[m][0] + bv[0])) | (static_cast<uint32_t>(T::from_f(acc[m][1] + bv[1])) << 16),
            static_cast<uint32_t>(T::from_f(acc[m][2] + bv[2])) | (static_cast<uint32_t>(T::from_f(acc[m][3] + bv[3])) << 16));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n0 + r < N) yp[r] = T::from_f(acc[m][r] + bv[r]);
      }
    }
  }
}

// Row-major [N, K] -> fragment-tiled (see the top of the file); rows beyond N are zero.
// With `interleave_halves` the source rows are first reordered so that 16-row tiles
// alternate between the first half (gate) and the second half (up) of the matrix.
template <int DUMMY>
__global__ void tile_weights_kernel(uint16_t* __restrict__ dst, const uint16_t* __restrict__ src, int N, int K,
                                    int tiles_total, int interleave_halves) {
  const int64_t blocks = static_cast<int64_t>(tiles_total) * (K >> 5);
  for (int64_t b = blockIdx.x; b < blocks; b += gridDim.x) {
    const int tile = static_cast<int>(b / (K >> 5)), ks = static_cast<int>(b % (K >> 5));
    const int lane = threadIdx.x, i = lane & 15, h = lane >> 4;
    int row;
    if (interleave_halves) {
      const int pair = tile >> 1;
      row = (tile & 1) * (N >> 1) + pair * 16 + i;
      if (pair * 16 + i >= (N >> 1)) row = -1;
    } else {
      row = tile * 16 + i;
      if (row >= N) row = -1;
    }
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row >= 0) v = *reinterpret_cast<const uint4*>(src + static_cast<int64_t>(row) * K + 32 * ks + 8 * h);
    *reinterpret_cast<uint4*>(dst + (b * 64 + lane) * 8) = v;
  }
}

template <typename T, int MT, int NW, int KSL>
static void launch_epi(int epi, dim3 grid, hipStream_t s, uint16_t* y, float* partial, const uint16_t* x,
                       const uint16_t* wt, const uint16_t* bias, int M, int N, int K, int64_t xs, int64_t ys,
                       int tiles_total, int ppg) {
  dim3 block(NW * 64);
  if (epi == EPI_SILU_MUL)
    hipLaunchKernelGGL((linear_tiled_kernel<T, MT, NW, KSL, EPI_SILU_MUL>), grid, block, 0, s, y, partial, x, wt, bias,
                       M, N, K, xs, ys, tiles_total, ppg);
  else if (epi == EPI_PARTIAL)
    hipLaunchKernelGGL((linear_tiled_kernel<T, MT, NW, KSL, EPI_PARTIAL>), grid, block, 0, s, y, partial, x, wt, bias, M,
                       N, K, xs, ys, tiles_total, ppg);
  else
    hipLaunchKernelGGL((linear_tiled_kernel<T, MT, NW, KSL, EPI_STORE>), grid, block, 0, s, y, partial, x, wt, bias, M, N,
                       K, xs, ys, tiles_total, ppg);
}

template <typename T, int MT>
static void launch_geom(int nw, int ksl, int epi, dim3 grid, hipStream_t s, uint16_t* y, float* partial,
                        const uint16_t* x, const uint16_t* wt, const uint16_t* bias, int M, int N, int K,
                        int64_t xs, int64_t ys, int tiles_total, int ppg) {
  if (nw == 8 && ksl == 1024) launch_epi<T, MT, 8, 1024>(epi, grid, s, y, partial, x, wt, bias, M, N, K, xs, ys, tiles_total, ppg);
  else if (nw == 8) launch_epi<T, MT, 8, 512>(epi, grid, s, y, partial, x, wt, bias, M, N, K, xs, ys, tiles_total, ppg);
  else if (ksl == 1024) launch_epi<T, MT, 4, 1024>(epi, grid, s, y, partial, x, wt, bias, M, N, K, xs, ys, tiles_total, ppg);
  else launch_epi<T, MT, 4, 512>(epi, grid, s, y, partial, x, wt, bias, M, N, K, xs, ys, tiles_total, ppg);
}

}  // namespace mi355

extern "C" int64_t mi355_tiled_weight_elems(int32_t N, int32_t K) {
  if (N <= 0 || K <= 0 || K % 32) return 0;
  return static_cast<int64_t>((N + 15) / 16) * 16 * K;
}

extern "C" int32_t mi355_tile_weights(void* dst, const void* src, int32_t N, int32_t K,
                                      int32_t interleave_halves, void* stream) {
  using namespace mi355;
  const char* name = "mi355_tile_weights";
  MI355_REQUIRE(dst && src && dst != src, MI355_ERR_INVALID, "%s: null or aliased pointers", name);
  MI355_REQUIRE(N > 0 && K > 0 && K % 32 == 0, MI355_ERR_UNSUPPORTED, "%s: K %d must be a multiple of 32", name, K);
  MI355_REQUIRE(!interleave_halves || N % 32 == 0, MI355_ERR_UNSUPPORTED,
                "%s: interleaving needs N %% 32 == 0 (two halves of whole 16-row tiles)", name);
  MI355_REQUIRE(aligned16(dst) && aligned16(src), MI355_ERR_INVALID, "%s: pointers must be 16-byte aligned", name);
  const int tiles = (N + 15) / 16;
  const int64_t blocks = static_cast<int64_t>(tiles) * (K / 32);
  hipLaunchKernelGGL((tile_weights_kernel<0>), dim3(static_cast<unsigned>(blocks > 65535 ? 65535 : blocks)), dim3(64), 0,
                     static_cast<hipStream_t>(stream), static_cast<uint16_t*>(dst), static_cast<const uint16_t*>(src), N,
                     K, tiles, interleave_halves);
  return check_launch(name);
}

extern "C" int32_t mi355_linear_tiled(void* y, int64_t y_stride, void* partial, const void* x, int64_t x_stride,
                                      const void* w_tiled, const void* bias, int32_t M, int32_t N, int32_t K,
                                      int32_t tiles_per_wg, int32_t k_groups, int32_t epilogue, int32_t dtype,
                                      void* stream) {
  using namespace mi355;
  const char* name = "mi355_linear_tiled";
  MI355_REQUIRE(x && w_tiled, MI355_ERR_INVALID, "%s: null pointer", name);
  MI355_REQUIRE(M >= 0 && M <= 32, MI355_ERR_UNSUPPORTED, "%s: M %d outside [0, 32]", name, M);
  MI355_REQUIRE(N > 0 && K > 0 && K % 512 == 0, MI355_ERR_UNSUPPORTED, "%s: K %d must be a multiple of 512", name, K);
  MI355_REQUIRE(epilogue == EPI_STORE || epilogue == EPI_SILU_MUL || epilogue == EPI_PARTIAL, MI355_ERR_INVALID,
                "%s: unknown epilogue %d", name, epilogue);
  // tiles_per_wg: 4 or 8 tiles (= waves) per workgroup; +100 selects 512-wide x slices (two
  // workgroups per CU) instead of 1024-wide ones
  const int nw = tiles_per_wg % 100, ksl = tiles_per_wg >= 100 ? 512 : 1024;
  MI355_REQUIRE((nw == 4 || nw == 8) && (tiles_per_wg == nw || tiles_per_wg == nw + 100), MI355_ERR_INVALID,
                "%s: tiles_per_wg must be 4, 8, 104 or 108", name);
  MI355_REQUIRE(K % ksl == 0, MI355_ERR_UNSUPPORTED, "%s: K %d must be a multiple of the x slice %d", name, K, ksl);
  const int phases = K / ksl;
  MI355_REQUIRE(k_groups >= 1 && phases % k_groups == 0, MI355_ERR_INVALID,
                "%s: k_groups %d must divide K/%d = %d", name, k_groups, ksl, phases);
  MI355_REQUIRE(k_groups == 1 || epilogue == EPI_PARTIAL, MI355_ERR_INVALID,
                "%s: k_groups > 1 needs the partial-sum epilogue", name);
  if (epilogue == EPI_PARTIAL)
    MI355_REQUIRE(partial && aligned16(partial), MI355_ERR_INVALID, "%s: partial buffer missing", name);
  else
    MI355_REQUIRE(y && y_stride >= (epilogue == EPI_SILU_MUL ? N / 2 : N), MI355_ERR_INVALID,
                  "%s: y missing or y_stride too small", name);
  if (epilogue == EPI_SILU_MUL)
    MI355_REQUIRE(N % 32 == 0 && bias == nullptr, MI355_ERR_UNSUPPORTED,
                  "%s: SwiGLU epilogue needs N %% 32 == 0 and no bias", name);
  MI355_REQUIRE(x_stride >= K && x_stride % 8 == 0 && aligned16(x) && aligned16(w_tiled), MI355_ERR_INVALID,
                "%s: x/w_tiled must be 16-byte aligned, x_stride a multiple of 8", name);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "%s: dtype %d not supported (bf16/fp16 only)", name, dtype);
  if (M == 0) return MI355_OK;
  const int tiles_total = (N + 15) / 16;
  dim3 grid((tiles_total + nw - 1) / nw, k_groups);
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto* yp = static_cast<uint16_t*>(y);
  auto* pp = static_cast<float*>(partial);
  auto* xp = static_cast<const uint16_t*>(x);
  auto* wp = static_cast<const uint16_t*>(w_tiled);
  auto* bp = static_cast<const uint16_t*>(bias);
  const int ppg = phases / k_groups;
  if (dtype == MI355_BF16) {
    if (M <= 16) launch_geom<BF16, 1>(nw, ksl, epilogue, grid, s, yp, pp, xp, wp, bp, M, N, K, x_stride, y_stride, tiles_total, ppg);
    else launch_geom<BF16, 2>(nw, ksl, epilogue, grid, s, yp, pp, xp, wp, bp, M, N, K, x_stride, y_stride, tiles_total, ppg);
  } else {
    if (M <= 16) launch_geom<F16, 1>(nw, ksl, epilogue, grid, s, yp, pp, xp, wp, bp, M, N, K, x_stride, y_stride, tiles_total, ppg);
    else launch_geom<F16, 2>(nw, ksl, epilogue, grid, s, yp, pp, xp, wp, bp, M, N, K, x_stride, y_stride, tiles_total, ppg);
  }
  return check_launch(name);
}
