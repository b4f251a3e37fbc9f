// EXPERIMENT (not built into the library): o_proj / down_proj with K split over the waves of a
// workgroup and the residual add in the epilogue, feeding the norm-on-load prologue of the next GEMM.
// Parity-green, measured slower than the K-split GEMM + add+norm pair it would replace (Llama-3-8B, 32
// rows: o 12.5 vs 12.6 us, down 29.2 vs 26.0 us, and the norm-on-load prologue adds 5.7 us to gate_up):
// every workgroup has to stage ALL of x (256 workgroups x 256 KB for o, x 900 KB for down), which
// costs more than the slabs and the norm launch it saves.

// Decode-time row-parallel projection (o_proj, down_proj) folded into the residual stream:
//
//     residual[M, N] <- round(round(x[M, K] . W[N, K]^T + bias) + residual)        (M <= 32)
//     sumsq[tile][m]  = sum over the tile's 16 columns of residual[m, .]^2
//
// on fragment-tiled weights (mi355_tile_weights).  The K-split form of mi355_linear_tiled hands
// fp32 slabs to a norm kernel: one more launch and a round trip of the slabs through memory.
// Here K is split across the eight WAVES of a workgroup instead of across workgroups: a
// workgroup owns one or two 16-row tiles of W, wave (tile, kq) streams every KW-th register
// group of its tile, the KW partial accumulators meet in LDS and are added in wave order
// (a fixed order: the result does not depend on timing), and the lead wave of each tile adds
// the residual in place.  Nothing crosses workgroups, so there is no ticket and no slab; the
// per-tile sums of squares are what the RMSNorm-on-load prologue of the next GEMM
// (mi355_linear_tiled_fused, norm_sumsq / norm_parts = number of tiles) turns into 1/rms.
//
// x staging, weight ring and MFMA mapping are those of linear_tiled.hip: the x slice of a phase
// (1024 columns) goes through LDS once per workgroup, double buffered; weights are 1 KB
// non-temporal loads, ring of two register groups.
#include "common.h"

namespace mi355 {

namespace {
constexpr int kU = 4;            // k-steps per register group (4 KB of weights)
constexpr int kSumsqPitch = 32;  // row pitch of the sum-of-squares table ([tile][32])
}  // namespace

template <typename T, int KW>
__global__ __launch_bounds__(512) void linear_resid_kernel(
    uint16_t* residual, int64_t res_stride, float* __restrict__ sumsq_out, const uint16_t* __restrict__ x,
    int64_t x_stride, const uint16_t* __restrict__ wt, const uint16_t* __restrict__ bias, int M, int N, int K,
    int tiles_total) {
  constexpr int NW = 8, NT = NW * 64, MT = 2, ROWS = 32, KSL = 1024, U = kU;
  constexpr int TPG = NW / KW;                 // tiles per workgroup
  constexpr int ROWB = KSL * 2 + 32;           // padded LDS row (bytes)
  constexpr int XBUF = ROWS * ROWB;
  constexpr int G = KSL / 32 / U;              // register groups per phase: 8
  constexpr int GL = G / KW;                   // ... of which this wave takes GL
  constexpr int D = 2;
  constexpr int PU = GL >= D ? 1 : D / GL;     // phases per main-loop iteration: ring slots stay compile-time
  static_assert(G % KW == 0 && (PU * GL) % D == 0, "ring slots must repeat every main-loop iteration");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * XBUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = wave / KW, kq = wave % KW;
  const int c = lane & 15, h = lane >> 4;
  const int tile_raw = blockIdx.x * TPG + tl;
  const int tile = min(tile_raw, tiles_total - 1);
  const int P = K / KSL;

  // ---- x slice staging (whole rows, 16-byte vectors, rows beyond M clamped)
  constexpr int NX = ROWS * (KSL / 8) / NT;
  typedef unsigned int xvec __attribute__((ext_vector_type(4)));
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

  // ---- weight stream: blocks ordered [k group][tile][k-step in group]; this wave's ls-th group is
  // group (ls / GL) * G + (ls % GL) * KW + kq of the matrix
  const uint16_t* wbase = wt + static_cast<int64_t>(tile) * (U * 512) + lane * 8;
  const int64_t gstride = static_cast<int64_t>(tiles_total) * (U * 512);
  const int groups_total = P * GL;
  auto group_of = [&](int ls) { return (ls / GL) * G + (ls % GL) * KW + kq; };
  auto load_group = [&](uint4 (&wd)[U], int ls) {       // ls clamped: the tail re-loads the last group (never used)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const uint16_t* p = wbase + group_of(min(ls, groups_total - 1)) * gstride;
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
  auto compute = [&](uint4 (&wd)[U], int buf, int g) {    // g = group index inside the phase
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
    x_issue(xs0, 0);
    x_commit(xs0, 0);
  }
  __syncthreads();
  for (int p = 0; p < P; p += PU) {
#pragma unroll
    for (int pp = 0; pp < PU; ++pp) {
      const int q = p + pp;
      if (q >= P) break;
      const bool more = q + 1 < P;
      xvec xs[NX];
      x_issue(xs, more ? q + 1 : q);                   // unconditional (the last phase re-loads its own slice, unused)
#pragma unroll
      for (int lg = 0; lg < GL; ++lg) {
        load_group(ring[(pp * GL + lg + D - 1) % D], q * GL + lg + D - 1);
        compute(ring[(pp * GL + lg) % D], q & 1, lg * KW + kq);
      }
      if (more) {
        x_commit(xs, (q + 1) & 1);
        __syncthreads();
      }
    }
  }

  // ---- the KW partial accumulators of a tile meet in LDS (the x buffers are dead)
  __syncthreads();
  float* red = reinterpret_cast<float*>(lds);            // [tile in workgroup][kq - 1][m-tile][r][lane]
  if (kq != 0) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(((tl * (KW - 1) + kq - 1) * MT + m) * 4 + r) * 64 + lane] = acc[m][r];
  }
  __syncthreads();
  if (kq != 0 || tile_raw >= tiles_total) return;
#pragma unroll
  for (int j = 1; j < KW; ++j)                           // wave order: a fixed order of additions
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[m][r] += red[(((tl * (KW - 1) + j - 1) * MT + m) * 4 + r) * 64 + lane];

  // ---- lane holds D[n = 4h + r][m = c] of each m-tile: bias, round, residual add, round, sums of squares
  const int n0 = tile_raw * 16 + 4 * h;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias != nullptr) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n0 + r < N) bv[r] = T::to_f(bias[n0 + r]);
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int mm = 16 * m + c;
    float ss = 0.f;
    if (mm < M) {
      uint16_t* rp = residual + static_cast<int64_t>(mm) * res_stride + n0;
      float f[4];
      if (n0 + 3 < N) {                                  // res_stride % 4 == 0 and n0 % 4 == 0: 8-byte aligned
        const uint2 rv = *reinterpret_cast<const uint2*>(rp);
        const float r0 = T::to_f(static_cast<uint16_t>(rv.x & 0xFFFFu)), r1 = T::to_f(static_cast<uint16_t>(rv.x >> 16));
        const float r2 = T::to_f(static_cast<uint16_t>(rv.y & 0xFFFFu)), r3 = T::to_f(static_cast<uint16_t>(rv.y >> 16));
        f[0] = round_f<T>(round_f<T>(acc[m][0] + bv[0]) + r0);
        f[1] = round_f<T>(round_f<T>(acc[m][1] + bv[1]) + r1);
        f[2] = round_f<T>(round_f<T>(acc[m][2] + bv[2]) + r2);
        f[3] = round_f<T>(round_f<T>(acc[m][3] + bv[3]) + r3);
        *reinterpret_cast<uint2*>(rp) = make_uint2(
            static_cast<uint32_t>(T::from_f(f[0])) | (static_cast<uint32_t>(T::from_f(f[1])) << 16),
            static_cast<uint32_t>(T::from_f(f[2])) | (static_cast<uint32_t>(T::from_f(f[3])) << 16));
#pragma unroll
        for (int r = 0; r < 4; ++r) ss += f[r] * f[r];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n0 + r < N) {
            f[r] = round_f<T>(round_f<T>(acc[m][r] + bv[r]) + T::to_f(rp[r]));
            rp[r] = T::from_f(f[r]);
            ss += f[r] * f[r];
          }
        }
      }
    }
    // the tile's 16 columns are spread over the four lanes (c, h = 0..3): fixed-order butterfly
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    if (h == 0 && mm < M) sumsq_out[static_cast<int64_t>(tile_raw) * kSumsqPitch + mm] = ss;
  }
}

}  // namespace mi355

extern "C" int32_t mi355_linear_tiled_resid(void* residual, int64_t residual_stride, float* sumsq_out, const void* x,
                                            int64_t x_stride, const void* w_tiled, const void* bias, int32_t M,
                                            int32_t N, int32_t K, int32_t waves_per_tile, int32_t dtype,
                                            void* stream) {
  using namespace mi355;
  const char* name = "mi355_linear_tiled_resid";
  MI355_REQUIRE(residual && sumsq_out && x && w_tiled, MI355_ERR_INVALID, "%s: null pointer", name);
  MI355_REQUIRE(M >= 0 && M <= 32, MI355_ERR_UNSUPPORTED, "%s: M %d outside [0, 32]", name, M);
  MI355_REQUIRE(N > 0 && K > 0 && K % 1024 == 0, MI355_ERR_UNSUPPORTED, "%s: K %d must be a multiple of 1024", name, K);
  MI355_REQUIRE(waves_per_tile == 2 || waves_per_tile == 4 || waves_per_tile == 8, MI355_ERR_INVALID,
                "%s: waves_per_tile must be 2, 4 or 8", name);
  MI355_REQUIRE(x_stride >= K && x_stride % 8 == 0 && residual_stride >= N && residual_stride % 4 == 0, MI355_ERR_INVALID,
                "%s: x_stride must be >= K and a multiple of 8, residual_stride >= N and a multiple of 4", name);
  MI355_REQUIRE(aligned16(x) && aligned16(w_tiled) && aligned16(residual) && aligned16(sumsq_out), MI355_ERR_INVALID,
                "%s: pointers must be 16-byte aligned", name);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "%s: dtype %d not supported (bf16/fp16 only)", name, dtype);
  if (M == 0) return MI355_OK;
  const int tiles_total = (N + 15) / 16;
  const int tpg = 8 / waves_per_tile;
  dim3 grid((tiles_total + tpg - 1) / tpg), block(512);
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto* rp = static_cast<uint16_t*>(residual);
  auto* xp = static_cast<const uint16_t*>(x);
  auto* wp = static_cast<const uint16_t*>(w_tiled);
  auto* bp = static_cast<const uint16_t*>(bias);
#define MI355_LR(TT, KW_)                                                                                     \
  hipLaunchKernelGGL((linear_resid_kernel<TT, KW_>), grid, block, 0, s, rp, residual_stride, sumsq_out, xp,  \
                     x_stride, wp, bp, M, N, K, tiles_total)
#define MI355_LR_T(TT)                                  \
  do {                                                  \
    if (waves_per_tile == 2) MI355_LR(TT, 2);           \
    else if (waves_per_tile == 4) MI355_LR(TT, 4);      \
    else MI355_LR(TT, 8);                               \
  } while (0)
  if (dtype == MI355_BF16) MI355_LR_T(BF16); else MI355_LR_T(F16);
#undef MI355_LR_T
#undef MI355_LR
  return check_launch(name);
}
