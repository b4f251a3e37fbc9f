// W8A8 skinny GEMM: fp8 (e4m3fn) weights AND fp8 activations on the fp8 matrix instruction
// (v_mfma_f32_16x16x32_fp8_fp8), fp32 accumulation, one fp32 scale per weight row and one per
// activation row:   y[m, n] = round16( ((sum_k xq[m, k] * wq[n, k]) * w_scale[n]) * x_scale[m] + bias[n] ).
//
// The weights are the fp8 fragment-tiled copy mi355_tile_weights_fp8 writes (1 KB blocks ordered
// [k group][tile][k-step pair]; a lane's 16 bytes are its 8 elements of two consecutive k-steps):
// a lane's 8 bytes of a k-step ARE its A operand of the fp8 instruction, nothing is converted.
// The activations come as row-major fp8 rows + one scale per row (mi355_quant_rows_fp8 below, or
// any producer that owns whole rows), are staged through LDS like the 16-bit kernels stage theirs
// (half the bytes: from L2, into LDS and out of it again), and a lane's B operand is 8 bytes of a
// row.  Work decomposition and loop structure are those of the two-tiles-per-wave kernel of
// linear_tiled.hip: NW waves per workgroup, each streaming two adjacent 16-row tiles (the gate / up
// pair of an interleaved SwiGLU matrix) and using every x fragment for both; K may also be split
// over workgroups (gridDim.y) into fp32 slabs.
#include "common.h"

namespace mi355 {
namespace {

constexpr int kU = 4;            // k-steps per register group: the fp8 tiled weight format's group size
enum { A8_STORE = 0, A8_SILU_MUL = 1, A8_PARTIAL = 2 };

__device__ __forceinline__ f32x4 mfma_fp8(uint32_t a_lo, uint32_t a_hi, uint2 b, f32x4 c) {
  const long a64 = static_cast<long>((static_cast<unsigned long>(a_hi) << 32) | a_lo);
  const long b64 = static_cast<long>((static_cast<unsigned long>(b.y) << 32) | b.x);
  return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a64, b64, c, 0, 0, 0);
}

template <typename T, int MT, int NW, int KSL, int EPI>
__global__ __launch_bounds__(NW * 64) void linear_w8a8_kernel(
    uint16_t* __restrict__ y, float* __restrict__ partial, const unsigned char* __restrict__ xq,
    const float* __restrict__ x_scale, const unsigned char* __restrict__ wt, const float* __restrict__ w_scale,
    const uint16_t* __restrict__ bias, int M, int N, int K, int64_t x_stride, int64_t y_stride, int tiles_total,
    int phases_total) {
  constexpr int TPW = 2;
  constexpr int ROWS = MT * 16;
  constexpr int U = kU;
  constexpr int NT = NW * 64;
  // LDS row: KSL bytes + 16.  A lane reads 8 bytes at row (lane & 15), byte 8 * (lane >> 4): with rows
  // 4 banks apart (mod 64) the 32 lanes of each half of the wave fall on 32 different bank pairs
  constexpr int ROWB = KSL + 16;
  constexpr int XBUF = ROWS * ROWB;
  constexpr int G = KSL / 32 / U;            // register groups per phase
  constexpr int D = 2;                       // ring: one group computing, one in flight (4 measured no faster)
  constexpr int PU = G >= D ? 1 : D / G;
  constexpr int LPG = U / 2;                 // 1 KB loads per register group and tile
  static_assert((PU * G) % D == 0, "ring slots must repeat every main-loop iteration");
  const int ubase = phases_total / static_cast<int>(gridDim.y), urem = phases_total % static_cast<int>(gridDim.y);
  const int P = ubase + (static_cast<int>(blockIdx.y) < urem ? 1 : 0);
  const int p0 = static_cast<int>(blockIdx.y) * ubase + min(static_cast<int>(blockIdx.y), urem);
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * XBUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, h = lane >> 4;
  const int tile_raw = (blockIdx.x * NW + wave) * TPW;
  const int tile0 = min(tile_raw, tiles_total - 1), tile1 = min(tile_raw + 1, tiles_total - 1);
  const int kgroup = blockIdx.y;

  // the scales of this lane's rows (m = 16 * mt + c): asked for now, used in the epilogue
  float xsc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) xsc[m] = MI355_LD(float, x_scale, x_scale + min(16 * m + c, M - 1));

  // ---- x slice staging: 16-byte vectors (16 elements), whole rows, coalesced; a thread count that
  // does not divide the slice repeats the slice's last vector
  constexpr int NXV = ROWS * (KSL / 16);
  constexpr int NX = (NXV + NT - 1) / NT;
  typedef unsigned int xvec __attribute__((ext_vector_type(4)));
  auto x_issue = [&](xvec (&xs)[NX], int p) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int v = NXV % NT == 0 ? i * NT + tid : min(i * NT + tid, NXV - 1);
      const int row = v / (KSL / 16), col = v % (KSL / 16);
      xs[i] = MI355_LD(xvec, xq, xq + static_cast<int64_t>(min(row, M - 1)) * x_stride +
                                     static_cast<int64_t>(p) * KSL + col * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto x_commit = [&](xvec (&xs)[NX], int buf) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int v = NXV % NT == 0 ? i * NT + tid : min(i * NT + tid, NXV - 1);
      const int row = v / (KSL / 16), col = v % (KSL / 16);
      *reinterpret_cast<xvec*>(&lds[buf * XBUF + row * ROWB + col * 16]) = xs[i];
    }
  };

  // ---- weight streams of the wave's two tiles
  const int64_t gstride = static_cast<int64_t>(tiles_total) * (LPG * 1024);      // bytes from one k group to the next
  const unsigned char* wbase0 = wt + (static_cast<int64_t>(p0) * G * tiles_total + tile0) * (LPG * 1024) + lane * 16;
  const unsigned char* wbase1 = wt + (static_cast<int64_t>(p0) * G * tiles_total + tile1) * (LPG * 1024) + lane * 16;
  const int groups_total = P * G;
  auto load_group = [&](uint4 (&wd)[TPW][LPG], int g) {   // g clamped: the tail re-loads the last group (never used)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int64_t off = min(g, groups_total - 1) * gstride;
#pragma unroll
    for (int u = 0; u < LPG; ++u) {
      const u32x4 v0 = MI355_LDNT(u32x4, wt, wbase0 + off + u * 1024);
      wd[0][u] = make_uint4(v0.x, v0.y, v0.z, v0.w);
    }
#pragma unroll
    for (int u = 0; u < LPG; ++u) {
      const u32x4 v1 = MI355_LDNT(u32x4, wt, wbase1 + off + u * 1024);
      wd[1][u] = make_uint4(v1.x, v1.y, v1.z, v1.w);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  f32x4 acc[TPW][MT];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](uint4 (&wd)[TPW][LPG], int buf, int g) {
    const unsigned char* xb = &lds[buf * XBUF + c * ROWB + g * (U * 32) + 8 * h];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const uint2 b = *reinterpret_cast<const uint2*>(xb + u * 32 + m * 16 * ROWB);   // one fragment, both tiles
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          const uint4 v = wd[j][u / 2];
          acc[j][m] = (u & 1) ? mfma_fp8(v.z, v.w, b, acc[j][m]) : mfma_fp8(v.x, v.y, b, acc[j][m]);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  uint4 ring[D][TPW][LPG];
#pragma unroll
  for (int j = 0; j < D - 1; ++j) load_group(ring[j], j);
  {
    xvec xs0[NX];
    x_issue(xs0, p0);
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
      x_issue(xs, p0 + (more ? q + 1 : q));        // unconditional (the last phase re-loads its own slice, unused)
#pragma unroll
      for (int g = 0; g < G; ++g) {
        load_group(ring[(pp * G + g + D - 1) % D], q * G + g + D - 1);
        compute(ring[(pp * G + g) % D], q & 1, g);
      }
      if (more) {
        x_commit(xs, (q + 1) & 1);
        __syncthreads();
      }
    }
  }

  // ---- epilogue: lane holds D[n = 4h + r][m = c] of each m-tile, for both tiles.  The weight row's
  // scale first, then the activation row's: two fp32 products, each a value of its own
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const float4 sc = MI355_LD(float4, w_scale, w_scale + (j ? tile1 : tile0) * 16 + 4 * h);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      acc[j][m][0] = pin_f32(pin_f32(acc[j][m][0] * sc.x) * xsc[m]);
      acc[j][m][1] = pin_f32(pin_f32(acc[j][m][1] * sc.y) * xsc[m]);
      acc[j][m][2] = pin_f32(pin_f32(acc[j][m][2] * sc.z) * xsc[m]);
      acc[j][m][3] = pin_f32(pin_f32(acc[j][m][3] * sc.w) * xsc[m]);
    }
  }
  if (EPI == A8_SILU_MUL) {            // tile_raw is even: tile0 = gate rows, tile1 = up rows of the same 16 columns
    if (tile_raw + 1 >= tiles_total) return;
    const int F = N >> 1;
    const int j0 = (tile_raw >> 1) * 16 + 4 * h;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int mm = 16 * m + c;
      if (mm >= M) continue;
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = silu_mul<T>(round_f<T>(acc[0][m][r]), round_f<T>(acc[1][m][r]));
      uint16_t* yp = y + static_cast<int64_t>(mm) * y_stride + j0;
      if (j0 + 3 < F) {
        MI355_ST(uint2, y, yp, make_uint2(
            static_cast<uint32_t>(T::from_f(o[0])) | (static_cast<uint32_t>(T::from_f(o[1])) << 16),
            static_cast<uint32_t>(T::from_f(o[2])) | (static_cast<uint32_t>(T::from_f(o[3])) << 16)));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (j0 + r < F) MI355_ST(uint16_t, y, yp + r, T::from_f(o[r]));
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    if (tile_raw + j >= tiles_total) continue;
    const int n0 = (tile_raw + j) * 16 + 4 * h;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (EPI == A8_STORE && bias != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n0 + r < N) bv[r] = T::to_f(MI355_LD(uint16_t, bias, bias + (n0 + r)));
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int mm = 16 * m + c;
      if (mm >= M) continue;
      if (EPI == A8_PARTIAL) {
        float* pp = partial + (static_cast<int64_t>(kgroup) * M + mm) * N + n0;
        if (n0 + 3 < N && (N & 3) == 0) {
          MI355_ST(float4, partial, pp, make_float4(acc[j][m][0], acc[j][m][1], acc[j][m][2], acc[j][m][3]));
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n0 + r < N) MI355_ST(float, partial, pp + r, acc[j][m][r]);
        }
      } else {
        uint16_t* yp = y + static_cast<int64_t>(mm) * y_stride + n0;
        if (n0 + 3 < N && (y_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0) {
          MI355_ST(uint2, y, yp, make_uint2(
              static_cast<uint32_t>(T::from_f(acc[j][m][0] + bv[0])) | (static_cast<uint32_t>(T::from_f(acc[j][m][1] + bv[1])) << 16),
              static_cast<uint32_t>(T::from_f(acc[j][m][2] + bv[2])) | (static_cast<uint32_t>(T::from_f(acc[j][m][3] + bv[3])) << 16)));
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n0 + r < N) MI355_ST(uint16_t, y, yp + r, T::from_f(acc[j][m][r] + bv[r]));
        }
      }
    }
  }
}

// ---- activation rows -> fp8: scale = max|row| / 448 (1 for an all-zero row), q = fp8_rne(x / scale):
// the recipe of the weight copies (mi355_tile_weights_fp8).  One workgroup per row; the row is read
// twice (the second time from cache).
template <typename T>
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(unsigned char* __restrict__ xq, int64_t xq_stride,
                                                             float* __restrict__ scale, const uint16_t* __restrict__ x,
                                                             int64_t x_stride, int K) {
  const int row = blockIdx.x;
  const uint16_t* xr = x + static_cast<int64_t>(row) * x_stride;
  float amax = 0.f;
  for (int v = threadIdx.x; v < (K >> 3); v += 256) {
    float f[8];
    unpack8<T>(MI355_LD(uint4, x, reinterpret_cast<const uint4*>(xr) + v), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
  const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
  if (threadIdx.x == 0) MI355_ST(float, scale, scale + row, sc);
  unsigned char* qr = xq + static_cast<int64_t>(row) * xq_stride;
  for (int v = threadIdx.x; v < (K >> 3); v += 256) {
    float f[8];
    unpack8<T>(MI355_LD(uint4, x, reinterpret_cast<const uint4*>(xr) + v), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = f[j] / sc;
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    MI355_ST(uint2, xq, qr + 8 * v, make_uint2(static_cast<uint32_t>(lo), static_cast<uint32_t>(hi)));
  }
}

#define MI355_A8_ARGS y, partial, xq, x_scale, wt, w_scale, bias, M, N, K, xs, ys, tiles_total, phases
#define MI355_A8_PARAMS                                                                                            \
  uint16_t *y, float *partial, const unsigned char *xq, const float *x_scale, const unsigned char *wt,             \
      const float *w_scale, const uint16_t *bias, int M, int N, int K, int64_t xs, int64_t ys, int tiles_total, \
      int phases

template <typename T, int MT, int NW, int KSL>
bool launch_cfg(int epi, dim3 grid, hipStream_t s, MI355_A8_PARAMS) {
#define MI355_A8(EPI_) \
  hipLaunchKernelGGL((linear_w8a8_kernel<T, MT, NW, KSL, EPI_>), grid, dim3(NW * 64), 0, s, MI355_A8_ARGS)
  if (epi == A8_STORE) { MI355_A8(A8_STORE); return true; }
  if (epi == A8_SILU_MUL) { MI355_A8(A8_SILU_MUL); return true; }
  if (epi == A8_PARTIAL) { MI355_A8(A8_PARTIAL); return true; }
#undef MI355_A8
  return false;
}

template <typename T>
bool launch_geometry(int mt, int nw, int ksl, int epi, dim3 grid, hipStream_t s, MI355_A8_PARAMS) {
#define MI355_A8_GEOM(MT_, NW_, KSL_) \
  if (mt == MT_ && nw == NW_ && ksl == KSL_) return launch_cfg<T, MT_, NW_, KSL_>(epi, grid, s, MI355_A8_ARGS);
  MI355_A8_GEOM(2, 4, 512) MI355_A8_GEOM(2, 8, 1024)
  MI355_A8_GEOM(4, 4, 256) MI355_A8_GEOM(4, 5, 512) MI355_A8_GEOM(4, 6, 512) MI355_A8_GEOM(4, 7, 512) MI355_A8_GEOM(4, 8, 512)
  MI355_A8_GEOM(8, 5, 256) MI355_A8_GEOM(8, 6, 256) MI355_A8_GEOM(8, 7, 256) MI355_A8_GEOM(8, 8, 256)
#undef MI355_A8_GEOM
  return false;
}

}  // namespace
}  // namespace mi355

extern "C" int32_t mi355_quant_rows_fp8(void* xq, int64_t xq_stride, float* scale, const void* x, int64_t x_stride,
                                        int32_t M, int32_t K, int32_t dtype, void* stream) {
  using namespace mi355;
  MI355_REQUIRE(xq && scale && x, MI355_ERR_INVALID, "quant_rows_fp8: null pointer");
  MI355_REQUIRE(M >= 0 && K > 0 && K % 8 == 0, MI355_ERR_INVALID, "quant_rows_fp8: M %d, K %d (K must be a multiple of 8)",
                M, K);
  MI355_REQUIRE(x_stride >= K && x_stride % 8 == 0 && aligned16(x), MI355_ERR_INVALID,
                "quant_rows_fp8: x must be 16-byte aligned, x_stride >= K and a multiple of 8");
  MI355_REQUIRE(xq_stride >= K && xq_stride % 8 == 0 && (reinterpret_cast<uintptr_t>(xq) & 7u) == 0, MI355_ERR_INVALID,
                "quant_rows_fp8: xq must be 8-byte aligned, xq_stride >= K and a multiple of 8");
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "quant_rows_fp8: dtype %d not supported (bf16/fp16 only)", dtype);
  if (M == 0) return MI355_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == MI355_BF16)
    hipLaunchKernelGGL((quant_rows_fp8_kernel<BF16>), dim3(M), dim3(256), 0, s, static_cast<unsigned char*>(xq), xq_stride,
                       scale, static_cast<const uint16_t*>(x), x_stride, K);
  else
    hipLaunchKernelGGL((quant_rows_fp8_kernel<F16>), dim3(M), dim3(256), 0, s, static_cast<unsigned char*>(xq), xq_stride,
                       scale, static_cast<const uint16_t*>(x), x_stride, K);
  return check_launch("quant_rows_fp8");
}

extern "C" int32_t mi355_linear_tiled_w8a8(void* y, int64_t y_stride, void* partial, const void* xq, int64_t xq_stride,
                                           const float* x_scale, const void* w_tiled_fp8, const float* w_scale_tiled,
                                           const void* bias, int32_t M, int32_t N, int32_t K, int32_t tiles_per_wg,
                                           int32_t k_groups, int32_t epilogue, int32_t dtype, void* stream) {
  using namespace mi355;
  const char* name = "linear_tiled_w8a8";
  MI355_REQUIRE(xq && x_scale && w_tiled_fp8 && w_scale_tiled, MI355_ERR_INVALID, "%s: null pointer", name);
  MI355_REQUIRE(M >= 0 && M <= 128, MI355_ERR_UNSUPPORTED, "%s: M %d outside [0, 128]", name, M);
  MI355_REQUIRE(epilogue >= A8_STORE && epilogue <= A8_PARTIAL, MI355_ERR_INVALID, "%s: epilogue %d outside [0, 2]", name,
                epilogue);
  MI355_REQUIRE(tiles_per_wg >= 208 && tiles_per_wg <= 216 && tiles_per_wg % 2 == 0, MI355_ERR_INVALID,
                "%s: tiles_per_wg must be 208, 210, 212, 214 or 216 (two tiles per wave, 4 .. 8 waves)", name);
  // the x slices of the 16-bit kernels of the same code: 1024 (208: 512) columns up to 32 rows, halved
  // with every doubling of the rows
  const int nw = (tiles_per_wg - 200) / 2;
  const int mt = M <= 32 ? 2 : M <= 64 ? 4 : 8;
  const int ksl = (tiles_per_wg == 208 ? 512 : 1024) / (mt / 2);
  MI355_REQUIRE(N > 0 && K > 0 && K % ksl == 0, MI355_ERR_UNSUPPORTED, "%s: K %d must be a multiple of the x slice %d (M %d)",
                name, K, ksl, M);
  const int phases = K / ksl;
  MI355_REQUIRE(k_groups >= 1 && k_groups <= phases, MI355_ERR_INVALID, "%s: k_groups %d outside [1, %d] (K / %d)", name,
                k_groups, phases, ksl);
  MI355_REQUIRE(k_groups == 1 || epilogue == A8_PARTIAL, MI355_ERR_INVALID, "%s: k_groups > 1 needs the partial-sum epilogue",
                name);
  MI355_REQUIRE(xq_stride >= K && xq_stride % 16 == 0 && aligned16(xq) && aligned16(w_tiled_fp8) && aligned16(w_scale_tiled),
                MI355_ERR_INVALID, "%s: xq / w_tiled_fp8 / w_scale_tiled must be 16-byte aligned, xq_stride a multiple of 16",
                name);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "%s: dtype %d not supported (bf16/fp16 output only)", name, dtype);
  if (epilogue == A8_PARTIAL)
    MI355_REQUIRE(partial && aligned16(partial), MI355_ERR_INVALID, "%s: partial buffer missing", name);
  else
    MI355_REQUIRE(y && y_stride >= (epilogue == A8_SILU_MUL ? N / 2 : N), MI355_ERR_INVALID,
                  "%s: y missing or y_stride too small", name);
  if (epilogue == A8_SILU_MUL)
    MI355_REQUIRE(N % 32 == 0 && bias == nullptr, MI355_ERR_UNSUPPORTED, "%s: SwiGLU epilogue needs N %% 32 == 0 and no bias",
                  name);
  if (M == 0) return MI355_OK;
  const int tiles_total = (N + 15) / 16;
  dim3 grid((tiles_total + 2 * nw - 1) / (2 * nw), k_groups);
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto* yp = static_cast<uint16_t*>(y);
  auto* pp = static_cast<float*>(partial);
  auto* xp = static_cast<const unsigned char*>(xq);
  auto* wp = static_cast<const unsigned char*>(w_tiled_fp8);
  auto* bp = static_cast<const uint16_t*>(bias);
  const bool ok = dtype == MI355_BF16
                      ? launch_geometry<BF16>(mt, nw, ksl, epilogue, grid, s, yp, pp, xp, x_scale, wp, w_scale_tiled, bp, M, N,
                                              K, xq_stride, y_stride, tiles_total, phases)
                      : launch_geometry<F16>(mt, nw, ksl, epilogue, grid, s, yp, pp, xp, x_scale, wp, w_scale_tiled, bp, M, N,
                                             K, xq_stride, y_stride, tiles_total, phases);
  MI355_REQUIRE(ok, MI355_ERR_UNSUPPORTED,
                "%s: no kernel for M %d with tiles_per_wg %d (M <= 32: 208, 216; M <= 64: 208 .. 216; M <= 128: 210 .. 216)",
                name, M, tiles_per_wg);
  return check_launch(name);
}
