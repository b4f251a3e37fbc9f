// Exploration, not built into the library: the decode GEMM with FOUR tiles per wave and the K range of a tile group
// split across the waves of a workgroup.  It was wired into csrc/linear_tiled.hip as tiles_per_wg 502 / 504 / 602 /
// 604 (launcher: launch_geometry4, 32 / KW tiles per workgroup), parity-checked against the two-tile form (0.03 % of
// the outputs differ by one rounding step: another order of additions) and measured slower on every shape it was
// built for: Qwen2-7B gate_up at 64 rows 36.8 us (fp8) / 55.0 (bf16) against 33.3 / 49.7 with two tiles per wave;
// the slices of a Llama-3-8B TP4 / TP8 rank at 128 / 256 rows in row blocks of 64: never among the best four forms.
// What it halves -- the LDS reads per weight byte -- is not what bounds those launches.  The same kernel with TWO
// tiles per wave and the K range split across 4 waves (as many workgroups as the one-tile form 302, half its LDS
// reads; TPW as a template parameter, tiles_per_wg 702 / 704 / 708 / 802 / 804) was measured too: gate_up of a TP4
// rank at 128 rows, unsplit, 21.65 us against 20.47 for 302; down 15.4 (708) against 13.8 (304).
// ---- FOUR tiles per wave, K split across the waves of a workgroup (33 .. 64 rows per row block).
// From 64 rows on the kernels above are bound by the LDS, not by HBM: a CU's LDS hands out about one
// 1 KB fragment of x per 16 cycles, the time ONE of its four SIMDs needs for the MFMA that consumes
// it -- so with one fresh fragment per MFMA (one tile per wave) the matrix pipes can be a quarter
// busy, with two tiles per wave half (measured: the 64-row gate_up of Qwen2-7B takes the same time
// from HBM and from the memory-side cache, 63 us with one tile per wave, 48 with two).  Here a wave
// uses every fragment for FOUR adjacent tiles.  That alone would leave a quarter of the waves (a
// matrix of 2 368 tiles: 592 waves on 1 024 SIMDs), so the K range of a tile group is also split
// across KW waves of the workgroup: wave (tile group, kq) streams every KW-th unit (2 k-steps: 2 KB
// of each of its four tiles, 8 KB per unit) of the group's K range, the KW partial accumulators meet
// in LDS and are added in wave order, and the group's first wave runs the epilogue (store, SwiGLU --
// the four tiles are two gate / up pairs --, or fp32 slabs).  x staging as in the kernels above.
template <typename T, int MT, int KW, int KSL, int EPI, bool W8>
__global__ __launch_bounds__(512) void linear_tiled4_kernel(
    uint16_t* __restrict__ y, float* __restrict__ partial, const uint16_t* __restrict__ x,
    const uint16_t* __restrict__ wt, const uint16_t* __restrict__ bias, int M, int N, int K,
    int64_t x_stride, int64_t y_stride, int tiles_total, int phases_total, const float* __restrict__ w_scale) {
  constexpr int TPW = 4, NW = 8, NT = NW * 64, ROWS = MT * 16;
  constexpr int TGS = NW / KW;                 // tile groups per workgroup
  constexpr int ROWB = KSL * 2 + 32;           // padded LDS row (bytes)
  constexpr int XBUF = ROWS * ROWB;
  constexpr int UK = 2;                        // k-steps per unit
  constexpr int LPU = W8 ? 1 : 2;              // 1 KB loads per tile and unit (an fp8 block holds two k-steps)
  constexpr int KGE = (W8 ? 2 : 4) * 512;      // 16-bit elements per (k group, tile): 4 k-steps
  constexpr int G = KSL / 32 / UK;             // units per phase
  constexpr int GL = G / KW;                   // ... of which this wave takes GL
  constexpr int D = 2;
  constexpr int PU = GL >= D ? 1 : D / GL;     // phases per main-loop iteration: ring slots stay compile-time
  static_assert(G % KW == 0 && GL >= 1 && (PU * GL) % D == 0, "ring slots must repeat every main-loop iteration");
  static_assert(G % 2 == 0, "a phase must start on a k-group boundary");
  static_assert(2 * XBUF >= TGS * TPW * MT * 4 * 64 * 4, "the reduction reuses the x buffers");
  static_assert(NT % (KSL / 8) == 0, "a thread must keep one column of the x slice");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * XBUF];

  const int ubase = phases_total / static_cast<int>(gridDim.y), urem = phases_total % static_cast<int>(gridDim.y);
  const int P = ubase + (static_cast<int>(blockIdx.y) < urem ? 1 : 0);
  const int p0 = static_cast<int>(blockIdx.y) * ubase + min(static_cast<int>(blockIdx.y), urem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = wave / KW, kq = wave % KW;
  const int c = lane & 15, h = lane >> 4;
  const int tile_raw = (blockIdx.x * TGS + tg) * TPW;     // first of the wave's four tiles
  const int kgroup = blockIdx.y;
  MI355_ROW_BLOCK();

  constexpr int NX = ROWS * (KSL / 8) / NT;
  typedef unsigned int xvec __attribute__((ext_vector_type(4)));
  auto x_issue = [&](xvec (&xs)[NX], int p, int live) {   // live = 0: a load nobody uses, one address for all lanes
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int v = i * NT + tid;
      const int row = v / (KSL / 8), col = v % (KSL / 8);
      xs[i] = MI355_LD(xvec, x, x + (static_cast<int64_t>(min(row, M - 1)) * x_stride + col * 8) * live +
                                    static_cast<int64_t>(p) * KSL);
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

  // weight blocks are ordered [k group][tile][k-step in group]; unit a (absolute, 2 per k group) of tile t starts at
  // ((a / 2) * tiles_total + t) * KGE + (a % 2) * LPU * 512.  This wave's ls-th unit is unit (ls / GL) * G +
  // (ls % GL) * KW + kq of the workgroup's K range.  Tiles beyond the matrix are clamped (never stored).
  const int64_t gstride = static_cast<int64_t>(tiles_total) * KGE;
  const uint16_t* wbase = wt + static_cast<int64_t>(min(tile_raw, tiles_total - 1)) * KGE;
  int toff[TPW];                               // the other tiles of the wave, relative to the first (clamped)
#pragma unroll
  for (int t = 0; t < TPW; ++t) toff[t] = (min(tile_raw + t, tiles_total - 1) - min(tile_raw, tiles_total - 1)) * KGE;
  const int units_total = P * GL;
  auto load_unit = [&](uint4 (&wd)[TPW][LPU], int ls) {   // beyond the last unit: its first 16 bytes for every lane (never used)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int l = min(ls, units_total - 1);
    const int a = p0 * G + (l / GL) * G + (l % GL) * KW + kq;
    const uint16_t* p = wbase + (a >> 1) * gstride + (a & 1) * (LPU * 512) + (ls < units_total ? lane * 8 : 0);
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int u = 0; u < LPU; ++u) {
        const u32x4 v = MI355_LDNT(u32x4, wt, p + toff[t] + u * 512);
        wd[t][u] = make_uint4(v.x, v.y, v.z, v.w);
      }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto dequant8 = [&](uint32_t lo, uint32_t hi) {
    return make_uint4(T::fp8x2(lo, false), T::fp8x2(lo, true), T::fp8x2(hi, false), T::fp8x2(hi, true));
  };

  f32x4 acc[TPW][MT];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](uint4 (&wd)[TPW][LPU], int buf, int gi) {   // gi = unit index inside the phase
    const unsigned char* xb = &lds[buf * XBUF + c * ROWB + gi * (UK * 64) + 16 * h];
#pragma unroll
    for (int u = 0; u < UK; ++u) {
      uint4 a[TPW];
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        if (W8) {
          const uint4 v = wd[t][0];
          a[t] = (u & 1) ? dequant8(v.z, v.w) : dequant8(v.x, v.y);
        } else {
          a[t] = wd[t][u % LPU];
        }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const uint4 b = *reinterpret_cast<const uint4*>(xb + u * 64 + m * 16 * ROWB);   // one fragment, four tiles
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t][m] = T::mfma(a[t], b, acc[t][m]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  uint4 ring[D][TPW][LPU];
#pragma unroll
  for (int j = 0; j < D - 1; ++j) load_unit(ring[j], j);
  {
    xvec xs0[NX];
    x_issue(xs0, p0, 1);
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
      x_issue(xs, p0 + (more ? q + 1 : q), more ? 1 : 0);   // unconditional; the last phase's is one address
#pragma unroll
      for (int lg = 0; lg < GL; ++lg) {
        load_unit(ring[(pp * GL + lg + D - 1) % D], q * GL + lg + D - 1);
        compute(ring[(pp * GL + lg) % D], q & 1, lg * KW + kq);
      }
      if (more) {
        x_commit(xs, (q + 1) & 1);
        __syncthreads();
      }
    }
  }

  // ---- the KW partial accumulators of a tile group meet in LDS (the x buffers are dead), one wave of
  // the group per round, added in wave order: [tile group][tile][m-tile][r][lane]
  float* red = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int j = 1; j < KW; ++j) {
    __syncthreads();
    if (kq == j) {
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(((tg * TPW + t) * MT + m) * 4 + r) * 64 + lane] = acc[t][m][r];
    }
    __syncthreads();
    if (kq == 0) {
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][m][r] += red[(((tg * TPW + t) * MT + m) * 4 + r) * 64 + lane];
    }
  }
  if (kq != 0 || tile_raw >= tiles_total) return;

  // ---- epilogue: lane holds D[n = 4h + r][m = c] of each m-tile, for the four tiles
  if (W8) {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const float4 sc = MI355_LD(float4, w_scale, w_scale + min(tile_raw + t, tiles_total - 1) * 16 + 4 * h);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        acc[t][m][0] = pin_f32(acc[t][m][0] * sc.x);
        acc[t][m][1] = pin_f32(acc[t][m][1] * sc.y);
        acc[t][m][2] = pin_f32(acc[t][m][2] * sc.z);
        acc[t][m][3] = pin_f32(acc[t][m][3] * sc.w);
      }
    }
  }
  if (EPI == EPI_SILU_MUL) {           // tiles (0, 1) and (2, 3): gate / up rows of the same 16 columns
    const int F = N >> 1;
#pragma unroll
    for (int pr = 0; pr < TPW / 2; ++pr) {
      if (tile_raw + 2 * pr + 1 >= tiles_total) continue;
      const int j0 = ((tile_raw >> 1) + pr) * 16 + 4 * h;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int mm = 16 * m + c;
        if (mm >= M) continue;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = silu_mul<T>(round_f<T>(acc[2 * pr][m][r]), round_f<T>(acc[2 * pr + 1][m][r]));
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
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    if (tile_raw + t >= tiles_total) continue;
    const int n0 = (tile_raw + t) * 16 + 4 * h;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_STORE && bias != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n0 + r < N) bv[r] = T::to_f(MI355_LD(uint16_t, bias, bias + (n0 + r)));
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int mm = 16 * m + c;
      if (mm >= M) continue;
      if (EPI == EPI_PARTIAL) {
        float* pp = partial + (static_cast<int64_t>(kgroup) * M_all + mm) * N + n0;
        if (n0 + 3 < N && (N & 3) == 0) {
          MI355_ST(float4, partial, pp, make_float4(acc[t][m][0], acc[t][m][1], acc[t][m][2], acc[t][m][3]));
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n0 + r < N) MI355_ST(float, partial, pp + r, acc[t][m][r]);
        }
      } else {
        uint16_t* yp = y + static_cast<int64_t>(mm) * y_stride + n0;
        if (n0 + 3 < N && (y_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0) {
          MI355_ST(uint2, y, yp, make_uint2(
              static_cast<uint32_t>(T::from_f(acc[t][m][0] + bv[0])) | (static_cast<uint32_t>(T::from_f(acc[t][m][1] + bv[1])) << 16),
              static_cast<uint32_t>(T::from_f(acc[t][m][2] + bv[2])) | (static_cast<uint32_t>(T::from_f(acc[t][m][3] + bv[3])) << 16)));
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n0 + r < N) MI355_ST(uint16_t, y, yp + r, T::from_f(acc[t][m][r] + bv[r]));
        }
      }
    }
  }
}

