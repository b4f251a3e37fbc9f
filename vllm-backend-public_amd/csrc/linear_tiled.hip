// Decode-time linear layer on FRAGMENT-TILED weights:  y[M, N] = x[M, K] . W[N, K]^T, M <= 256
// (a workgroup covers up to 128 rows; 129 .. 256 rows run as two row blocks of one launch).
//
// Weight layout (built once at load time, mi355_tile_weights): W is cut into
// blocks of 16 rows x 32 k; block (tile, kstep) is 1 KB stored as
// [lane 0..63][8 elements] with lane (i = lane & 15, h = lane >> 4) holding
// W[16*tile + i][32*kstep + 8h .. +8], blocks ordered [kstep / 4][tile][kstep % 4]
// (K a multiple of 128).  One wave load instruction therefore reads 1 KB of
// CONTIGUOUS memory that is already the v_mfma_f32_16x16x32 A operand, a wave's
// register group is 4 KB of contiguous memory, and the waves of a launch, all
// near the same k, sweep memory front to back together (fragment loads from
// row-major weights touch 16 rows x 64 B per instruction and stop ~15 % lower;
// per-tile contiguous streams [tile][kstep] stop ~10 % lower).
//
// Work decomposition.  A workgroup is NW waves (4 or 8) and owns NW consecutive
// 16-row tiles x one K group; WAVE w streams the blocks of tile w, so nothing is
// ever summed across waves.  x is staged
// through LDS one KSL-wide slice (a "phase") at a time, double buffered: the
// global loads of slice p+1 are issued at the start of phase p and written to the
// other LDS buffer at its end, one workgroup barrier per phase.  Weight loads run
// through a ring of 2 register groups of 4 k-steps (4 KB each), one group
// ahead of the MFMAs; scheduling barriers pin loads where they are written (the
// compiler otherwise sinks every load down to its use and the wave ends up
// waiting for one load at a time).  B operands are ds_read_b128 from LDS rows
// padded by 32 bytes (conflict-free).  Epilogues:
//   EPI_STORE     y = round(sum + bias)
//   EPI_SILU_MUL  tiles alternate gate/up rows of the same 16 columns (waves 2u,
//                 2u+1): act = round(round(silu(round(gate))) * round(up))
//   EPI_PARTIAL   fp32 partial sums to slab [kgroup][M][N] (K split over
//                 workgroups; the consumer kernel adds the slabs in order)
//   EPI_REDUCE_STORE / EPI_REDUCE_RESID   K split over workgroups, finished in the
//                 same launch: every workgroup writes its slab, then takes a ticket
//                 from a per-column-group counter; the workgroup that draws the last
//                 ticket adds the slabs in slab order (so the result does not depend
//                 on which one that is) and writes y = round(sum + bias), or
//                 residual = round(y + residual) together with the sum of squares of
//                 its columns of each residual row.  Nobody waits for anybody.
// Prologue NORM: x is not read as is but formed while it is staged,
//   x = round(round(residual * inv_rms[row]) * norm_weight), inv_rms from the sums of
//   squares the EPI_REDUCE_RESID epilogue of the previous GEMM left behind (added in
//   a fixed order) -- the RMSNorm between two GEMMs costs no launch of its own.
// HBM bytes: N*K*2 once; x is read from L2 once per workgroup.
#include "common.h"

namespace mi355 {

constexpr int kTU = 4;  // k-steps per register group
constexpr int kMaxRows = 256;  // rows of x a launch takes: one or two row blocks of up to 128

enum { EPI_STORE = 0, EPI_SILU_MUL = 1, EPI_PARTIAL = 2, EPI_REDUCE_STORE = 3, EPI_REDUCE_RESID = 4 };

constexpr int kSumsqRows = 32;  // row pitch of the sum-of-squares tables ([column group][32])

struct FusedArgs {
  const uint16_t* norm_w;    // prologue NORM: [K]
  const float* sumsq_in;     //   [norm_parts][32] partial sums of squares of the rows of x
  int norm_parts;
  float eps;
  uint16_t* resid;           // EPI_REDUCE_RESID: [M][N] in/out
  int64_t resid_stride;
  float* sumsq_out;          //   [gridDim.x][32]
  unsigned* counters;        // EPI_REDUCE_*: [gridDim.x], zero before the first launch, left zero
  const float* w_scale;      // W8: [tiles * 16] fp32, one scale per weight row in tiled row order
};

// Row blocks (gridDim.z): a launch covers more rows than a workgroup's ROWS (129 .. 256 rows of a
// wide decode batch) by giving workgroup (.., .., z) rows [z * ROWS, z * ROWS + ROWS): x, y and the
// slab rows move to the block's first row, M becomes the rows of this block, M_all stays the slab
// pitch.  The row blocks of a column block stream the same weights at the same time; the second
// read is served by the caches.
#define MI355_ROW_BLOCK()                                                   \
  const int M_all = M;                                                      \
  {                                                                         \
    const int m0 = static_cast<int>(blockIdx.z) * ROWS;                     \
    x += static_cast<int64_t>(m0) * x_stride;                               \
    if (y != nullptr) y += static_cast<int64_t>(m0) * y_stride;             \
    if (partial != nullptr) partial += static_cast<int64_t>(m0) * N;        \
    M = min(M - m0, ROWS);                                                  \
  }

template <typename T, int MT, int NW, int KSL, int EPI, bool NORM, bool W8 = false>
__global__ __launch_bounds__(NW * 64) void linear_tiled_kernel(
    uint16_t* __restrict__ y, float* __restrict__ partial, const uint16_t* __restrict__ x,
    const uint16_t* __restrict__ wt, const uint16_t* __restrict__ bias, int M, int N, int K,
    int64_t x_stride, int64_t y_stride, int tiles_total, int phases_total, FusedArgs fa) {
  constexpr int ROWS = MT * 16;
  constexpr int U = kTU;
  constexpr int NT = NW * 64;
  constexpr int ROWB = KSL * 2 + 32;       // padded LDS row (bytes)
  constexpr int XBUF = ROWS * ROWB;
  constexpr int G = KSL / 32 / U;          // register groups per phase (8, 4 or 2)
  constexpr int D = 2;                     // ring depth: one group computing, one in flight; deeper rings measured no faster (more outstanding requests only lengthen the memory queues) and cost registers
  constexpr int PU = G >= D ? 1 : D / G;   // phases per main-loop iteration: ring slots stay compile-time
  const int ubase = phases_total / static_cast<int>(gridDim.y), urem = phases_total % static_cast<int>(gridDim.y);
  const int P = ubase + (static_cast<int>(blockIdx.y) < urem ? 1 : 0);
  const int p0 = static_cast<int>(blockIdx.y) * ubase + min(static_cast<int>(blockIdx.y), urem);
  static_assert((PU * G) % D == 0, "ring slots must repeat every main-loop iteration");
  static_assert(!NORM || MT <= 2, "the norm prologue is built for M <= 32");
  static_assert(EPI != EPI_REDUCE_RESID || MT <= 2, "the sum-of-squares table has 32 rows");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * XBUF];
  __shared__ float inv_s[NORM ? ROWS : 1];
  __shared__ unsigned last_s;
  static_assert(NT % (KSL / 8) == 0, "a thread must keep one column of the x slice");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, h = lane >> 4;
  const int tile_raw = blockIdx.x * NW + wave;
  const int tile = min(tile_raw, tiles_total - 1);
  const int kgroup = blockIdx.y;
  // K range of this workgroup: the slices are dealt to the K groups as evenly as they go, the
  // first `rem` groups taking one more -- any group count up to the slice count works
  MI355_ROW_BLOCK();

  // ---- x slice staging: 16-byte vectors, whole rows, coalesced
  constexpr int NX = ROWS * (KSL / 8) / NT;
  typedef unsigned int xvec __attribute__((ext_vector_type(4)));   // native vector: stays in registers
  // xs[NX] holds the slice; with NORM xs[NX] one more vector: the norm weights of this thread's column
  constexpr int NXA = NX + (NORM ? 1 : 0);
  // live = 0 (wave-uniform): a load nobody will use (the last phase has no next slice, but a load inside a branch
  // would make the compiler drain the queue at the join): every lane then asks for the SAME 16 bytes -- one
  // cache request per instruction instead of 64 KB of x per workgroup
  auto x_issue = [&](xvec (&xs)[NXA], int p, int live) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int v = i * NT + tid;
      const int row = v / (KSL / 8), col = v % (KSL / 8);
      xs[i] = MI355_LD(xvec, x, x + (static_cast<int64_t>(min(row, M - 1)) * x_stride + col * 8) * live +
                                    static_cast<int64_t>(p) * KSL);
    }
    if (NORM)
      xs[NXA - 1] = MI355_LD(xvec, fa.norm_w, fa.norm_w + static_cast<int64_t>(p) * KSL + (tid % (KSL / 8)) * 8);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto x_commit = [&](xvec (&xs)[NXA], int buf) {
    float wn[8];
    if (NORM) unpack8<T>(make_uint4(xs[NXA - 1].x, xs[NXA - 1].y, xs[NXA - 1].z, xs[NXA - 1].w), wn);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int v = i * NT + tid;
      const int row = v / (KSL / 8), col = v % (KSL / 8);
      xvec o = xs[i];
      if (NORM) {
        float f[8];
        unpack8<T>(make_uint4(o.x, o.y, o.z, o.w), f);
        const float inv = inv_s[row];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = round_f<T>(f[j] * inv) * wn[j];
        const uint4 q = pack8<T>(f);
        o = xvec{q.x, q.y, q.z, q.w};
      }
      *reinterpret_cast<xvec*>(&lds[buf * XBUF + row * ROWB + col * 16]) = o;
    }
  };

  // ---- weight stream of this wave: 1 KB blocks ordered [k group][tile][block in group], a
  // group being the U k-steps of one register group (4 KB; W8: 2 KB).  At any moment the
  // waves of a launch are all near the same k group, so together they sweep one contiguous
  // region of memory -- with each tile's stream contiguous instead ([tile][k-step]) every wave
  // reads its own distant stream and the DRAM pages are opened for 1 KB at a time
  // (measured: gate_up 39.8 vs 44.0 us, lm_head 163 vs 181 us).
  // W8: fp8 weights, 1 KB blocks of 16 rows x 64 k (a lane's 16 bytes are its 8 elements of two
  // consecutive k-steps), so a group of U k-steps is U/2 loads and half the bytes
  constexpr int LPG = W8 ? U / 2 : U;      // 1 KB loads per register group
  const uint16_t* wbase = wt + (static_cast<int64_t>(p0) * G * tiles_total + tile) * (LPG * 512);
  const int64_t gstride = static_cast<int64_t>(tiles_total) * (LPG * 512);   // from one k group to the next
  const int groups_total = P * G;
  auto load_group = [&](uint4 (&wd)[LPG], int g) {  // g clamped: the tail asks for the last group's first 16 bytes (never used)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const uint16_t* p = wbase + min(g, groups_total - 1) * gstride + (g < groups_total ? lane * 8 : 0);
#pragma unroll
    for (int u = 0; u < LPG; ++u) {
      const u32x4 v = MI355_LDNT(u32x4, wt, p + u * 512);
      wd[u] = make_uint4(v.x, v.y, v.z, v.w);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // 8 fp8 (two dwords) -> the 8-element MFMA operand in the model's 16-bit type: four
  // v_cvt_scalef32_pk_*_fp8 with scale 1 (exact: every e4m3 value fits both 16-bit types)
  auto dequant8 = [&](uint32_t lo, uint32_t hi) {
    return make_uint4(T::fp8x2(lo, false), T::fp8x2(lo, true), T::fp8x2(hi, false), T::fp8x2(hi, true));
  };

  f32x4 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](uint4 (&wd)[LPG], int buf, int g) {  // g = group index inside the phase
    const unsigned char* xb = &lds[buf * XBUF + c * ROWB + g * (U * 64) + 16 * h];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint4 a;
      if (W8) {
        const uint4 v = wd[u / 2];
        a = (u & 1) ? dequant8(v.z, v.w) : dequant8(v.x, v.y);
      } else {
        a = wd[u];
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
        acc[m] = T::mfma(a, *reinterpret_cast<const uint4*>(xb + u * 64 + m * 16 * ROWB), acc[m]);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  uint4 ring[D][LPG];
#pragma unroll
  for (int j = 0; j < D - 1; ++j) load_group(ring[j], j);
  {
    xvec xs0[NXA];
    x_issue(xs0, p0, 1);
    if (NORM) {
      // 1/rms of every row from the partial sums of squares: 8 threads per row, each adds
      // every 8th part (8 loads in flight at a time), then a fixed-order butterfly
      if (tid < ROWS * 8) {
        const int row = tid >> 3, j = tid & 7;
        float ss = 0.f;
        if (row < M) {
          for (int q0 = 0; q0 < fa.norm_parts; q0 += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int q = q0 + 8 * u + j;
              v[u] = q < fa.norm_parts ? MI355_LD(float, fa.sumsq_in, fa.sumsq_in + (q * kSumsqRows + row)) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) ss += v[u];
          }
        }
        ss += __shfl_xor(ss, 1);
        ss += __shfl_xor(ss, 2);
        ss += __shfl_xor(ss, 4);
        if (j == 0) inv_s[row] = row < M ? 1.0f / sqrtf(ss / static_cast<float>(K) + fa.eps) : 0.f;
      }
      __syncthreads();
    }
    x_commit(xs0, 0);
  }
  __syncthreads();
  for (int p = 0; p < P; p += PU) {        // every iteration starts at ring slot 0 ((PU * G) % D == 0)
#pragma unroll
    for (int pp = 0; pp < PU; ++pp) {
      const int q = p + pp;
      if (q >= P) break;                     // P need not be a multiple of PU
      const bool more = q + 1 < P;
      xvec xs[NXA];                          // lives inside one phase only
      x_issue(xs, p0 + (more ? q + 1 : q), more ? 1 : 0);  // unconditional; the last phase's is one address
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

  // ---- epilogue: lane holds D[n = 4h + r][m = c] of each m-tile
  const int n0 = tile_raw * 16 + 4 * h;
  const bool live = tile_raw < tiles_total;
  if (W8) {   // per-row weight scales, stored in tiled row order (so interleaved gate/up tiles find theirs)
    const float4 sc = MI355_LD(float4, fa.w_scale, fa.w_scale + tile * 16 + 4 * h);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      acc[m][0] = pin_f32(acc[m][0] * sc.x);   // an fp32 value of its own before any 16-bit rounding
      acc[m][1] = pin_f32(acc[m][1] * sc.y);
      acc[m][2] = pin_f32(acc[m][2] * sc.z);
      acc[m][3] = pin_f32(acc[m][3] * sc.w);
    }
  }
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
        o[r] = silu_mul<T>(g, u);
      }
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
  if (EPI == EPI_REDUCE_STORE || EPI == EPI_REDUCE_RESID) {
    // The slabs cross workgroups (and XCDs, whose L2s are not coherent with each other)
    // inside one launch.  They are written and read with device-scope atomic accesses,
    // which go past the non-coherent cache levels one access at a time (a device-scope
    // fence executed by every wave around plain accesses measured +35 us); ONE lane per
    // workgroup still fences at device scope around the ticket, see below.
    const int kgs = gridDim.y;
    if (live) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int mm = 16 * m + c;
        if (mm >= M) continue;
        float* pp = partial + (static_cast<int64_t>(kgroup) * M_all + mm) * N + n0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n0 + r < N && MI355_OK_AT(partial, pp + r, 4))
            __hip_atomic_store(pp + r, acc[m][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // Hand-off to the workgroup that will sum the slabs (it may run on another XCD): every wave
    // waits for its slab stores, then ONE lane makes them visible device-wide (agent-scope
    // release) before it draws the ticket; the workgroup that draws the last ticket acquires at
    // agent scope before anyone in it reads a slab.  (With only a workgroup-scope fence before
    // the ticket the last workgroup now and then summed a slab that had not arrived yet: one
    // 4-column strip wrong in about one run of the test suite in five.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      last_s = MI355_OK_AT(fa.counters, fa.counters + blockIdx.x, 4) &&
               __hip_atomic_fetch_add(&fa.counters[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                   static_cast<unsigned>(kgs - 1);
      if (last_s) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    if (!last_s) return;
    if (tid == 0) MI355_ST(unsigned, fa.counters, fa.counters + blockIdx.x, 0u);   // all tickets drawn: re-arm for the next launch
    // this workgroup's NW*16 columns, 8 per thread, NW*2 threads per row, NT / (NW*2) = 32 rows per pass
    constexpr int VPR = NW * 2;
    const int col = blockIdx.x * (NW * 16) + (tid % VPR) * 8;
    for (int row = tid / VPR; row < ROWS; row += NT / VPR) {   // same trip count for every lane (shuffles below)
      const bool on = row < M && col < N;       // N % 8 == 0 (checked on the host)
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = 0.f;
      float ss = 0.f;
      if (on) {
        for (int s = 0; s < kgs; ++s) {
          const float* p = partial + (static_cast<int64_t>(s) * M_all + row) * N + col;
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j)
            v[j] = MI355_OK_AT(partial, p + j, 4) ? __hip_atomic_load(p + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] += v[j];
        }
        if (bias != nullptr) {
          float bb[8];
          unpack8<T>(MI355_LD(uint4, bias, bias + col), bb);
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] += bb[j];
        }
        if (EPI == EPI_REDUCE_RESID) {
          uint4* rp = reinterpret_cast<uint4*>(fa.resid + static_cast<int64_t>(row) * fa.resid_stride + col);
          float r[8];
          unpack8<T>(MI355_LD(uint4, fa.resid, rp), r);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            f[j] = round_f<T>(round_f<T>(f[j]) + r[j]);
            ss += f[j] * f[j];
          }
          MI355_ST(uint4, fa.resid, rp, pack8<T>(f));
        } else {
          MI355_ST(uint4, y, y + static_cast<int64_t>(row) * y_stride + col, pack8<T>(f));
        }
      }
      if (EPI == EPI_REDUCE_RESID) {
#pragma unroll
        for (int d = 1; d < VPR; d <<= 1) ss += __shfl_xor(ss, d);
        if (row < M && tid % VPR == 0) MI355_ST(float, fa.sumsq_out, fa.sumsq_out + (blockIdx.x * kSumsqRows + row), ss);
      }
    }
    return;
  }
  if (!live) return;
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
        MI355_ST(float4, partial, pp, make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n0 + r < N) MI355_ST(float, partial, pp + r, acc[m][r]);
      }
    } else {
      uint16_t* yp = y + static_cast<int64_t>(mm) * y_stride + n0;
      if (n0 + 3 < N && (y_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0) {
        MI355_ST(uint2, y, yp, make_uint2(
            static_cast<uint32_t>(T::from_f(acc[m][0] + bv[0])) | (static_cast<uint32_t>(T::from_f(acc[m][1] + bv[1])) << 16),
            static_cast<uint32_t>(T::from_f(acc[m][2] + bv[2])) | (static_cast<uint32_t>(T::from_f(acc[m][3] + bv[3])) << 16)));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n0 + r < N) MI355_ST(uint16_t, y, yp + r, T::from_f(acc[m][r] + bv[r]));
      }
    }
  }
}

// ---- two tiles per wave (33..128 rows).  At these row counts a k-step of one tile is 4 or 8 MFMAs,
// each fed by a 1 KB B fragment read from LDS, and with one tile per wave every wave of the
// workgroup reads the whole x slice again: the LDS reads, not the weight stream, bound the kernel
// (64 rows: 512 KB of fragments per 256 columns of x per CU).  Here a wave streams TWO adjacent
// tiles (8 KB of contiguous memory per register group) and uses every fragment for both, which
// halves the LDS reads per weight byte.  Adjacent tiles are the gate / up pair of an interleaved
// SwiGLU matrix, so that epilogue needs no exchange between waves.  Epilogues: store, SwiGLU,
// partial slabs (no norm prologue, no in-launch reduction).
template <typename T, int MT, int NW, int KSL, int EPI, bool W8>
__global__ __launch_bounds__(NW * 64) void linear_tiled2_kernel(
    uint16_t* __restrict__ y, float* __restrict__ partial, const uint16_t* __restrict__ x,
    const uint16_t* __restrict__ wt, const uint16_t* __restrict__ bias, int M, int N, int K,
    int64_t x_stride, int64_t y_stride, int tiles_total, int phases_total, const float* __restrict__ w_scale) {
  constexpr int TPW = 2;
  constexpr int ROWS = MT * 16;
  constexpr int U = kTU;
  constexpr int NT = NW * 64;
  constexpr int ROWB = KSL * 2 + 32;
  constexpr int XBUF = ROWS * ROWB;
  constexpr int G = KSL / 32 / U;
  constexpr int D = 2;
  constexpr int PU = G >= D ? 1 : D / G;
  static_assert((PU * G) % D == 0, "ring slots must repeat every main-loop iteration");
  static_assert(NT % (KSL / 8) == 0, "a thread must keep one column of the x slice");
  const int ubase = phases_total / static_cast<int>(gridDim.y), urem = phases_total % static_cast<int>(gridDim.y);
  const int P = ubase + (static_cast<int>(blockIdx.y) < urem ? 1 : 0);
  const int p0 = static_cast<int>(blockIdx.y) * ubase + min(static_cast<int>(blockIdx.y), urem);
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * XBUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, h = lane >> 4;
  const int tile_raw = (blockIdx.x * NW + wave) * TPW;            // first of the wave's two tiles
  // both tiles clamped into the matrix; an odd tile count leaves the last wave's second tile a copy (never stored)
  const int tile0 = min(tile_raw, tiles_total - 1), tile1 = min(tile_raw + 1, tiles_total - 1);
  const int kgroup = blockIdx.y;
  MI355_ROW_BLOCK();

  // 5..7 waves do not divide the slice's 16-byte vectors evenly: the last round's surplus threads
  // repeat the slice's last vector (the same bytes to the same LDS address)
  constexpr int NXV = ROWS * (KSL / 8);
  constexpr int NX = (NXV + NT - 1) / NT;
  typedef unsigned int xvec __attribute__((ext_vector_type(4)));
  auto x_issue = [&](xvec (&xs)[NX], int p, int live) {   // live = 0: a load nobody uses, one address for all lanes
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int v = NXV % NT == 0 ? i * NT + tid : min(i * NT + tid, NXV - 1);
      const int row = v / (KSL / 8), col = v % (KSL / 8);
      xs[i] = MI355_LD(xvec, x, x + (static_cast<int64_t>(min(row, M - 1)) * x_stride + col * 8) * live +
                                    static_cast<int64_t>(p) * KSL);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto x_commit = [&](xvec (&xs)[NX], int buf) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int v = NXV % NT == 0 ? i * NT + tid : min(i * NT + tid, NXV - 1);
      const int row = v / (KSL / 8), col = v % (KSL / 8);
      *reinterpret_cast<xvec*>(&lds[buf * XBUF + row * ROWB + col * 16]) = xs[i];
    }
  };

  constexpr int LPG = W8 ? U / 2 : U;
  const int64_t gstride = static_cast<int64_t>(tiles_total) * (LPG * 512);
  const uint16_t* wbase0 = wt + (static_cast<int64_t>(p0) * G * tiles_total + tile0) * (LPG * 512);
  const uint16_t* wbase1 = wt + (static_cast<int64_t>(p0) * G * tiles_total + tile1) * (LPG * 512);
  const int groups_total = P * G;
  auto load_group = [&](uint4 (&wd)[TPW][LPG], int g) {   // beyond the last group: its first 16 bytes for every lane (never used)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int64_t off = min(g, groups_total - 1) * gstride + (g < groups_total ? lane * 8 : 0);
#pragma unroll
    for (int u = 0; u < LPG; ++u) {
      const u32x4 v0 = MI355_LDNT(u32x4, wt, wbase0 + off + u * 512);
      wd[0][u] = make_uint4(v0.x, v0.y, v0.z, v0.w);
    }
#pragma unroll
    for (int u = 0; u < LPG; ++u) {
      const u32x4 v1 = MI355_LDNT(u32x4, wt, wbase1 + off + u * 512);
      wd[1][u] = make_uint4(v1.x, v1.y, v1.z, v1.w);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto dequant8 = [&](uint32_t lo, uint32_t hi) {
    return make_uint4(T::fp8x2(lo, false), T::fp8x2(lo, true), T::fp8x2(hi, false), T::fp8x2(hi, true));
  };

  f32x4 acc[TPW][MT];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](uint4 (&wd)[TPW][LPG], int buf, int g) {
    const unsigned char* xb = &lds[buf * XBUF + c * ROWB + g * (U * 64) + 16 * h];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint4 a[TPW];
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        if (W8) {
          const uint4 v = wd[j][u / 2];
          a[j] = (u & 1) ? dequant8(v.z, v.w) : dequant8(v.x, v.y);
        } else {
          a[j] = wd[j][u];
        }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const uint4 b = *reinterpret_cast<const uint4*>(xb + u * 64 + m * 16 * ROWB);   // one fragment, both tiles
#pragma unroll
        for (int j = 0; j < TPW; ++j) acc[j][m] = T::mfma(a[j], b, acc[j][m]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  uint4 ring[D][TPW][LPG];
#pragma unroll
  for (int j = 0; j < D - 1; ++j) load_group(ring[j], j);
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
      x_issue(xs, p0 + (more ? q + 1 : q), more ? 1 : 0);
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

  // ---- epilogue: lane holds D[n = 4h + r][m = c] of each m-tile, for both tiles
  if (W8) {
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const float4 sc = MI355_LD(float4, w_scale, w_scale + (j ? tile1 : tile0) * 16 + 4 * h);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        acc[j][m][0] = pin_f32(acc[j][m][0] * sc.x);
        acc[j][m][1] = pin_f32(acc[j][m][1] * sc.y);
        acc[j][m][2] = pin_f32(acc[j][m][2] * sc.z);
        acc[j][m][3] = pin_f32(acc[j][m][3] * sc.w);
      }
    }
  }
  if (EPI == EPI_SILU_MUL) {           // tile_raw is even: tile0 = gate rows, tile1 = up rows of the same 16 columns
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

// ---- K split across the WAVES of a workgroup: for matrices with few 16-row tiles
// (the slices of a tensor-parallel rank: 80 tiles for the qkv projection of Llama-3-70B on 8 GPUs)
// one wave per tile leaves most of the chip idle, and a K split across workgroups alone needs fp32
// slabs and a consumer that adds them -- which a SwiGLU epilogue or a 16-bit output for an
// all-reduce rules out.  Here a workgroup of 8 waves owns 8 / KW tiles; wave (tile, kq) streams
// every KW-th register group of its tile's K range, the KW partial accumulators meet in LDS and
// are added in wave order (a fixed order: the result does not depend on timing), and the tile's
// lead wave runs the epilogue: store (+ bias), SwiGLU (the gate and up tiles of a column block sit
// in one workgroup, KW <= 4) or fp32 slabs when K is ALSO split across workgroups (gridDim.y).
// x staging, weight ring and MFMA mapping are those of linear_tiled_kernel.
template <typename T, int MT, int KW, int KSL, int EPI>
__global__ __launch_bounds__(512) void linear_tiled_kw_kernel(
    uint16_t* __restrict__ y, float* __restrict__ partial, const uint16_t* __restrict__ x,
    const uint16_t* __restrict__ wt, const uint16_t* __restrict__ bias, int M, int N, int K,
    int64_t x_stride, int64_t y_stride, int tiles_total, int phases_total) {
  constexpr int NW = 8, NT = NW * 64, ROWS = MT * 16, U = kTU;
  constexpr int TPG = NW / KW;                 // tiles per workgroup
  constexpr int ROWB = KSL * 2 + 32;           // padded LDS row (bytes)
  constexpr int XBUF = ROWS * ROWB;
  constexpr int G = KSL / 32 / U;              // register groups per phase
  constexpr int GL = G / KW;                   // ... of which this wave takes GL
  constexpr int D = 2;
  constexpr int PU = GL >= D ? 1 : D / GL;     // phases per main-loop iteration: ring slots stay compile-time
  static_assert(G % KW == 0 && GL >= 1 && (PU * GL) % D == 0, "ring slots must repeat every main-loop iteration");
  static_assert(EPI != EPI_SILU_MUL || TPG % 2 == 0, "gate and up tiles of a column block share a workgroup");
  static_assert(2 * XBUF >= NW * MT * 4 * 64 * 4, "the reduction reuses the x buffers");
  static_assert(NT % (KSL / 8) == 0, "a thread must keep one column of the x slice");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * XBUF];

  const int ubase = phases_total / static_cast<int>(gridDim.y), urem = phases_total % static_cast<int>(gridDim.y);
  const int P = ubase + (static_cast<int>(blockIdx.y) < urem ? 1 : 0);
  const int p0 = static_cast<int>(blockIdx.y) * ubase + min(static_cast<int>(blockIdx.y), urem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = wave / KW, kq = wave % KW;
  const int c = lane & 15, h = lane >> 4;
  const int tile_raw = blockIdx.x * TPG + tl;
  const int tile = min(tile_raw, tiles_total - 1);
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

  // weight blocks are ordered [k group][tile][k-step in group]; this wave's ls-th group is group
  // (ls / GL) * G + (ls % GL) * KW + kq of the workgroup's K range
  const uint16_t* wbase = wt + (static_cast<int64_t>(p0) * G * tiles_total + tile) * (U * 512);
  const int64_t gstride = static_cast<int64_t>(tiles_total) * (U * 512);
  const int groups_total = P * GL;
  auto load_group = [&](uint4 (&wd)[U], int ls) {       // ls clamped: the tail asks for the last group's first 16 bytes (never used)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int l = min(ls, groups_total - 1);
    const uint16_t* p = wbase + ((l / GL) * G + (l % GL) * KW + kq) * gstride + (ls < groups_total ? lane * 8 : 0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const u32x4 v = MI355_LDNT(u32x4, wt, p + u * 512);
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
        load_group(ring[(pp * GL + lg + D - 1) % D], q * GL + lg + D - 1);
        compute(ring[(pp * GL + lg) % D], q & 1, lg * KW + kq);
      }
      if (more) {
        x_commit(xs, (q + 1) & 1);
        __syncthreads();
      }
    }
  }

  // ---- the partial accumulators of a tile meet in LDS (the x buffers are dead): [wave][m-tile][r][lane]
  __syncthreads();
  float* red = reinterpret_cast<float*>(lds);
  const bool lead = kq == 0 && (EPI != EPI_SILU_MUL || (tl & 1) == 0);
  if (!lead) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wave * MT + m) * 4 + r) * 64 + lane] = acc[m][r];
  }
  __syncthreads();
  if (!lead || tile_raw >= tiles_total) return;
#pragma unroll
  for (int j = 1; j < KW; ++j)                           // wave order: a fixed order of additions
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[m][r] += red[(((wave + j) * MT + m) * 4 + r) * 64 + lane];

  if (EPI == EPI_SILU_MUL) {                             // this wave: the gate tile; waves wave + KW ...: the up tile
    if (tile_raw + 1 >= tiles_total) return;
    const int F = N >> 1;
    const int j0 = (tile_raw >> 1) * 16 + 4 * h;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int mm = 16 * m + c;
      if (mm >= M) continue;
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float up = 0.f;
#pragma unroll
        for (int j = 0; j < KW; ++j) up += red[(((wave + KW + j) * MT + m) * 4 + r) * 64 + lane];
        o[r] = silu_mul<T>(round_f<T>(acc[m][r]), round_f<T>(up));
      }
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
  const int n0 = tile_raw * 16 + 4 * h;
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
        MI355_ST(float4, partial, pp, make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n0 + r < N) MI355_ST(float, partial, pp + r, acc[m][r]);
      }
    } else {
      uint16_t* yp = y + static_cast<int64_t>(mm) * y_stride + n0;
      if (n0 + 3 < N && (y_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0) {
        MI355_ST(uint2, y, yp, make_uint2(
            static_cast<uint32_t>(T::from_f(acc[m][0] + bv[0])) | (static_cast<uint32_t>(T::from_f(acc[m][1] + bv[1])) << 16),
            static_cast<uint32_t>(T::from_f(acc[m][2] + bv[2])) | (static_cast<uint32_t>(T::from_f(acc[m][3] + bv[3])) << 16)));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n0 + r < N) MI355_ST(uint16_t, y, yp + r, T::from_f(acc[m][r] + bv[r]));
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
  for (int64_t b = blockIdx.x; b < blocks; b += gridDim.x) {   // b: destination block, [k group][tile][k-step in group]
    const int ks = static_cast<int>(b / (static_cast<int64_t>(tiles_total) * kTU)) * kTU + static_cast<int>(b % kTU);
    const int tile = static_cast<int>(b / kTU % tiles_total);
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
    if (row >= 0) v = MI355_LD(uint4, src, src + static_cast<int64_t>(row) * K + 32 * ks + 8 * h);
    MI355_ST(uint4, dst, dst + (b * 64 + lane) * 8, v);
  }
}

// ---- fp8 (e4m3fn) weight copies: one fp32 scale per weight row, scale = max|row| / 448,
// q = fp8_rne(w / scale).  Pass 1 writes the scales in TILED row order (what the GEMM
// epilogue indexes); pass 2 tiles and quantises: block (tile, k-step pair) is 1 KB stored
// as [lane][16 bytes], lane (i, h) holding row i's elements [64p + 8h, +8) and
// [64p + 32 + 8h, +8) -- the lane's A operands of two consecutive k-steps.
__device__ __forceinline__ int tiled_source_row(int tile, int i, int N, int interleave_halves) {
  if (interleave_halves) {
    const int pair = tile >> 1;
    return pair * 16 + i < (N >> 1) ? (tile & 1) * (N >> 1) + pair * 16 + i : -1;
  }
  return tile * 16 + i < N ? tile * 16 + i : -1;
}

template <typename T>
__global__ __launch_bounds__(256) void fp8_row_scale_kernel(float* __restrict__ scale_t, const uint16_t* __restrict__ src,
                                                            int N, int K, int interleave_halves) {
  const int tr = blockIdx.x;   // tiled row index
  const int row = tiled_source_row(tr >> 4, tr & 15, N, interleave_halves);
  float amax = 0.f;
  if (row >= 0) {
    const uint4* p = reinterpret_cast<const uint4*>(src + static_cast<int64_t>(row) * K);
    for (int v = threadIdx.x; v < (K >> 3); v += 256) {
      float f[8];
      unpack8<T>(MI355_LD(uint4, src, p + v), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = amax;
  __syncthreads();
  if (threadIdx.x == 0) {
    amax = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    MI355_ST(float, scale_t, scale_t + tr, amax > 0.f ? amax / 448.0f : 1.0f);
  }
}

template <typename T>
__global__ __launch_bounds__(64) void tile_weights_fp8_kernel(unsigned char* __restrict__ dst, const float* __restrict__ scale_t,
                                                              const uint16_t* __restrict__ src, int N, int K,
                                                              int tiles_total, int interleave_halves) {
  const int64_t blocks = static_cast<int64_t>(tiles_total) * (K >> 6);
  const int lane = threadIdx.x, i = lane & 15, h = lane >> 4;
  constexpr int BPG = kTU / 2;                                  // 1 KB blocks (k-step pairs) per k group
  for (int64_t b = blockIdx.x; b < blocks; b += gridDim.x) {   // b: destination block, [k group][tile][pair in group]
    const int kp = static_cast<int>(b / (static_cast<int64_t>(tiles_total) * BPG)) * BPG + static_cast<int>(b % BPG);
    const int tile = static_cast<int>(b / BPG % tiles_total);
    const int row = tiled_source_row(tile, i, N, interleave_halves);
    uint32_t w[4] = {0, 0, 0, 0};
    if (row >= 0) {
      const float sc = MI355_LD(float, scale_t, scale_t + (tile * 16 + i));
      const uint16_t* p = src + static_cast<int64_t>(row) * K + 64 * kp + 8 * h;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float f[8];
        unpack8<T>(MI355_LD(uint4, src, p + 32 * half), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = f[j] / sc;
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
        w[2 * half] = static_cast<uint32_t>(lo);
        w[2 * half + 1] = static_cast<uint32_t>(hi);
      }
    }
    MI355_ST(uint4, dst, dst + (b * 64 + lane) * 16, make_uint4(w[0], w[1], w[2], w[3]));
  }
}

// ---- the inverse re-layouts: tiled blocks back to a row-major [N, K] matrix in the model's 16-bit
// type (the fp8 form decodes: round(fp8 * row scale), the matrix the tiled GEMM multiplies by up to
// its accumulation order).  An engine that keeps only the tiled copy of a weight rebuilds the
// row-major one into a scratch buffer when a library GEMM needs it.
template <int DUMMY>
__global__ __launch_bounds__(64) void untile_weights_kernel(uint16_t* __restrict__ dst, const uint16_t* __restrict__ src,
                                                            int N, int K, int tiles_total, int interleave_halves) {
  const int64_t blocks = static_cast<int64_t>(tiles_total) * (K >> 5);
  const int lane = threadIdx.x, i = lane & 15, h = lane >> 4;
  for (int64_t b = blockIdx.x; b < blocks; b += gridDim.x) {   // b: source block, [k group][tile][k-step in group]
    const int ks = static_cast<int>(b / (static_cast<int64_t>(tiles_total) * kTU)) * kTU + static_cast<int>(b % kTU);
    const int tile = static_cast<int>(b / kTU % tiles_total);
    const int row = tiled_source_row(tile, i, N, interleave_halves);
    if (row < 0) continue;
    const uint4 v = MI355_LD(uint4, src, src + (b * 64 + lane) * 8);
    MI355_ST(uint4, dst, dst + static_cast<int64_t>(row) * K + 32 * ks + 8 * h, v);
  }
}

template <typename T>
__global__ __launch_bounds__(64) void untile_weights_fp8_kernel(uint16_t* __restrict__ dst, const unsigned char* __restrict__ src,
                                                                const float* __restrict__ scale_t, int N, int K,
                                                                int tiles_total, int interleave_halves) {
  const int64_t blocks = static_cast<int64_t>(tiles_total) * (K >> 6);
  const int lane = threadIdx.x, i = lane & 15, h = lane >> 4;
  constexpr int BPG = kTU / 2;
  for (int64_t b = blockIdx.x; b < blocks; b += gridDim.x) {   // b: source block, [k group][tile][pair in group]
    const int kp = static_cast<int>(b / (static_cast<int64_t>(tiles_total) * BPG)) * BPG + static_cast<int>(b % BPG);
    const int tile = static_cast<int>(b / BPG % tiles_total);
    const int row = tiled_source_row(tile, i, N, interleave_halves);
    if (row < 0) continue;
    const float sc = MI355_LD(float, scale_t, scale_t + (tile * 16 + i));
    const uint4 w = MI355_LD(uint4, src, src + (b * 64 + lane) * 16);
    const uint32_t q[4] = {w.x, w.y, w.z, w.w};
    uint16_t* p = dst + static_cast<int64_t>(row) * K + 64 * kp + 8 * h;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const uint32_t lo = q[2 * half], hi = q[2 * half + 1];
      float f[8];
      unpack8<T>(make_uint4(T::fp8x2(lo, false), T::fp8x2(lo, true), T::fp8x2(hi, false), T::fp8x2(hi, true)), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] *= sc;
      MI355_ST(uint4, dst, p + 32 * half, pack8<T>(f));
    }
  }
}

#define MI355_LT_ARGS y, partial, x, wt, bias, M, N, K, xs, ys, tiles_total, phases
#define MI355_LT_PARAMS                                                                                       \
  uint16_t *y, float *partial, const uint16_t *x, const uint16_t *wt, const uint16_t *bias, int M, int N, int K, \
      int64_t xs, int64_t ys, int tiles_total, int phases

template <typename T, int MT, int NW, int KSL, int EPI, bool NORM, bool W8 = false>
static void launch_one(dim3 grid, hipStream_t s, MI355_LT_PARAMS, const FusedArgs& fa) {
  hipLaunchKernelGGL((linear_tiled_kernel<T, MT, NW, KSL, EPI, NORM, W8>), grid, dim3(NW * 64), 0, s, MI355_LT_ARGS, fa);
}

// Which (epilogue, prologue) combinations exist for a geometry.  PLAIN: epilogues 0-2 on
// plain x.  FUSED (the geometries the decode step uses): also the reducing epilogues, and
// for M <= 32 the norm prologue and the residual epilogue.
template <typename T, int MT, int NW, int KSL, bool FUSED>
static bool launch_cfg(int epi, bool norm, dim3 grid, hipStream_t s, MI355_LT_PARAMS, const FusedArgs& fa) {
  if (fa.w_scale != nullptr) {   // fp8 weights: epilogues 0-2 on plain x, the decode step's geometries
    if constexpr (FUSED) {
      if (norm) return false;
      if (epi == EPI_STORE) { launch_one<T, MT, NW, KSL, EPI_STORE, false, true>(grid, s, MI355_LT_ARGS, fa); return true; }
      if (epi == EPI_SILU_MUL) { launch_one<T, MT, NW, KSL, EPI_SILU_MUL, false, true>(grid, s, MI355_LT_ARGS, fa); return true; }
      if (epi == EPI_PARTIAL) { launch_one<T, MT, NW, KSL, EPI_PARTIAL, false, true>(grid, s, MI355_LT_ARGS, fa); return true; }
    }
    return false;
  }
  if (!norm) {
    if (epi == EPI_STORE) { launch_one<T, MT, NW, KSL, EPI_STORE, false>(grid, s, MI355_LT_ARGS, fa); return true; }
    if (epi == EPI_SILU_MUL) { launch_one<T, MT, NW, KSL, EPI_SILU_MUL, false>(grid, s, MI355_LT_ARGS, fa); return true; }
    if (epi == EPI_PARTIAL) { launch_one<T, MT, NW, KSL, EPI_PARTIAL, false>(grid, s, MI355_LT_ARGS, fa); return true; }
  }
  if constexpr (FUSED) {
    if (!norm && epi == EPI_REDUCE_STORE) {
      launch_one<T, MT, NW, KSL, EPI_REDUCE_STORE, false>(grid, s, MI355_LT_ARGS, fa);
      return true;
    }
    if constexpr (MT <= 2) {
      if (!norm && epi == EPI_REDUCE_RESID) {
        launch_one<T, MT, NW, KSL, EPI_REDUCE_RESID, false>(grid, s, MI355_LT_ARGS, fa);
        return true;
      }
      if (norm && epi == EPI_STORE) { launch_one<T, MT, NW, KSL, EPI_STORE, true>(grid, s, MI355_LT_ARGS, fa); return true; }
      if (norm && epi == EPI_SILU_MUL) { launch_one<T, MT, NW, KSL, EPI_SILU_MUL, true>(grid, s, MI355_LT_ARGS, fa); return true; }
      if (norm && epi == EPI_PARTIAL) { launch_one<T, MT, NW, KSL, EPI_PARTIAL, true>(grid, s, MI355_LT_ARGS, fa); return true; }
    }
  }
  return false;
}

// The instantiated geometries: (m-tiles, waves per workgroup, x slice width).
template <typename T>
static bool launch_geometry(int mt, int nw, int ksl, int epi, bool norm, dim3 grid, hipStream_t s, MI355_LT_PARAMS,
                            const FusedArgs& fa) {
#define MI355_LT_GEOM(MT_, NW_, KSL_, FUSED_)           \
  if (mt == MT_ && nw == NW_ && ksl == KSL_)            \
    return launch_cfg<T, MT_, NW_, KSL_, FUSED_>(epi, norm, grid, s, MI355_LT_ARGS, fa);
  MI355_LT_GEOM(1, 8, 1024, true) MI355_LT_GEOM(1, 4, 512, true) MI355_LT_GEOM(1, 4, 1024, false) MI355_LT_GEOM(1, 8, 512, false)
  MI355_LT_GEOM(2, 8, 1024, true) MI355_LT_GEOM(2, 4, 512, true) MI355_LT_GEOM(2, 4, 1024, false) MI355_LT_GEOM(2, 8, 512, false)
  MI355_LT_GEOM(4, 8, 512, true) MI355_LT_GEOM(4, 4, 256, true) MI355_LT_GEOM(4, 8, 256, true)
  MI355_LT_GEOM(8, 8, 256, true)
#undef MI355_LT_GEOM
  return false;
}

// Two tiles per wave (tiles_per_wg 208 .. 216): epilogues 0-2, 16-bit or fp8 weights, 33..128 rows.
template <typename T, int MT, int NW, int KSL>
static bool launch_cfg2(int epi, dim3 grid, hipStream_t s, MI355_LT_PARAMS, const float* w_scale) {
#define MI355_LT2(EPI_, W8_)                                                                                   \
  hipLaunchKernelGGL((linear_tiled2_kernel<T, MT, NW, KSL, EPI_, W8_>), grid, dim3(NW * 64), 0, s, MI355_LT_ARGS, \
                     w_scale)
  if (w_scale != nullptr) {
    if (epi == EPI_STORE) { MI355_LT2(EPI_STORE, true); return true; }
    if (epi == EPI_SILU_MUL) { MI355_LT2(EPI_SILU_MUL, true); return true; }
    if (epi == EPI_PARTIAL) { MI355_LT2(EPI_PARTIAL, true); return true; }
  } else {
    if (epi == EPI_STORE) { MI355_LT2(EPI_STORE, false); return true; }
    if (epi == EPI_SILU_MUL) { MI355_LT2(EPI_SILU_MUL, false); return true; }
    if (epi == EPI_PARTIAL) { MI355_LT2(EPI_PARTIAL, false); return true; }
  }
#undef MI355_LT2
  return false;
}

template <typename T>
static bool launch_geometry2(int mt, int nw, int ksl, int epi, dim3 grid, hipStream_t s, MI355_LT_PARAMS,
                             const float* w_scale) {
  if (mt == 4 && nw == 4 && ksl == 256) return launch_cfg2<T, 4, 4, 256>(epi, grid, s, MI355_LT_ARGS, w_scale);
  if (mt == 4 && nw == 8 && ksl == 512) return launch_cfg2<T, 4, 8, 512>(epi, grid, s, MI355_LT_ARGS, w_scale);
  if (mt == 8 && nw == 8 && ksl == 256) return launch_cfg2<T, 8, 8, 256>(epi, grid, s, MI355_LT_ARGS, w_scale);
  // 5 / 6 / 7 waves: for tile counts that 16 per workgroup would spread over too few CUs
  // (2368 tiles: 148 workgroups of 16 tiles, 237 of 10)
  if (mt == 4 && nw == 5 && ksl == 512) return launch_cfg2<T, 4, 5, 512>(epi, grid, s, MI355_LT_ARGS, w_scale);
  if (mt == 4 && nw == 6 && ksl == 512) return launch_cfg2<T, 4, 6, 512>(epi, grid, s, MI355_LT_ARGS, w_scale);
  if (mt == 4 && nw == 7 && ksl == 512) return launch_cfg2<T, 4, 7, 512>(epi, grid, s, MI355_LT_ARGS, w_scale);
  if (mt == 8 && nw == 5 && ksl == 256) return launch_cfg2<T, 8, 5, 256>(epi, grid, s, MI355_LT_ARGS, w_scale);
  if (mt == 8 && nw == 6 && ksl == 256) return launch_cfg2<T, 8, 6, 256>(epi, grid, s, MI355_LT_ARGS, w_scale);
  if (mt == 8 && nw == 7 && ksl == 256) return launch_cfg2<T, 8, 7, 256>(epi, grid, s, MI355_LT_ARGS, w_scale);
  return false;
}

// K split across waves (tiles_per_wg 302 / 304 / 308: wide x slices; 402 / 404: narrow ones): epilogues 0-2,
// 16-bit weights.
template <typename T, int MT, int KW, int KSL>
static bool launch_cfg_kw(int epi, dim3 grid, hipStream_t s, MI355_LT_PARAMS) {
#define MI355_LTKW(EPI_)                                                                                          \
  hipLaunchKernelGGL((linear_tiled_kw_kernel<T, MT, KW, KSL, EPI_>), grid, dim3(512), 0, s, MI355_LT_ARGS)
  if (epi == EPI_STORE) { MI355_LTKW(EPI_STORE); return true; }
  if (epi == EPI_PARTIAL) { MI355_LTKW(EPI_PARTIAL); return true; }
  if constexpr (KW <= 4) {
    if (epi == EPI_SILU_MUL) { MI355_LTKW(EPI_SILU_MUL); return true; }
  }
#undef MI355_LTKW
  return false;
}

template <typename T>
static bool launch_geometry_kw(int mt, int kw, int ksl, int epi, dim3 grid, hipStream_t s, MI355_LT_PARAMS) {
#define MI355_LTKW_GEOM(MT_, KW_, KSL_)             \
  if (mt == MT_ && kw == KW_ && ksl == KSL_) return launch_cfg_kw<T, MT_, KW_, KSL_>(epi, grid, s, MI355_LT_ARGS);
  MI355_LTKW_GEOM(1, 2, 1024) MI355_LTKW_GEOM(1, 4, 1024) MI355_LTKW_GEOM(1, 8, 1024)
  MI355_LTKW_GEOM(2, 2, 1024) MI355_LTKW_GEOM(2, 4, 1024) MI355_LTKW_GEOM(2, 8, 1024)
  MI355_LTKW_GEOM(1, 2, 512) MI355_LTKW_GEOM(1, 4, 512) MI355_LTKW_GEOM(2, 2, 512) MI355_LTKW_GEOM(2, 4, 512)
  // 33..128 rows: the x slice halves with every doubling of the rows (its double-buffered LDS image keeps
  // its size), and a wave needs at least one register group of it: 2 or 4 waves per tile at up to 64 rows, 2 beyond
  MI355_LTKW_GEOM(4, 2, 512) MI355_LTKW_GEOM(4, 4, 512) MI355_LTKW_GEOM(4, 2, 256) MI355_LTKW_GEOM(8, 2, 256)
#undef MI355_LTKW_GEOM
  return false;
}

// Shared argument checks + launch of both entry points.
static int32_t linear_tiled_launch(const char* name, void* y, int64_t y_stride, float* partial, const void* x,
                                   int64_t x_stride, const void* w_tiled, const void* bias, int32_t M, int32_t N,
                                   int32_t K, int32_t tiles_per_wg, int32_t k_groups, int32_t epilogue,
                                   const FusedArgs& fa, int32_t dtype, void* stream) {
  const bool norm = fa.norm_w != nullptr;
  const bool reduce = epilogue == EPI_REDUCE_STORE || epilogue == EPI_REDUCE_RESID;
  const int rb_code = tiles_per_wg / 1000;
  MI355_REQUIRE(tiles_per_wg >= 0 && rb_code <= 2 && (rb_code == 0 || (!norm && !reduce)), MI355_ERR_INVALID,
                "%s: tiles_per_wg %d: + 1000 / + 2000 (row blocks of 64 / 32) go with epilogues 0-2 on plain x", name,
                tiles_per_wg);
  const int rb_rows = rb_code == 1 ? 64 : rb_code == 2 ? 32 : 128;
  tiles_per_wg %= 1000;
  MI355_REQUIRE(x && w_tiled, MI355_ERR_INVALID, "%s: null pointer", name);
  MI355_REQUIRE(M >= 0 && M <= kMaxRows, MI355_ERR_UNSUPPORTED, "%s: M %d outside [0, %d]", name, M, kMaxRows);
  MI355_REQUIRE(M <= 128 || (!norm && !reduce), MI355_ERR_UNSUPPORTED,
                "%s: the norm prologue and the in-launch reduction are built for up to 128 rows (M %d)", name, M);
  MI355_REQUIRE(epilogue >= EPI_STORE && epilogue <= EPI_REDUCE_RESID, MI355_ERR_INVALID, "%s: unknown epilogue %d",
                name, epilogue);
  // tiles_per_wg: 4 or 8 tiles (= waves) per workgroup; +100 selects the narrow x slice.  The slice is
  // 1024 (narrow: 512) wide up to 32 rows and halves with every doubling of the rows, so that the
  // double-buffered x image keeps its LDS footprint.
  // 208 / 216: two tiles per wave (4 waves with the narrow slice / 8 waves with the wide one), 33..128 rows;
  // 210 / 212 / 214: 5 / 6 / 7 waves with the wide slice
  const bool two = tiles_per_wg >= 208 && tiles_per_wg <= 216 && tiles_per_wg % 2 == 0;
  // 302 / 304 / 308, 402 / 404: K split across 2 / 4 / 8 waves of an 8-wave workgroup (1024- / 512-wide slices)
  const bool kws = tiles_per_wg == 302 || tiles_per_wg == 304 || tiles_per_wg == 308 || tiles_per_wg == 402 ||
                   tiles_per_wg == 404;
  const int nw = two ? (tiles_per_wg - 200) / 2 : tiles_per_wg % 100;
  MI355_REQUIRE(two || kws || ((nw == 4 || nw == 8) && (tiles_per_wg == nw || tiles_per_wg == nw + 100)),
                MI355_ERR_INVALID, "%s: tiles_per_wg must be 4, 8, 104, 108, 208-216 (even), 302, 304, 308, 402 or 404", name);
  // Rows of a row block (gridDim.z): 128, or with tiles_per_wg + 1000 / + 2000 blocks of 64 / 32 rows, each run
  // by the kernel form built for that many rows (a wave's fragments of x from LDS per weight fragment halve with
  // the block; the blocks of a column block read the same weights, once from HBM).
  const int Mb = min(M, rb_rows);
  const int mt = Mb <= 16 ? 1 : Mb <= 32 ? 2 : Mb <= 64 ? 4 : 8;
  const int ksl = ((kws ? tiles_per_wg >= 400 : two ? tiles_per_wg == 208 : tiles_per_wg >= 100) ? 512 : 1024) /
                  (mt <= 2 ? 1 : mt / 2);
  MI355_REQUIRE(!kws || (!norm && !reduce && fa.w_scale == nullptr), MI355_ERR_UNSUPPORTED,
                "%s: the K split across waves (tiles_per_wg %d) is built for 16-bit weights and epilogues 0-2", name,
                tiles_per_wg);
  MI355_REQUIRE(!kws || (tiles_per_wg % 100) * 128 <= ksl, MI355_ERR_UNSUPPORTED,
                "%s: %d waves per tile need an x slice of at least %d columns, M %d stages %d (up to 64 rows: 302, 304, "
                "402; up to 128 rows: 302)", name, tiles_per_wg % 100, (tiles_per_wg % 100) * 128, M, ksl);
  MI355_REQUIRE(!kws || epilogue != EPI_SILU_MUL || tiles_per_wg % 100 <= 4, MI355_ERR_UNSUPPORTED,
                "%s: the SwiGLU epilogue needs the gate and up tiles in one workgroup: at most 4 waves per tile", name);
  MI355_REQUIRE(!two || (M > 32 && !norm && !reduce), MI355_ERR_UNSUPPORTED,
                "%s: two tiles per wave (tiles_per_wg %d) is built for 33 rows and more and epilogues 0-2", name, tiles_per_wg);
  MI355_REQUIRE(N > 0 && K > 0 && K % ksl == 0, MI355_ERR_UNSUPPORTED,
                "%s: K %d must be a multiple of the x slice %d (M %d)", name, K, ksl, M);
  MI355_REQUIRE(fa.w_scale == nullptr || (aligned16(fa.w_scale) && epilogue <= EPI_PARTIAL), MI355_ERR_INVALID,
                "%s: fp8 weights need a 16-byte aligned scale vector and epilogue 0-2", name);
  const int phases = K / ksl;
  MI355_REQUIRE(k_groups >= 1 && k_groups <= phases, MI355_ERR_INVALID, "%s: k_groups %d outside [1, %d] (K / %d)",
                name, k_groups, phases, ksl);
  MI355_REQUIRE(k_groups == 1 || epilogue == EPI_PARTIAL || reduce, MI355_ERR_INVALID,
                "%s: k_groups > 1 needs a partial-sum or reducing epilogue", name);
  MI355_REQUIRE(x_stride >= K && x_stride % 8 == 0 && aligned16(x) && aligned16(w_tiled), MI355_ERR_INVALID,
                "%s: x/w_tiled must be 16-byte aligned, x_stride a multiple of 8", name);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "%s: dtype %d not supported (bf16/fp16 only)", name, dtype);
  if (norm)
    MI355_REQUIRE(fa.sumsq_in && fa.norm_parts >= 1 && aligned16(fa.norm_w) && fa.eps >= 0.f && !reduce && M <= 32,
                  MI355_ERR_INVALID,
                  "%s: norm prologue needs norm_sumsq, norm_parts >= 1, a 16-byte aligned weight, M <= 32 and a "
                  "non-reducing epilogue", name);
  if (epilogue == EPI_PARTIAL || reduce)
    MI355_REQUIRE(partial && aligned16(partial), MI355_ERR_INVALID, "%s: partial buffer missing", name);
  if (reduce) {
    MI355_REQUIRE(fa.counters && N % 8 == 0, MI355_ERR_INVALID, "%s: reducing epilogues need counters and N %% 8 == 0",
                  name);
    MI355_REQUIRE(bias == nullptr || aligned16(bias), MI355_ERR_INVALID, "%s: bias must be 16-byte aligned", name);
  }
  if (epilogue == EPI_REDUCE_RESID)
    MI355_REQUIRE(fa.resid && fa.sumsq_out && aligned16(fa.resid) && fa.resid_stride >= N && fa.resid_stride % 8 == 0 &&
                      M <= 32,
                  MI355_ERR_INVALID,
                  "%s: residual (16-byte aligned, stride >= N, multiple of 8), sumsq_out and M <= 32 needed", name);
  if (epilogue == EPI_STORE || epilogue == EPI_SILU_MUL || epilogue == EPI_REDUCE_STORE)
    MI355_REQUIRE(y && y_stride >= (epilogue == EPI_SILU_MUL ? N / 2 : N) &&
                      (epilogue != EPI_REDUCE_STORE || (aligned16(y) && y_stride % 8 == 0)),
                  MI355_ERR_INVALID, "%s: y missing, misaligned or y_stride too small", name);
  if (epilogue == EPI_SILU_MUL)
    MI355_REQUIRE(N % 32 == 0 && bias == nullptr, MI355_ERR_UNSUPPORTED,
                  "%s: SwiGLU epilogue needs N %% 32 == 0 and no bias", name);
  if (M == 0) return MI355_OK;
  const int tiles_total = (N + 15) / 16;
  const int tiles_wg = kws ? 8 / (tiles_per_wg % 100) : two ? 2 * nw : nw;
  dim3 grid((tiles_total + tiles_wg - 1) / tiles_wg, k_groups, (M + rb_rows - 1) / rb_rows);
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto* yp = static_cast<uint16_t*>(y);
  auto* xp = static_cast<const uint16_t*>(x);
  auto* wp = static_cast<const uint16_t*>(w_tiled);
  auto* bp = static_cast<const uint16_t*>(bias);
  bool ok;
  if (kws)
    ok = dtype == MI355_BF16 ? launch_geometry_kw<BF16>(mt, tiles_per_wg % 100, ksl, epilogue, grid, s, yp, partial, xp, wp, bp,
                                                        M, N, K, x_stride, y_stride, tiles_total, phases)
                             : launch_geometry_kw<F16>(mt, tiles_per_wg % 100, ksl, epilogue, grid, s, yp, partial, xp, wp, bp,
                                                       M, N, K, x_stride, y_stride, tiles_total, phases);
  else if (two)
    ok = dtype == MI355_BF16 ? launch_geometry2<BF16>(mt, nw, ksl, epilogue, grid, s, yp, partial, xp, wp, bp, M, N, K,
                                                      x_stride, y_stride, tiles_total, phases, fa.w_scale)
                             : launch_geometry2<F16>(mt, nw, ksl, epilogue, grid, s, yp, partial, xp, wp, bp, M, N, K,
                                                     x_stride, y_stride, tiles_total, phases, fa.w_scale);
  else
    ok = dtype == MI355_BF16 ? launch_geometry<BF16>(mt, nw, ksl, epilogue, norm, grid, s, yp, partial, xp, wp, bp, M, N, K,
                                                     x_stride, y_stride, tiles_total, phases, fa)
                             : launch_geometry<F16>(mt, nw, ksl, epilogue, norm, grid, s, yp, partial, xp, wp, bp, M, N, K,
                                                    x_stride, y_stride, tiles_total, phases, fa);
  MI355_REQUIRE(ok, MI355_ERR_UNSUPPORTED,
                "%s: no kernel for M %d with tiles_per_wg %d, epilogue %d%s (M <= 32: 4/8/104/108/302/304/308/402/404, "
                "fused and fp8 forms 8/104; M <= 64: 8/104/108/208-216/302/304/402; beyond: 8/210-216/302)", name, M,
                tiles_per_wg, epilogue,
                norm ? " + norm prologue" : "");
  return check_launch(name);
}

}  // namespace mi355

extern "C" int64_t mi355_tiled_weight_elems(int32_t N, int32_t K) {
  if (N <= 0 || K <= 0 || K % 128) return 0;
  return static_cast<int64_t>((N + 15) / 16) * 16 * K;
}

extern "C" int32_t mi355_tile_weights(void* dst, const void* src, int32_t N, int32_t K,
                                      int32_t interleave_halves, void* stream) {
  using namespace mi355;
  const char* name = "mi355_tile_weights";
  MI355_REQUIRE(dst && src && dst != src, MI355_ERR_INVALID, "%s: null or aliased pointers", name);
  MI355_REQUIRE(N > 0 && K > 0 && K % 128 == 0, MI355_ERR_UNSUPPORTED, "%s: K %d must be a multiple of 128", name, K);
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

extern "C" int32_t mi355_untile_weights(void* dst, const void* src_tiled, int32_t N, int32_t K,
                                        int32_t interleave_halves, void* stream) {
  using namespace mi355;
  const char* name = "mi355_untile_weights";
  MI355_REQUIRE(dst && src_tiled && dst != src_tiled, MI355_ERR_INVALID, "%s: null or aliased pointers", name);
  MI355_REQUIRE(N > 0 && K > 0 && K % 128 == 0, MI355_ERR_UNSUPPORTED, "%s: K %d must be a multiple of 128", name, K);
  MI355_REQUIRE(!interleave_halves || N % 32 == 0, MI355_ERR_UNSUPPORTED,
                "%s: interleaving needs N %% 32 == 0 (two halves of whole 16-row tiles)", name);
  MI355_REQUIRE(aligned16(dst) && aligned16(src_tiled), MI355_ERR_INVALID, "%s: pointers must be 16-byte aligned", name);
  const int tiles = (N + 15) / 16;
  const int64_t blocks = static_cast<int64_t>(tiles) * (K / 32);
  hipLaunchKernelGGL((untile_weights_kernel<0>), dim3(static_cast<unsigned>(blocks > 65535 ? 65535 : blocks)), dim3(64), 0,
                     static_cast<hipStream_t>(stream), static_cast<uint16_t*>(dst),
                     static_cast<const uint16_t*>(src_tiled), N, K, tiles, interleave_halves);
  return check_launch(name);
}

extern "C" int32_t mi355_untile_weights_fp8(void* dst, const void* src_tiled, const float* scale_tiled, int32_t N,
                                            int32_t K, int32_t interleave_halves, int32_t dtype, void* stream) {
  using namespace mi355;
  const char* name = "mi355_untile_weights_fp8";
  MI355_REQUIRE(dst && src_tiled && scale_tiled && dst != src_tiled, MI355_ERR_INVALID, "%s: null or aliased pointers", name);
  MI355_REQUIRE(N > 0 && K > 0 && K % 128 == 0, MI355_ERR_UNSUPPORTED, "%s: K %d must be a multiple of 128", name, K);
  MI355_REQUIRE(!interleave_halves || N % 32 == 0, MI355_ERR_UNSUPPORTED,
                "%s: interleaving needs N %% 32 == 0 (two halves of whole 16-row tiles)", name);
  MI355_REQUIRE(aligned16(dst) && aligned16(src_tiled) && aligned16(scale_tiled), MI355_ERR_INVALID,
                "%s: pointers must be 16-byte aligned", name);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "%s: dtype %d not supported (bf16/fp16 only)", name, dtype);
  const int tiles = (N + 15) / 16;
  const int64_t blocks = static_cast<int64_t>(tiles) * (K / 64);
  const dim3 grid(static_cast<unsigned>(blocks > 65535 ? 65535 : blocks)), block(64);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == MI355_BF16)
    hipLaunchKernelGGL((untile_weights_fp8_kernel<BF16>), grid, block, 0, s, static_cast<uint16_t*>(dst),
                       static_cast<const unsigned char*>(src_tiled), scale_tiled, N, K, tiles, interleave_halves);
  else
    hipLaunchKernelGGL((untile_weights_fp8_kernel<F16>), grid, block, 0, s, static_cast<uint16_t*>(dst),
                       static_cast<const unsigned char*>(src_tiled), scale_tiled, N, K, tiles, interleave_halves);
  return check_launch(name);
}

extern "C" int32_t mi355_linear_tiled(void* y, int64_t y_stride, void* partial, const void* x, int64_t x_stride,
                                      const void* w_tiled, const void* bias, int32_t M, int32_t N, int32_t K,
                                      int32_t tiles_per_wg, int32_t k_groups, int32_t epilogue, int32_t dtype,
                                      void* stream) {
  using namespace mi355;
  const char* name = "mi355_linear_tiled";
  MI355_REQUIRE(epilogue == EPI_STORE || epilogue == EPI_SILU_MUL || epilogue == EPI_PARTIAL, MI355_ERR_INVALID,
                "%s: unknown epilogue %d", name, epilogue);
  return linear_tiled_launch(name, y, y_stride, static_cast<float*>(partial), x, x_stride, w_tiled, bias, M, N, K,
                             tiles_per_wg, k_groups, epilogue, FusedArgs{}, dtype, stream);
}

extern "C" int32_t mi355_linear_tiled_fused(void* y, int64_t y_stride, float* partial, const void* x,
                                            int64_t x_stride, const void* w_tiled, const void* bias, int32_t M,
                                            int32_t N, int32_t K, int32_t tiles_per_wg, int32_t k_groups,
                                            int32_t epilogue, const void* norm_weight, const float* norm_sumsq,
                                            int32_t norm_parts, float eps, void* residual, int64_t residual_stride,
                                            float* sumsq_out, uint32_t* counters, int32_t dtype, void* stream) {
  using namespace mi355;
  const FusedArgs fa{static_cast<const uint16_t*>(norm_weight), norm_sumsq, norm_parts, eps,
                     static_cast<uint16_t*>(residual), residual_stride, sumsq_out, counters};
  return linear_tiled_launch("mi355_linear_tiled_fused", y, y_stride, partial, x, x_stride, w_tiled, bias, M, N, K,
                             tiles_per_wg, k_groups, epilogue, fa, dtype, stream);
}

extern "C" int64_t mi355_tiled_weight_fp8_bytes(int32_t N, int32_t K) {
  if (N <= 0 || K <= 0 || K % 128) return 0;
  return static_cast<int64_t>((N + 15) / 16) * 16 * K;
}

extern "C" int32_t mi355_tile_weights_fp8(void* dst, float* scale_tiled, const void* src, int32_t N, int32_t K,
                                          int32_t interleave_halves, int32_t dtype, void* stream) {
  using namespace mi355;
  const char* name = "mi355_tile_weights_fp8";
  MI355_REQUIRE(dst && src && scale_tiled && dst != src, MI355_ERR_INVALID, "%s: null or aliased pointers", name);
  MI355_REQUIRE(N > 0 && K > 0 && K % 128 == 0, MI355_ERR_UNSUPPORTED, "%s: K %d must be a multiple of 128", name, K);
  MI355_REQUIRE(!interleave_halves || N % 32 == 0, MI355_ERR_UNSUPPORTED,
                "%s: interleaving needs N %% 32 == 0 (two halves of whole 16-row tiles)", name);
  MI355_REQUIRE(aligned16(dst) && aligned16(src) && aligned16(scale_tiled), MI355_ERR_INVALID,
                "%s: pointers must be 16-byte aligned", name);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "%s: dtype %d not supported (bf16/fp16 only)", name, dtype);
  const int tiles = (N + 15) / 16;
  const int64_t blocks = static_cast<int64_t>(tiles) * (K / 64);
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto* sp = static_cast<const uint16_t*>(src);
  auto* dp = static_cast<unsigned char*>(dst);
  const dim3 g1(tiles * 16), g2(static_cast<unsigned>(blocks > 65535 ? 65535 : blocks));
  if (dtype == MI355_BF16) {
    hipLaunchKernelGGL((fp8_row_scale_kernel<BF16>), g1, dim3(256), 0, s, scale_tiled, sp, N, K, interleave_halves);
    hipLaunchKernelGGL((tile_weights_fp8_kernel<BF16>), g2, dim3(64), 0, s, dp, scale_tiled, sp, N, K, tiles, interleave_halves);
  } else {
    hipLaunchKernelGGL((fp8_row_scale_kernel<F16>), g1, dim3(256), 0, s, scale_tiled, sp, N, K, interleave_halves);
    hipLaunchKernelGGL((tile_weights_fp8_kernel<F16>), g2, dim3(64), 0, s, dp, scale_tiled, sp, N, K, tiles, interleave_halves);
  }
  return check_launch(name);
}

extern "C" int32_t mi355_linear_tiled_w8(void* y, int64_t y_stride, void* partial, const void* x, int64_t x_stride,
                                         const void* w_tiled_fp8, const float* w_scale, const void* bias, int32_t M,
                                         int32_t N, int32_t K, int32_t tiles_per_wg, int32_t k_groups,
                                         int32_t epilogue, int32_t dtype, void* stream) {
  using namespace mi355;
  const char* name = "mi355_linear_tiled_w8";
  MI355_REQUIRE(w_scale != nullptr, MI355_ERR_INVALID, "%s: w_scale missing", name);
  MI355_REQUIRE(epilogue == EPI_STORE || epilogue == EPI_SILU_MUL || epilogue == EPI_PARTIAL, MI355_ERR_INVALID,
                "%s: unknown epilogue %d", name, epilogue);
  FusedArgs fa{};
  fa.w_scale = w_scale;
  return linear_tiled_launch(name, y, y_stride, static_cast<float*>(partial), x, x_stride, w_tiled_fp8, bias, M, N, K,
                             tiles_per_wg, k_groups, epilogue, fa, dtype, stream);
}
