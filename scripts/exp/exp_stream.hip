// Experiment: what limits the weight stream?  Variants of the streaming-GEMM inner loop.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma(uint4 a, uint4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// V: 0 frag+nt+x, 1 frag+nt no x, 2 frag plain + x, 4 full-line rows no x, 5 contiguous 1KB per instr no x
template <int V, int U, int XMODE = 0, int LDSPAD = 0>
__global__ __launch_bounds__(256) void k(float* out, const uint16_t* x, const uint16_t* w, int N, int K, int XS) {
  __shared__ float pad[LDSPAD > 0 ? LDSPAD : 1];
  if (LDSPAD > 0 && threadIdx.x == 1000) pad[N % LDSPAD] = 1.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, h = lane >> 4;
  const int tile = blockIdx.x * 4 + wave;
  if (tile * 16 >= N) return;
  const uint16_t* wp;
  int step_stride;  // elements per k-step advance
  if (V == 4) { wp = w + (int64_t)(tile * 16 + (lane >> 3)) * K + 8 * (lane & 7); step_stride = 64; }      // 8 rows x 128B, rows 0-7 then +8 rows
  else if (V == 5) { wp = w + (int64_t)tile * 16 * K + 8 * lane; step_stride = 512; }                       // 1 KB contiguous
  else { wp = w + (int64_t)(tile * 16 + c) * K + 8 * h; step_stride = 32; }
  const uint16_t* xp0 = x + (int64_t)c * XS + 8 * h;
  const uint16_t* xp1 = x + (int64_t)(16 + c) * XS + 8 * h;
  const int steps = (V == 5) ? (16 * K / 512) : (V == 4 ? K / 64 * 2 : K / 32);
  auto load_w = [&](int s) -> uint4 {
    const uint16_t* p;
    if (V == 4) p = wp + (int64_t)(s & 1) * 8 * K + (s >> 1) * 64; else p = wp + (int64_t)s * step_stride;
    if (V == 2) return *reinterpret_cast<const uint4*>(p);
    u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  };
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  uint4 xc = make_uint4(lane, lane * 3, 7, 9);
  uint4 xacc = make_uint4(0, 0, 0, 0);
  uint4 wa[U], xa[U][2], wb[U], xb[U][2];
  const int groups = steps / U;
  auto load_group = [&](uint4 (&wd)[U], uint4 (&xd)[U][2], int g) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int s = g * U + u;
      wd[u] = load_w(s);
      if (V == 0 || V == 2) { const int sx = (XMODE == 1) ? (s & 1) : s; xd[u][0] = *(const uint4*)(xp0 + 32 * sx); xd[u][1] = *(const uint4*)(xp1 + 32 * sx); }
    }
  };
  auto compute = [&](uint4 (&wd)[U], uint4 (&xd)[U][2]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if ((V == 0 || V == 2) && XMODE == 2) { a0 = mfma(wd[u], xc, a0); a1 = mfma(wd[u], xc, a1); xacc.x |= xd[u][0].x ^ xd[u][1].y; }
      else if (V == 0 || V == 2) { a0 = mfma(wd[u], xd[u][0], a0); a1 = mfma(wd[u], xd[u][1], a1); }
      else { a0 = mfma(wd[u], xc, a0); a1 = mfma(wd[u], xc, a1); }
    }
  };
  const int kg = (V == 5 || V == 4) ? groups : groups;   // x index stays in range for V0/V2 because steps == K/32 there
  load_group(wa, xa, 0);
  for (int g = 0; g + 1 < kg; g += 2) {
    load_group(wb, xb, g + 1);
    compute(wa, xa);
    load_group(wa, xa, min(g + 2, kg - 1));
    compute(wb, xb);
  }
  if (kg & 1) compute(wa, xa);
  float r = a0[0] + a0[1] + a0[2] + a0[3] + a1[0] + a1[1] + a1[2] + a1[3];
  if (r == 123.456f || xacc.x == 0x12345u) out[tile * 64 + lane] = r;
}
// x staged in LDS per 512-wide K chunk (single buffer, two barriers per chunk), W direct, U=4
template <int KC>
__global__ __launch_bounds__(256) void klds(float* out, const uint16_t* x, const uint16_t* w, int N, int K, int XS) {
  constexpr int ROWB = KC * 2 + 32;              // padded LDS row (bytes)
  __shared__ __attribute__((aligned(16))) unsigned char xl[32 * ROWB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, h = lane >> 4;
  const int tile = blockIdx.x * 4 + wave;
  const bool act = tile * 16 < N;
  const uint16_t* wp = w + (int64_t)(min(tile * 16 + c, N - 1)) * K + 8 * h;
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  constexpr int U = 4;
  auto load_w = [&](int s) -> uint4 { u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + 32 * s)); return make_uint4(v.x, v.y, v.z, v.w); };
  const int chunks = K / KC;
  constexpr int VPT = 32 * KC / 8 / 256;         // 16B vectors per thread per chunk
  uint4 stage[VPT];
  auto gload = [&](int ch) {
#pragma unroll
    for (int i = 0; i < VPT; ++i) { int v = threadIdx.x + i * 256; int row = v / (KC / 8), col = v % (KC / 8);
      stage[i] = *reinterpret_cast<const uint4*>(x + (int64_t)row * XS + ch * KC + col * 8); } };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < VPT; ++i) { int v = threadIdx.x + i * 256; int row = v / (KC / 8), col = v % (KC / 8);
      *reinterpret_cast<uint4*>(xl + row * ROWB + col * 16) = stage[i]; } };
  gload(0);
  uint4 wa[U];
  for (int u = 0; u < U; ++u) wa[u] = load_w(u);
  int s = 0;
  for (int ch = 0; ch < chunks; ++ch) {
    __syncthreads();
    lstore();
    __syncthreads();
    if (ch + 1 < chunks) gload(ch + 1);
    for (int g = 0; g < KC / 32 / U; ++g, s += U) {
      uint4 wn[U];
      const bool more = s + U < K / 32;
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u) wn[u] = load_w(s + U + u);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int kk = (g * U + u) * 64;   // byte offset of this k-step inside the chunk row
        uint4 b0 = *reinterpret_cast<const uint4*>(xl + c * ROWB + kk + 16 * h);
        uint4 b1 = *reinterpret_cast<const uint4*>(xl + (16 + c) * ROWB + kk + 16 * h);
        a0 = mfma(wa[u], b0, a0); a1 = mfma(wa[u], b1, a1);
      }
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u) wa[u] = wn[u];
      }
    }
  }
  float r = a0[0] + a0[1] + a0[2] + a0[3] + a1[0] + a1[1] + a1[2] + a1[3];
  if (act && r == 123.456f) out[tile * 64 + lane] = r;
}
// v9: x double-buffered in LDS (KC-wide chunks, one barrier per chunk), W fragments direct with ping-pong groups
template <int KC, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k9(float* out, const uint16_t* x, const uint16_t* w, int N, int K, int XS) {
  constexpr int ROWB = KC * 2 + 32;
  constexpr int NT = WAVES * 64;
  constexpr int U = KC / 64;                       // two groups of U k-steps per chunk
  __shared__ __attribute__((aligned(16))) unsigned char xl[2][32 * ROWB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, h = lane >> 4;
  const int tile = blockIdx.x * WAVES + wave;
  const bool act = tile * 16 < N;
  const uint16_t* wp = w + (int64_t)(min(tile * 16 + c, N - 1)) * K + 8 * h;
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  auto load_w = [&](int s) -> uint4 { u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + 32 * s)); return make_uint4(v.x, v.y, v.z, v.w); };
  const int chunks = K / KC;
  constexpr int VPT = 32 * KC / 8 / NT;
  uint4 stage[VPT];
  auto gload = [&](int ch) {
#pragma unroll
    for (int i = 0; i < VPT; ++i) { int v = threadIdx.x + i * NT; int row = v / (KC / 8), col = v % (KC / 8);
      stage[i] = (XS < 0) ? *reinterpret_cast<const uint4*>(x + col * 8) : *reinterpret_cast<const uint4*>(x + (int64_t)row * XS + ch * KC + col * 8); } };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < VPT; ++i) { int v = threadIdx.x + i * NT; int row = v / (KC / 8), col = v % (KC / 8);
      *reinterpret_cast<uint4*>(&xl[buf][row * ROWB + col * 16]) = stage[i]; } };
  uint4 wa[U], wb[U];
  auto load_group = [&](uint4 (&wd)[U], int s0) {
#pragma unroll
    for (int u = 0; u < U; ++u) wd[u] = load_w(s0 + u); };
  auto compute = [&](uint4 (&wd)[U], int buf, int koff) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = (koff + u) * 64;
      uint4 b0 = *reinterpret_cast<const uint4*>(&xl[buf][c * ROWB + kk + 16 * h]);
      uint4 b1 = *reinterpret_cast<const uint4*>(&xl[buf][(16 + c) * ROWB + kk + 16 * h]);
      a0 = mfma(wd[u], b0, a0); a1 = mfma(wd[u], b1, a1);
    } };
  gload(0);
  load_group(wa, 0);
  load_group(wb, U);
  lstore(0);
  __syncthreads();
  const int last = K / 32 - U;
  for (int ch = 0; ch < chunks; ++ch) {
    const int nx = min(ch + 1, chunks - 1);
    gload(nx);
    compute(wa, ch & 1, 0);
    load_group(wa, min((ch + 1) * 2 * U, last));
    compute(wb, ch & 1, U);
    load_group(wb, min((ch + 1) * 2 * U + U, last));
    lstore((ch + 1) & 1);
    __syncthreads();
  }
  float r = a0[0] + a0[1] + a0[2] + a0[3] + a1[0] + a1[1] + a1[2] + a1[3];
  if (act && r == 123.456f) out[tile * 64 + lane] = r;
}
// v17: x K-slice resident in LDS (filled once, one barrier), free-running waves, W direct ping-pong
template <int KSL, int U>
__global__ __launch_bounds__(256) void k17(float* out, const uint16_t* x, const uint16_t* w, int N, int K, int XS) {
  constexpr int ROWB = KSL * 2 + 32;
  __shared__ __attribute__((aligned(16))) unsigned char xl[32 * ROWB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, h = lane >> 4;
  for (int v = threadIdx.x; v < 32 * KSL / 8; v += 256) { int row = v / (KSL / 8), col = v % (KSL / 8);
    *reinterpret_cast<uint4*>(&xl[row * ROWB + col * 16]) = *reinterpret_cast<const uint4*>(x + (int64_t)row * XS + col * 8); }
  __syncthreads();
  const int tile = blockIdx.x * 4 + wave;
  if (tile * 16 >= N) return;
  const uint16_t* wp = w + (int64_t)(tile * 16 + c) * K + 8 * h;
  auto load_w = [&](int s) -> uint4 { u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + 32 * s)); return make_uint4(v.x, v.y, v.z, v.w); };
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  uint4 wa[U], wb[U];
  auto load_group = [&](uint4 (&wd)[U], int g) {
#pragma unroll
    for (int u = 0; u < U; ++u) wd[u] = load_w(g * U + u); };
  auto compute = [&](uint4 (&wd)[U], int g) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = (((g * U + u) * 64) & (KSL * 2 - 1));
      uint4 b0 = *reinterpret_cast<const uint4*>(&xl[c * ROWB + kk + 16 * h]);
      uint4 b1 = *reinterpret_cast<const uint4*>(&xl[(16 + c) * ROWB + kk + 16 * h]);
      a0 = mfma(wd[u], b0, a0); a1 = mfma(wd[u], b1, a1);
    } };
  const int groups = K / 32 / U;
  load_group(wa, 0);
  for (int g = 0; g + 1 < groups; g += 2) {
    load_group(wb, g + 1);
    compute(wa, g);
    load_group(wa, min(g + 2, groups - 1));
    compute(wb, g + 1);
  }
  float r = a0[0] + a0[1] + a0[2] + a0[3] + a1[0] + a1[1] + a1[2] + a1[3];
  if (r == 123.456f) out[tile * 64 + lane] = r;
}
extern "C" int exp_stream(int v, int u, float* out, const void* x, const void* w, int N, int K, int XS, void* stream) {
  dim3 grid((N / 16 + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  auto X = (const uint16_t*)x; auto W = (const uint16_t*)w;
#define L(VV, UU) if (v == VV && u == UU) { hipLaunchKernelGGL((k<VV, UU>), grid, block, 0, s, out, X, W, N, K, XS); return 0; }
  if (v == 17) { hipLaunchKernelGGL((k17<1024, 4>), grid, block, 0, s, out, X, W, N, K, XS); return 0; }
  if (v == 18) { hipLaunchKernelGGL((k17<512, 4>), grid, block, 0, s, out, X, W, N, K, XS); return 0; }
  if (v == 19) { hipLaunchKernelGGL((k17<512, 8>), grid, block, 0, s, out, X, W, N, K, XS); return 0; }
  if (v == 13) { hipLaunchKernelGGL((k<0, 4, 1, 0>), grid, block, 0, s, out, X, W, N, K, XS); return 0; }
  if (v == 14) { hipLaunchKernelGGL((k<1, 4, 0, 10000>), grid, block, 0, s, out, X, W, N, K, XS); return 0; }
  if (v == 15) { hipLaunchKernelGGL((k<0, 4, 2, 0>), grid, block, 0, s, out, X, W, N, K, XS); return 0; }
  if (v == 16) { hipLaunchKernelGGL((k<1, 4, 0, 20000>), grid, block, 0, s, out, X, W, N, K, XS); return 0; }
  if (v == 9) { dim3 g9((N / 16 + 3) / 4); hipLaunchKernelGGL((k9<256, 4>), g9, dim3(256), 0, s, out, X, W, N, K, XS); return 0; }
  if (v == 10) { dim3 g9((N / 16 + 3) / 4); hipLaunchKernelGGL((k9<512, 4>), g9, dim3(256), 0, s, out, X, W, N, K, XS); return 0; }
  if (v == 11) { dim3 g9((N / 16 + 7) / 8); hipLaunchKernelGGL((k9<256, 8>), g9, dim3(512), 0, s, out, X, W, N, K, XS); return 0; }
  if (v == 12) { dim3 g9((N / 16 + 7) / 8); hipLaunchKernelGGL((k9<512, 8>), g9, dim3(512), 0, s, out, X, W, N, K, XS); return 0; }
  if (v == 6) { hipLaunchKernelGGL((klds<512>), grid, block, 0, s, out, X, W, N, K, XS); return 0; }
  if (v == 7) { hipLaunchKernelGGL((klds<256>), grid, block, 0, s, out, X, W, N, K, XS); return 0; }
  if (v == 8) { hipLaunchKernelGGL((klds<1024>), grid, block, 0, s, out, X, W, N, K, XS); return 0; }
  L(0, 4) L(1, 4) L(2, 4) L(4, 4) L(5, 4) L(0, 8) L(1, 8) L(2, 8) L(4, 8) L(5, 8) L(1, 16) L(5, 16) L(4, 16)
  return -1;
}
