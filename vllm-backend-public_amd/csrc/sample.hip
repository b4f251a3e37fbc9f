// Greedy sampling: per-row argmax over the vocabulary.
//
// One 1024-thread workgroup per row streams the row with 16-byte loads and
// keeps, per thread, a single 64-bit key
//     (order-preserving image of the logit) << 32  |  (0xFFFFFFFF - index)
// whose maximum is the largest logit and, among equal logits, the LOWEST index.
// NaN maps to the top key (torch.argmax treats NaN as the maximum) and -0.0 is
// folded onto +0.0.  Keys are max-reduced with wave64 shuffles and one LDS
// exchange.  HBM-bound single pass: vocab * sizeof(logit) bytes per row.
#include "common.h"

namespace mi355 {

constexpr int kArgmaxThreads = 1024;

__device__ __forceinline__ unsigned long long argmax_key(float v, uint32_t idx) {
  uint32_t u = __float_as_uint(v);
  uint32_t k;
  if (v != v) {
    k = 0xFFFFFFFFu;
  } else {
    if (v == 0.0f) u = 0u;
    k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  }
  return (static_cast<unsigned long long>(k) << 32) | static_cast<unsigned long long>(0xFFFFFFFFu - idx);
}

__device__ __forceinline__ unsigned long long umax64(unsigned long long a, unsigned long long b) {
  return a > b ? a : b;
}

// KEYED: out[row] is the row's best key itself, with the index moved by index_offset (the
// position of this slice of the vocabulary in the whole one) and the top bit flipped so that
// SIGNED 64-bit max over the slices' keys orders them as the unsigned keys do.
template <int LDT, bool KEYED>
__global__ __launch_bounds__(kArgmaxThreads) void argmax_kernel(int64_t* __restrict__ out,
                                                                const void* __restrict__ logits,
                                                                int64_t stride, int vocab, uint32_t index_offset) {
  const int row = blockIdx.x;
  unsigned long long best = 0ull;
  if (LDT == MI355_F32) {
    const float* p = static_cast<const float*>(logits) + row * stride;
    const int nvec = vocab >> 2;
    for (int v = threadIdx.x; v < nvec; v += kArgmaxThreads) {
      const float4 f = MI355_LD(float4, logits, p + 4 * v);
      best = umax64(best, argmax_key(f.x, 4 * v));
      best = umax64(best, argmax_key(f.y, 4 * v + 1));
      best = umax64(best, argmax_key(f.z, 4 * v + 2));
      best = umax64(best, argmax_key(f.w, 4 * v + 3));
    }
    for (int i = (nvec << 2) + threadIdx.x; i < vocab; i += kArgmaxThreads)
      best = umax64(best, argmax_key(MI355_LD(float, logits, p + i), i));
  } else {
    const uint16_t* p = static_cast<const uint16_t*>(logits) + row * stride;
    const int nvec = vocab >> 3;
    for (int v = threadIdx.x; v < nvec; v += kArgmaxThreads) {
      float f[8];
      const uint4 raw = MI355_LD(uint4, logits, p + 8 * v);
      if (LDT == MI355_BF16) unpack8<BF16>(raw, f); else unpack8<F16>(raw, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) best = umax64(best, argmax_key(f[j], 8 * v + j));
    }
    for (int i = (nvec << 3) + threadIdx.x; i < vocab; i += kArgmaxThreads) {
      const float f = (LDT == MI355_BF16) ? BF16::to_f(MI355_LD(uint16_t, logits, p + i)) : F16::to_f(MI355_LD(uint16_t, logits, p + i));
      best = umax64(best, argmax_key(f, i));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) best = umax64(best, __shfl_xor(best, off, 64));
  __shared__ unsigned long long part[kArgmaxThreads / 64];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long b = part[0];
#pragma unroll
    for (int i = 1; i < kArgmaxThreads / 64; ++i) b = umax64(b, part[i]);
    const uint32_t idx = 0xFFFFFFFFu - static_cast<uint32_t>(b & 0xFFFFFFFFull);
    if (KEYED) {
      const unsigned long long moved = (b & 0xFFFFFFFF00000000ull) |
                                       static_cast<unsigned long long>(0xFFFFFFFFu - (idx + index_offset));
      MI355_ST(int64_t, out, out + row, static_cast<int64_t>(moved ^ 0x8000000000000000ull));
    } else {
      MI355_ST(int64_t, out, out + row, static_cast<int64_t>(idx));
    }
  }
}

// Chunked form for few rows: one 1024-thread workgroup per row draws its 8 MB alone (14 us for
// 32 x 128256); here a row is cut into `chunks` column ranges, every (row, chunk) workgroup
// max-reduces its range to one key and merges it into keys[row] with a 64-bit atomic max (order
// does not matter for a max: same result on every launch), and a second, tiny kernel turns the
// keys into token ids and clears them for the next launch.
template <int LDT>
__global__ __launch_bounds__(256) void argmax_chunk_kernel(unsigned long long* __restrict__ keys,
                                                            const void* __restrict__ logits, int64_t stride,
                                                            int vocab, int chunk_vecs) {
  constexpr int PER = LDT == MI355_F32 ? 4 : 8;     // elements per 16-byte vector
  const int row = blockIdx.x;
  const int nvec = vocab / PER;
  const int v0 = blockIdx.y * chunk_vecs, v1 = min(nvec, v0 + chunk_vecs);
  unsigned long long best = 0ull;
  if (LDT == MI355_F32) {
    const float* p = static_cast<const float*>(logits) + row * stride;
    for (int v = v0 + threadIdx.x; v < v1; v += 256) {
      const float4 f = MI355_LD(float4, logits, p + 4 * v);
      best = umax64(best, argmax_key(f.x, 4 * v));
      best = umax64(best, argmax_key(f.y, 4 * v + 1));
      best = umax64(best, argmax_key(f.z, 4 * v + 2));
      best = umax64(best, argmax_key(f.w, 4 * v + 3));
    }
    if (blockIdx.y == gridDim.y - 1)
      for (int i = nvec * PER + threadIdx.x; i < vocab; i += 256) best = umax64(best, argmax_key(MI355_LD(float, logits, p + i), i));
  } else {
    const uint16_t* p = static_cast<const uint16_t*>(logits) + row * stride;
    for (int v = v0 + threadIdx.x; v < v1; v += 256) {
      float f[8];
      const uint4 raw = MI355_LD(uint4, logits, p + 8 * v);
      if (LDT == MI355_BF16) unpack8<BF16>(raw, f); else unpack8<F16>(raw, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) best = umax64(best, argmax_key(f[j], 8 * v + j));
    }
    if (blockIdx.y == gridDim.y - 1)
      for (int i = nvec * PER + threadIdx.x; i < vocab; i += 256) {
        const float f = (LDT == MI355_BF16) ? BF16::to_f(MI355_LD(uint16_t, logits, p + i)) : F16::to_f(MI355_LD(uint16_t, logits, p + i));
        best = umax64(best, argmax_key(f, i));
      }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) best = umax64(best, __shfl_xor(best, off, 64));
  __shared__ unsigned long long part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    best = umax64(umax64(part[0], part[1]), umax64(part[2], part[3]));
    if (best != 0ull && MI355_OK_AT(keys, keys + row, 8)) atomicMax(&keys[row], best);   // 0 = an empty column range (every real key is > 0)
  }
}

__global__ void argmax_finish_kernel(int64_t* __restrict__ out, int64_t* __restrict__ out2,
                                     unsigned long long* __restrict__ keys, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long b = MI355_LD(unsigned long long, keys, keys + i);
  MI355_ST(unsigned long long, keys, keys + i, 0ull);                                   // re-armed for the next launch
  const int64_t idx = static_cast<int64_t>(0xFFFFFFFFu - static_cast<uint32_t>(b & 0xFFFFFFFFull));
  MI355_ST(int64_t, out, out + i, idx);
  if (out2 != nullptr) MI355_ST(int64_t, out2, out2 + i, idx);
}

__global__ void argmax_key_to_index_kernel(int64_t* __restrict__ out, const int64_t* __restrict__ keys, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    MI355_ST(int64_t, out, out + i, static_cast<int64_t>(0xFFFFFFFFu - static_cast<uint32_t>(static_cast<unsigned long long>(MI355_LD(int64_t, keys, keys + i)) & 0xFFFFFFFFull)));
}


// ---------------------------------------------------------------------------
// Random sampling: temperature, top-k (any k), top-p (over the whole vocabulary), one
// 1024-thread workgroup per row.  Every arithmetic step is plain fp32 in a fixed order
// (contraction off, polynomial exp and ln) or exact 64-bit integer arithmetic (the
// probability masses), so the result is a pure function of (logits, parameters, seed,
// step) and equals the CPU restatement bit for bit.  Three kinds of row:
//   temperature <= 0      the greedy token (one pass, the argmax key)
//   no cut (top_k <= 0 or >= vocab, and top_p >= 1)
//                         Gumbel-max over the whole row (one pass)
//   a cut                 the kept set as the published samplers define it (see sample_mass
//                         below), then Gumbel-max over the kept set.  Two ways to the set:
//     list  1. one pass keeps every thread's two largest logits (registers); two 8-bit radix
//              passes over those 2048 keys (LDS histogram) give the upper 16 bits of their
//              R-th largest (R = top_k, or 1024 without one): a threshold that at least R
//              logits reach and, for logits without a pathological layout, only a few more do
//           2. a second pass appends every logit >= it to an LDS list (<= 4096 entries) and,
//              for a row without top-k, adds up the mass of the whole row; a bitonic sort
//              orders the list (logit desc, index asc); running sums of the masses; the
//              prefix that reaches the top-p threshold is the kept set
//     exact    when top_k exceeds 1024, the list overflows (huge ties, adversarial layouts)
//              or the nucleus reaches past the list (flat distributions): radix selection
//              straight from the row, by count for top-k and by mass for top-p (sample_exact)
//   Both ways find the same set (integer masses: no rounding), so which one ran cannot be
//   told from the result.
// The streaming passes read the row (vocab * 2 or 4 bytes, L2-resident after the
// first) with 16-byte loads and touch LDS only for the few survivors.
// ---------------------------------------------------------------------------
constexpr int kSampleThreads = 1024;
constexpr int kSampleGuessRank = 1024;   // the LDS list is built around the row's 1024 largest logits (or its top_k, if smaller)
constexpr int kSampleCap = 4096;         // candidates at or above the threshold guess that LDS holds (4 a thread)

// NaN sorts on top, as in argmax_key: a row that holds a NaN yields the index of its first NaN
// from greedy and from random sampling alike
__device__ __forceinline__ uint32_t sample_key(float v) {
  if (v != v) return 0xFFFFFFFFu;
  uint32_t u = (v == 0.0f) ? 0u : __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float key_to_logit(uint32_t k) {
  if (k == 0u) return -INFINITY;
  const uint32_t u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
  return __uint_as_float(u);
}

template <int LDT>
__device__ __forceinline__ float load_logit(const void* row, int i) {
  if (LDT == MI355_F32) return MI355_LD(float, row, static_cast<const float*>(row) + i);
  const uint16_t raw = MI355_LD(uint16_t, row, static_cast<const uint16_t*>(row) + i);
  return LDT == MI355_BF16 ? BF16::to_f(raw) : F16::to_f(raw);
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}

__device__ __forceinline__ uint32_t sample_hash(uint64_t seed, uint64_t step) {
  uint32_t h = mix32(static_cast<uint32_t>(seed) ^ 0x9E3779B9u);
  h = mix32(h ^ static_cast<uint32_t>(seed >> 32));
  h = mix32(h ^ static_cast<uint32_t>(step));
  return mix32(h ^ static_cast<uint32_t>(step >> 32) ^ 0x85EBCA6Bu);
}

__device__ __forceinline__ float sample_uniform(uint64_t seed, uint64_t step) {
  return (static_cast<float>(sample_hash(seed, step) >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

// One uniform per (seed, step, token index), strictly inside (0, 1): 23 hash bits, centred, so
// that both 2^-24 and 1 - 2^-24 are exact fp32 values and log(u), log(-log(u)) stay finite.
__device__ __forceinline__ float sample_uniform_at(uint32_t row_hash, uint32_t index) {
  const uint32_t h = mix32(row_hash ^ mix32(index + 0x632BE5ABu));
  return (static_cast<float>(h >> 9) + 0.5f) * (1.0f / 8388608.0f);
}

// ln(x) for a positive normal x: x = 2^e * m with m in [sqrt(1/2), sqrt(2)), ln(m) = 2 atanh(s),
// s = (m - 1) / (m + 1), |s| < 0.172, odd series to s^9 (next term < 1e-9).  Every operation is one
// IEEE fp32 operation in the order written.
__device__ __forceinline__ float ln_det(float x) {
#pragma clang fp contract(off)
  const uint32_t bits = __float_as_uint(x);
  int e = static_cast<int>(bits >> 23) - 127;
  float m = __uint_as_float((bits & 0x007FFFFFu) | 0x3F800000u);
  if (m > 1.41421354f) {
    m = m * 0.5f;
    e += 1;
  }
  const float s = (m - 1.0f) / (m + 1.0f);
  const float s2 = s * s;
  float p = 0.11111111f;
  p = p * s2;
  p = p + 0.14285715f;
  p = p * s2;
  p = p + 0.2f;
  p = p * s2;
  p = p + 0.33333334f;
  p = p * s2;
  p = p + 1.0f;
  float r = s * p;
  r = r + r;
  const float base = static_cast<float>(e) * 0.69314718f;
  return base + r;
}

// exp(z), z <= 0: 2^n * poly(f) with y = z*log2(e), n = floor(y), f = y - n.
__device__ __forceinline__ float exp_det(float z) {
#pragma clang fp contract(off)
  const float y = z * 1.4426950408889634f;
  const float n = floorf(y);
  if (!(n >= -126.0f)) return 0.0f;
  const float f = y - n;
  const float c[8] = {1.0f, 0.6931471805599453f, 0.2402265069591007f, 0.05550410866482158f,
                      0.009618129107628477f, 0.0013333558146428443f, 0.00015403530393381608f,
                      1.5252733804059840e-05f};
  float r = c[7];
#pragma unroll
  for (int i = 6; i >= 0; --i) {
    r = r * f;
    r = r + c[i];
  }
  const int ni = static_cast<int>(n);
  const float scale = __uint_as_float(static_cast<uint32_t>(ni + 127) << 23);
  return r * scale;
}

// fn(logit, index) for every logit of the row, 16 bytes per lane per load.
template <int LDT, int UNROLL, typename F>
__device__ __forceinline__ void for_each_logit_in(const void* base, int vocab, int v0, int v1, bool tail, F&& fn);

template <int LDT, int UNROLL = 2, typename F>
__device__ __forceinline__ void for_each_logit(const void* base, int vocab, F&& fn) {
  for_each_logit_in<LDT, UNROLL>(base, vocab, 0, vocab, true, fn);
}

// The same over the 16-byte vectors [v0, v1) of the row, plus, if `tail`, the elements after the
// last whole vector.
template <int LDT, int UNROLL, typename F>
__device__ __forceinline__ void for_each_logit_in(const void* base, int vocab, int v0, int v1, bool tail, F&& fn) {
  constexpr int PER = LDT == MI355_F32 ? 4 : 8;
  const int nvec = min(v1, vocab / PER);
#pragma unroll UNROLL
  for (int v = v0 + threadIdx.x; v < nvec; v += kSampleThreads) {
    float f[8];
    if (LDT == MI355_F32) {
      const float4 q = MI355_LD(float4, base, static_cast<const float*>(base) + 4 * v);
      f[0] = q.x, f[1] = q.y, f[2] = q.z, f[3] = q.w;
    } else {
      const uint4 raw = MI355_LD(uint4, base, static_cast<const uint16_t*>(base) + 8 * v);
      if (LDT == MI355_BF16) unpack8<BF16>(raw, f); else unpack8<F16>(raw, f);
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) fn(f[j], v * PER + j);
  }
  if (tail)
    for (int i = vocab / PER * PER + threadIdx.x; i < vocab; i += kSampleThreads) fn(load_logit<LDT>(base, i), i);
}

// fn(f, n, i0) for every 16-byte vector of the row (n = 4 or 8 logits f[0, n) starting at index
// i0) and, one by one (n = 1), for the elements after the last whole vector.
template <int LDT, int UNROLL, typename F>
__device__ __forceinline__ void for_each_vector_in(const void* base, int vocab, int v0, int v1, bool tail, F&& fn);

template <int LDT, int UNROLL, typename F>
__device__ __forceinline__ void for_each_vector(const void* base, int vocab, F&& fn) {
  for_each_vector_in<LDT, UNROLL>(base, vocab, 0, vocab, true, fn);
}

// The same over the 16-byte vectors [v0, v1) of the row, plus, if `tail`, the elements after the last whole vector.
template <int LDT, int UNROLL, typename F>
__device__ __forceinline__ void for_each_vector_in(const void* base, int vocab, int v0, int v1, bool tail, F&& fn) {
  constexpr int PER = LDT == MI355_F32 ? 4 : 8;
  const int nvec = min(v1, vocab / PER);
#pragma unroll UNROLL
  for (int v = v0 + threadIdx.x; v < nvec; v += kSampleThreads) {
    float f[8];
    if (LDT == MI355_F32) {
      const float4 q = MI355_LD(float4, base, static_cast<const float*>(base) + 4 * v);
      f[0] = q.x, f[1] = q.y, f[2] = q.z, f[3] = q.w;
      f[4] = f[5] = f[6] = f[7] = 0.f;
    } else {
      const uint4 raw = MI355_LD(uint4, base, static_cast<const uint16_t*>(base) + 8 * v);
      if (LDT == MI355_BF16) unpack8<BF16>(raw, f); else unpack8<F16>(raw, f);
    }
    fn(f, PER, v * PER);
  }
  if (tail)
    for (int i = vocab / PER * PER + threadIdx.x; i < vocab; i += kSampleThreads) {
      float f[8] = {load_logit<LDT>(base, i), 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      fn(f, 1, i);
    }
}

// Workgroup-wide max of the 64-bit keys; thread 0 gets the result.
__device__ __forceinline__ unsigned long long sample_reduce_max(unsigned long long best, unsigned long long* part) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) best = umax64(best, __shfl_xor(best, off, 64));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0)
#pragma unroll
    for (int i = 1; i < kSampleThreads / 64; ++i) best = umax64(best, part[i]);
  return best;
}

// In-place descending bitonic sort of a[0, P), P a power of two, by the whole workgroup.  The
// caller has made a[] visible (a barrier) before the call; a[] is visible to all after it.
// The array is cut into chunks of C >= 128 elements, one per wave: a compare-exchange step whose
// partner distance j is below C stays inside the chunks and needs no workgroup barrier (a wave's
// LDS operations execute in order), so of the 66 steps of a 2048-element sort only 10 end in one.
template <typename T>
__device__ __forceinline__ void bitonic_desc(T* a, int P) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = P >= 128 * (kSampleThreads / 64) ? P / (kSampleThreads / 64) : min(P, 128);
  const bool mine = wave * C < P;
  T* chunk = a + wave * C;
  bool local_dirty = false;
  for (int size = 2; size <= P; size <<= 1) {
    for (int j = size >> 1; j > 0; j >>= 1) {
      if (j < C) {
        if (mine) {
          for (int q = lane; q < (C >> 1); q += 64) {
            const int il = ((q & ~(j - 1)) << 1) | (q & (j - 1));
            const T x = chunk[il], y = chunk[il | j];
            const bool desc = ((wave * C + il) & size) == 0;
            if (desc ? (x < y) : (x > y)) {
              chunk[il] = y;
              chunk[il | j] = x;
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        local_dirty = true;
      } else {
        if (local_dirty) __syncthreads();
        local_dirty = false;
        for (int idx = threadIdx.x; idx < (P >> 1); idx += kSampleThreads) {
          const int i = ((idx & ~(j - 1)) << 1) | (idx & (j - 1));
          const int p = i | j;
          const T x = a[i], y = a[p];
          const bool desc = (i & size) == 0;
          if (desc ? (x < y) : (x > y)) {
            a[i] = y;
            a[p] = x;
          }
        }
        __syncthreads();
      }
    }
  }
  __syncthreads();
}

// By wave 0: the largest digit d of a 256-bin histogram with at least `need` entries in bins
// >= d, and how many of them bin d itself has to supply.  Lane l owns bins 4l .. 4l+3.
__device__ __forceinline__ void radix_pick(const int* hist, int need, int* digit, int* rem) {
  const int lane = threadIdx.x & 63;
  int h4[4], s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h4[j] = hist[4 * lane + j];
    s += h4[j];
  }
  int suf = s;                                   // entries in the bins of lanes >= this one
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_down(suf, off, 64);
    if (lane + off < 64) suf += t;
  }
  int run = suf - s, best = -1, left = 0;
#pragma unroll
  for (int j = 3; j >= 0; --j) {
    if (best < 0 && run + h4[j] >= need) {
      best = 4 * lane + j;
      left = need - run;
    }
    run += h4[j];
  }
  unsigned long long packed = best >= 0 ? (static_cast<unsigned long long>(best + 1) << 32) | static_cast<uint32_t>(left) : 0ull;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) packed = umax64(packed, __shfl_xor(packed, off, 64));
  if (lane == 0) {
    *digit = packed != 0ull ? static_cast<int>(packed >> 32) - 1 : 0;
    *rem = packed != 0ull ? static_cast<int>(packed & 0xFFFFFFFFull) : need;
  }
}

// Whole-vocabulary draw for a row with neither top-k nor top-p: the Gumbel-max form of sampling
// from softmax(logits / t),
//     token = argmax_i ( logit_i / t  -  ln(-ln(u_i)) ),   u_i = sample_uniform_at(hash(seed, step), i),
// one pass over the row, no candidate set.  Scores are compared through the same order-preserving
// keys as everywhere else (a NaN logit wins, ties go to the lowest index).
template <int LDT>
__device__ __forceinline__ unsigned long long sample_full_vocab(const void* __restrict__ base, int vocab, int v0, int v1,
                                                                bool tail, float temp, uint64_t seed, uint64_t step) {
#pragma clang fp contract(off)
  const uint32_t row_hash = sample_hash(seed, step);
  unsigned long long best = 0ull;
  for_each_logit_in<LDT, 2>(base, vocab, v0, v1, tail, [&](float l, int i) {
    const float u = sample_uniform_at(row_hash, static_cast<uint32_t>(i));
    const float a = -ln_det(u);
    const float g = ln_det(a);
    const float scaled = l / temp;
    const float score = scaled - g;
    best = umax64(best, argmax_key(l != l ? l : score, static_cast<uint32_t>(i)));
  });
  return best;
}

// ---- the kept set of a row with a cut, as the published samplers define it (top-k, then top-p
// over what top-k kept, both on logits / temperature):
//   top-k (0 < k < vocab)  keeps every token whose logit reaches the k-th largest logit (ties at
//                          that value all stay)
//   top-p (p < 1)          orders the survivors by (logit descending, index ascending) and keeps
//                          the shortest prefix whose probability mass reaches p of theirs (the
//                          token that crosses p included; never fewer than one)
// Probability masses are exact integers: F_i = floor(exp_det((l_i - l_max) / t) * 2^40), sums in 64
// bits -- no rounding, so no dependence on the order of additions, and the two ways below of
// finding the prefix agree to the token.  The threshold is P = max(1, ceil(double(p) * double(Z))),
// Z the mass of the top-k survivors; the prefix ends at the first token whose running sum
// reaches P.  The token is then drawn from the kept set in the Gumbel-max form (see
// sample_full_vocab): argmax over the kept i of l_i / t - ln(-ln(u_i)), one uniform per (seed, step,
// token index) -- the same draw whichever way the set was found.
__device__ __forceinline__ unsigned long long sample_mass(float l, float l0, float temp) {
#pragma clang fp contract(off)
  float z = (l - l0) / temp;
  if (l == l0) z = 0.0f;                     // also the row maximum itself when it is +inf
  if (z != z) z = -INFINITY;
  const float w = exp_det(z);
  return static_cast<unsigned long long>(w * 1099511627776.0f);   // 2^40: exact scaling, then truncation
}

__device__ __forceinline__ unsigned long long mass_threshold(float top_p, unsigned long long total) {
  const double pz = static_cast<double>(top_p) * static_cast<double>(total);
  const unsigned long long P = static_cast<unsigned long long>(ceil(pz));
  return P < 1ull ? 1ull : P;
}

__device__ __forceinline__ unsigned long long gumbel_key(float l, uint32_t index, float temp, uint32_t row_hash) {
#pragma clang fp contract(off)
  const float u = sample_uniform_at(row_hash, index);
  const float a = -ln_det(u);
  const float g = ln_det(a);
  const float scaled = l / temp;
  const float score = scaled - g;
  return argmax_key(score, index);
}

// By wave 0: the largest digit d of a 256-bin histogram of 64-bit weights (counts or masses) whose
// bins >= d hold at least `need`, and how much of `need` bin d itself has to supply.  Lane l owns
// bins 4l .. 4l+3.  total_out (may be null): the sum of all bins.
__device__ __forceinline__ void radix_pick64(const unsigned long long* hist, unsigned long long need, int* digit,
                                             unsigned long long* rem, unsigned long long* bin_mass) {
  const int lane = threadIdx.x & 63;
  unsigned long long h4[4], s = 0ull;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h4[j] = hist[4 * lane + j];
    s += h4[j];
  }
  unsigned long long suf = s;                   // weight in the bins of lanes >= this one
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long t = __shfl_down(suf, off, 64);
    if (lane + off < 64) suf += t;
  }
  unsigned long long run = suf - s;
  int best = -1;
  unsigned long long left = 0ull, mass = 0ull;
#pragma unroll
  for (int j = 3; j >= 0; --j) {
    if (best < 0 && run + h4[j] >= need) {
      best = 4 * lane + j;
      left = need - run;
      mass = h4[j];
    }
    run += h4[j];
  }
  // the highest lane with a hit wins
  const unsigned long long any = __ballot(best >= 0);
  const int win = any != 0ull ? 63 - __builtin_clzll(any) : -1;
  if (lane == (win < 0 ? 0 : win)) {
    *digit = win < 0 ? 0 : best;
    *rem = win < 0 ? need : left;
    *bin_mass = win < 0 ? 0ull : mass;
  }
}

__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// for_each_vector that stops early: `go()` is asked before every vector of a thread (the tail elements are
// visited only if it still says yes).
template <int LDT, typename G, typename F>
__device__ __forceinline__ void for_each_vector_while(const void* base, int vocab, G&& go, F&& fn) {
  constexpr int PER = LDT == MI355_F32 ? 4 : 8;
  const int nvec = vocab / PER;
  for (int v = threadIdx.x; v < nvec && go(); v += kSampleThreads) {
    float f[8];
    if (LDT == MI355_F32) {
      const float4 q = MI355_LD(float4, base, static_cast<const float*>(base) + 4 * v);
      f[0] = q.x, f[1] = q.y, f[2] = q.z, f[3] = q.w;
      f[4] = f[5] = f[6] = f[7] = 0.f;
    } else {
      const uint4 raw = MI355_LD(uint4, base, static_cast<const uint16_t*>(base) + 8 * v);
      if (LDT == MI355_BF16) unpack8<BF16>(raw, f); else unpack8<F16>(raw, f);
    }
    fn(f, PER, v * PER);
  }
  if (go())
    for (int i = nvec * PER + threadIdx.x; i < vocab; i += kSampleThreads) {
      float f[8] = {load_logit<LDT>(base, i), 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      fn(f, 1, i);
    }
}

// One 16-byte vector of the row as fp32 (4 logits of an fp32 row, 8 of a 16-bit one).
template <int LDT>
__device__ __forceinline__ void load_vector(const void* base, int v, float (&f)[8]) {
  if (LDT == MI355_F32) {
    const float4 q = MI355_LD(float4, base, static_cast<const float*>(base) + 4 * v);
    f[0] = q.x, f[1] = q.y, f[2] = q.z, f[3] = q.w;
    f[4] = f[5] = f[6] = f[7] = 0.f;
  } else {
    const uint4 raw = MI355_LD(uint4, base, static_cast<const uint16_t*>(base) + 8 * v);
    if (LDT == MI355_BF16) unpack8<BF16>(raw, f); else unpack8<F16>(raw, f);
  }
}

// fn(logit, index) for every kept token among the vectors [v_begin, v_end) of the row (and, if `tail`, the
// elements after the row's last whole vector): kept = key > Tp, or key == Tp and fewer than `take` tokens with
// that key precede it in index order (take == ~0: all of them stay).  ties_before: such tokens before v_begin.
// Waves own contiguous runs of vectors and a lane one vector of 64 consecutive ones at a time, so a token's rank
// among the ties is the count in the waves before + a scan over the lanes + its place inside the vector.
// eq_base: kSampleThreads / 64 + 1 ints of LDS.  All threads of the workgroup call it.
template <int LDT, typename F>
__device__ __forceinline__ void ordered_kept_pass(const void* base, int vocab, int v_begin, int v_end, bool tail,
                                                  uint32_t Tp, unsigned long long take,
                                                  unsigned long long ties_before, int* eq_base, F&& fn) {
  constexpr int PER = LDT == MI355_F32 ? 4 : 8;
  constexpr int kWaves = kSampleThreads / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  v_end = min(v_end, vocab / PER);
  v_begin = min(v_begin, v_end);
  const int VW = (((v_end - v_begin) + kWaves - 1) / kWaves + 63) / 64 * 64;
  const int w0 = min(v_end, v_begin + wave * VW), w1 = min(v_end, w0 + VW);
  const bool ranked = take != ~0ull;                 // the same in every thread
  unsigned long long eq_run = ties_before;
  if (ranked) {
    int c = 0;
    for (int vb = w0; vb < w1; vb += 64) {
      const int v = vb + lane;
      if (v < w1) {
        float f[8];
        load_vector<LDT>(base, v, f);
#pragma unroll
        for (int j = 0; j < PER; ++j) c += sample_key(f[j]) == Tp ? 1 : 0;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if (lane == 0) eq_base[wave + 1] = c;
    __syncthreads();
    if (tid == 0) {
      eq_base[0] = 0;
      for (int w = 0; w < kWaves; ++w) eq_base[w + 1] += eq_base[w];
    }
    __syncthreads();
    eq_run += static_cast<unsigned long long>(eq_base[wave]);
  }
  for (int vb = w0; vb < w1; vb += 64) {
    const int v = vb + lane;
    const bool valid = v < w1;
    float f[8];
    if (valid) load_vector<LDT>(base, v, f);
    int excl = 0, total = 0;
    if (ranked) {
      int c = 0;
      if (valid) {
#pragma unroll
        for (int j = 0; j < PER; ++j) c += sample_key(f[j]) == Tp ? 1 : 0;
      }
      int incl = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
      }
      excl = incl - c;
      total = __shfl(incl, 63, 64);
    }
    if (valid) {
      unsigned long long rank = eq_run + static_cast<unsigned long long>(excl);
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const uint32_t key = sample_key(f[j]);
        bool kept = key > Tp;
        if (key == Tp) {
          kept = !ranked || rank < take;
          ++rank;
        }
        if (kept) fn(f[j], v * PER + j);
      }
    }
    eq_run += static_cast<unsigned long long>(total);
  }
  if (tail && tid == 0) {                              // fewer than PER elements, after every vector of the row
    unsigned long long rank = ties_before + (ranked ? static_cast<unsigned long long>(eq_base[kWaves]) : 0ull);
    for (int i = vocab / PER * PER; i < vocab; ++i) {
      const float f = load_logit<LDT>(base, i);
      const uint32_t key = sample_key(f);
      bool kept = key > Tp;
      if (key == Tp) {
        kept = !ranked || rank < take;
        ++rank;
      }
      if (kept) fn(f, i);
    }
  }
}

// The exact kept set straight from the row, for rows whose candidates do not fit the LDS list
// (top_k beyond the list, a nucleus that reaches past it, huge ties): radix selection over the
// 32-bit order-preserving keys, 8 bits a pass -- by COUNT for the k-th largest key Tk, then by
// MASS among the keys >= Tk for the key Tp at which the running mass reaches the threshold -- and
// the number m of tokens with key == Tp that the prefix still takes (lowest indices first).  Then
// one ordered pass draws by Gumbel-max from {key > Tp} and the first m of {key == Tp}.
// Returns the winning 64-bit key in thread 0.
template <int LDT>
__device__ __forceinline__ unsigned long long sample_exact(const void* base, int vocab, int K1, float tp, float temp,
                                                           float l0, uint32_t row_hash, unsigned long long* hist,
                                                           unsigned long long* part, int* eq_base, int* shi,
                                                           unsigned long long* shl) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // A 16-bit logit widened to fp32 has zero low bits (bf16: 16, fp16: 13), so the low bits of its key
  // follow from the sign -- zeros for a key with the top bit set, ones otherwise: the last passes of
  // a selection over such a row have nothing to decide and are skipped (bf16: two of four, fp16: one).
  constexpr int kLastShift = LDT == MI355_BF16 ? 16 : LDT == MI355_F16 ? 8 : 0;
  auto low_bits = [](uint32_t prefix) -> uint32_t {
    return (prefix & 0x80000000u) ? 0u : ((1u << kLastShift) - 1u);
  };
  // One histogram add per run of equal digits inside a 16-byte vector (a row's neighbours mostly share
  // their leading key bits): the LDS serialises adds to one bin, and at the first level that is most of a pass.
  auto add_runs = [&](const uint32_t (&dig)[8], const unsigned long long (&val)[8], int nf) {
    unsigned long long acc = 0ull;
    uint32_t cur = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < nf && dig[j] != 0xFFFFFFFFu) {
        if (dig[j] != cur) {
          if (cur != 0xFFFFFFFFu && acc != 0ull) atomicAdd(&hist[cur], acc);
          cur = dig[j];
          acc = 0ull;
        }
        acc += val[j];
      }
    }
    if (cur != 0xFFFFFFFFu && acc != 0ull) atomicAdd(&hist[cur], acc);
  };
  uint32_t Tk = 0u;
  if (K1 > 0) {
    uint32_t prefix = 0u, mask = 0u;
    unsigned long long need = static_cast<unsigned long long>(K1);
    for (int shift = 24; shift >= kLastShift; shift -= 8) {
      if (tid < 256) hist[tid] = 0ull;
      __syncthreads();
      for_each_vector<LDT, 2>(base, vocab, [&](const float (&f)[8], int nf, int) {
        uint32_t dig[8];
        unsigned long long val[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t key = sample_key(f[j]);
          dig[j] = (key & mask) == prefix ? (key >> shift) & 255u : 0xFFFFFFFFu;
          val[j] = 1ull;
        }
        add_runs(dig, val, nf);
      });
      __syncthreads();
      if (wave == 0) radix_pick64(hist, need, &shi[0], &shl[0], &shl[1]);
      __syncthreads();
      prefix |= static_cast<uint32_t>(shi[0]) << shift;
      mask |= 255u << shift;
      need = shl[0];
      __syncthreads();
    }
    Tk = prefix | (kLastShift ? low_bits(prefix) : 0u);
  }
  uint32_t Tp = Tk;
  unsigned long long take = ~0ull;             // tokens with key == Tp that stay (all of them unless top-p cuts inside the tie)
  if (tp < 1.0f) {
    uint32_t prefix = 0u, mask = 0u;
    unsigned long long need = 0ull;
    for (int shift = 24; shift >= kLastShift; shift -= 8) {
      if (tid < 256) hist[tid] = 0ull;
      __syncthreads();
      for_each_vector<LDT, 2>(base, vocab, [&](const float (&f)[8], int nf, int) {
        uint32_t dig[8];
        unsigned long long val[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t key = sample_key(f[j]);
          const bool in = key >= Tk && (key & mask) == prefix;
          dig[j] = in ? (key >> shift) & 255u : 0xFFFFFFFFu;
          val[j] = in && j < nf ? sample_mass(f[j], l0, temp) : 0ull;
        }
        add_runs(dig, val, nf);
      });
      __syncthreads();
      if (shift == 24) {                         // the first level's histogram holds the whole mass Z of the top-k survivors
        if (wave == 0) {
          unsigned long long t = hist[4 * lane] + hist[4 * lane + 1] + hist[4 * lane + 2] + hist[4 * lane + 3];
          t = wave_sum64(t);
          if (lane == 0) shl[2] = mass_threshold(tp, t);
        }
        __syncthreads();
        need = shl[2];
      }
      if (wave == 0) radix_pick64(hist, need, &shi[0], &shl[0], &shl[1]);
      __syncthreads();
      prefix |= static_cast<uint32_t>(shi[0]) << shift;
      mask |= 255u << shift;
      need = shl[0];
      __syncthreads();
    }
    Tp = prefix | (kLastShift ? low_bits(prefix) : 0u);
    const unsigned long long Fp = sample_mass(key_to_logit(Tp), l0, temp);
    take = Fp > 0ull ? (need + Fp - 1ull) / Fp : ~0ull;     // need >= 1; every token with key == Tp weighs Fp
  }
  unsigned long long best = 0ull;
  ordered_kept_pass<LDT>(base, vocab, 0, vocab, true, Tp, take, 0ull, eq_base, [&](float f, int i) {
    best = umax64(best, gumbel_key(f, static_cast<uint32_t>(i), temp, row_hash));
  });
  __syncthreads();
  return sample_reduce_max(best, part);
}

// ---- the exact path spread over many workgroups.  One workgroup computes exp / ln for a whole row
// at the pace of ONE CU (a pass over 128256 logits with a mass per logit: ~35 us; a flat row needs
// five of them); the launch has up to 256.  A row of the cut launch that needs the exact path
// (and has a slot in the scratch below: the first kExactRows rows) therefore only raises its flag
// there, and a fixed sequence of follow-up launches does the radix selection with the row cut into
// column ranges, one workgroup each:
//   exact_hist_kernel(stage)   stage s < NL: counts per digit of key level s (top-k);
//                              s >= NL: masses per digit of level s - NL (top-p).  Every
//                              workgroup first replays the picks of the earlier stages from their
//                              finished histograms (a few hundred instructions, the same result
//                              in every workgroup), then adds its range's histogram to the row's
//                              with 64-bit integer atomics -- exact, so order-free
//   exact_final_kernel         Gumbel-max over {key > Tp} and the first `take` tokens with
//                              key == Tp (rank = such tokens in the ranges before, counted from
//                              the row, + ordered rank inside the range), merged into out[row]
//                              by a 64-bit atomic max
// NL = key levels that can differ for the logits' type (sample_exact: 2 / 3 / 4).  The launches
// always run (a fixed sequence replays from a HIP graph) and return at once for rows whose flag is
// down.  The scratch is one set per process and GPU: two sampling calls in flight on one GPU at
// the same time (different streams) would share it -- the engine samples on one stream.
constexpr int kExactRows = 256;
constexpr int kExactStages = 8;
static __device__ unsigned long long g_exact_hist[kExactRows][kExactStages][256];
static __device__ unsigned int g_exact_flag[kExactRows];
static __device__ float g_exact_l0[kExactRows];

template <int LDT>
struct ExactLevels {
  static constexpr int kLastShift = LDT == MI355_BF16 ? 16 : LDT == MI355_F16 ? 8 : 0;
  static constexpr int NL = (24 - kLastShift) / 8 + 1;
  __device__ static uint32_t low_bits(uint32_t prefix) { return (prefix & 0x80000000u) ? 0u : ((1u << kLastShift) - 1u); }
};

struct ExactState {
  uint32_t Tk, Tp;               // keys >= Tk survive top-k; Tp: the key at which the top-p prefix ends
  unsigned long long take;       // tokens with key == Tp that stay (~0: all)
  uint32_t prefix, mask;         // of the selection in progress (stage replay)
  unsigned long long need;
};

// radix_pick64 on bins already in registers (lane l holds bins 4l .. 4l+3), result in every lane.
__device__ __forceinline__ void radix_pick_regs(const unsigned long long (&h4)[4], unsigned long long need, int& digit,
                                                unsigned long long& rem) {
  const int lane = threadIdx.x & 63;
  const unsigned long long s = h4[0] + h4[1] + h4[2] + h4[3];
  unsigned long long suf = s;                   // weight in the bins of lanes >= this one
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long t = __shfl_down(suf, off, 64);
    if (lane + off < 64) suf += t;
  }
  unsigned long long run = suf - s, left = 0ull;
  int best = -1;
#pragma unroll
  for (int j = 3; j >= 0; --j) {
    if (best < 0 && run + h4[j] >= need) {
      best = 4 * lane + j;
      left = need - run;
    }
    run += h4[j];
  }
  const unsigned long long any = __ballot(best >= 0);
  const int win = any != 0ull ? 63 - __builtin_clzll(any) : 0;      // the highest lane with a hit wins
  digit = any != 0ull ? __shfl(best, win, 64) : 0;
  rem = any != 0ull ? __shfl(left, win, 64) : need;
}

// Replays the picks of stages [0, upto) for `row` from the finished histograms.  Wave 0 does it alone: it asks
// for every histogram it will need at once (one memory round trip instead of one per level, each followed by two
// workgroup barriers -- a third of the follow-up launches' time), picks level after level with shuffles only and
// hands the state to the workgroup through LDS (shi / shl scratch).  upto = 2 * NL: the full state.
template <int LDT>
__device__ __forceinline__ ExactState exact_replay(int row, int upto, int K1, float tp, float temp, float l0, int* shi,
                                                   unsigned long long* shl) {
  using L = ExactLevels<LDT>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  ExactState st;
  st.Tk = 0u, st.Tp = 0u, st.take = ~0ull, st.prefix = 0u, st.mask = 0u, st.need = 0ull;
  if (wave == 0) {
    unsigned long long h[2 * L::NL][4];
#pragma unroll
    for (int s_ = 0; s_ < 2 * L::NL; ++s_) {
      const bool on = s_ < upto && (s_ < L::NL ? K1 > 0 : tp < 1.0f);
#pragma unroll
      for (int j = 0; j < 4; ++j) h[s_][j] = on ? g_exact_hist[row][s_][4 * lane + j] : 0ull;
    }
    // top-k: stages 0 .. NL-1
    if (K1 > 0) {
      st.need = static_cast<unsigned long long>(K1);
#pragma unroll
      for (int lvl = 0; lvl < L::NL; ++lvl) {
        if (lvl < upto) {
          const int shift = 24 - 8 * lvl;
          int digit;
          radix_pick_regs(h[lvl], st.need, digit, st.need);
          st.prefix |= static_cast<uint32_t>(digit) << shift;
          st.mask |= 255u << shift;
        }
      }
      if (upto >= L::NL) st.Tk = st.prefix | (L::kLastShift ? L::low_bits(st.prefix) : 0u);
    }
    st.Tp = st.Tk;
    if (upto >= L::NL) {
      // top-p: stages NL .. 2 NL - 1, a fresh selection among the keys >= Tk
      st.prefix = 0u, st.mask = 0u, st.need = 0ull;
      if (tp < 1.0f) {
#pragma unroll
        for (int lvl = 0; lvl < L::NL; ++lvl) {
          if (L::NL + lvl < upto) {
            const int shift = 24 - 8 * lvl;
            if (lvl == 0)                          // the first level's histogram holds the whole mass Z of the top-k survivors
              st.need = mass_threshold(tp, wave_sum64(h[L::NL][0] + h[L::NL][1] + h[L::NL][2] + h[L::NL][3]));
            int digit;
            radix_pick_regs(h[L::NL + lvl], st.need, digit, st.need);
            st.prefix |= static_cast<uint32_t>(digit) << shift;
            st.mask |= 255u << shift;
          }
        }
        if (upto >= 2 * L::NL) {
          st.Tp = st.prefix | (L::kLastShift ? L::low_bits(st.prefix) : 0u);
          const unsigned long long Fp = sample_mass(key_to_logit(st.Tp), l0, temp);
          st.take = Fp > 0ull ? (st.need + Fp - 1ull) / Fp : ~0ull;
        }
      }
    }
    if (lane == 0) {
      shl[0] = st.take;
      shl[1] = st.need;
      shl[2] = (static_cast<unsigned long long>(st.Tk) << 32) | st.Tp;
      shl[3] = (static_cast<unsigned long long>(st.prefix) << 32) | st.mask;
    }
  }
  __syncthreads();
  st.take = shl[0];
  st.need = shl[1];
  st.Tk = static_cast<uint32_t>(shl[2] >> 32), st.Tp = static_cast<uint32_t>(shl[2] & 0xFFFFFFFFull);
  st.prefix = static_cast<uint32_t>(shl[3] >> 32), st.mask = static_cast<uint32_t>(shl[3] & 0xFFFFFFFFull);
  __syncthreads();
  (void)shi;
  return st;
}

template <int LDT>
__global__ __launch_bounds__(kSampleThreads) void exact_hist_kernel(
    const void* __restrict__ logits, int64_t stride, int vocab, const float* __restrict__ temperature,
    const int32_t* __restrict__ top_k, const float* __restrict__ top_p, int chunk_vecs, int stage) {
  using L = ExactLevels<LDT>;
  const int row = blockIdx.x;
  if (row >= kExactRows || g_exact_flag[row] == 0u) return;
  const int k = MI355_LD(int32_t, top_k, top_k + row);
  const int K1 = (k > 0 && k < vocab) ? k : 0;
  const float tp = MI355_LD(float, top_p, top_p + row), temp = MI355_LD(float, temperature, temperature + row);
  const bool counting = stage < L::NL;
  if (counting ? K1 == 0 : !(tp < 1.0f)) return;       // this selection is off for the row
  const float l0 = g_exact_l0[row];
  __shared__ unsigned long long hist[256];
  __shared__ int shi[2];
  __shared__ unsigned long long shl[4];
  const int tid = threadIdx.x;
  const ExactState st = exact_replay<LDT>(row, stage, K1, tp, temp, l0, shi, shl);
  const int shift = 24 - 8 * (counting ? stage : stage - L::NL);
  if (tid < 256) hist[tid] = 0ull;
  __syncthreads();
  const void* base = LDT == MI355_F32 ? static_cast<const void*>(static_cast<const float*>(logits) + row * stride)
                                      : static_cast<const void*>(static_cast<const uint16_t*>(logits) + row * stride);
  const int v0 = blockIdx.y * chunk_vecs, v1 = v0 + chunk_vecs;
  const bool tail = blockIdx.y == gridDim.y - 1;
  for_each_vector_in<LDT, 2>(base, vocab, v0, v1, tail, [&](const float (&f)[8], int nf, int) {
    unsigned long long acc = 0ull;
    uint32_t cur = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < nf) {
        const uint32_t key = sample_key(f[j]);
        if (key >= st.Tk && (key & st.mask) == st.prefix) {
          const uint32_t d = (key >> shift) & 255u;
          if (d != cur) {
            if (cur != 0xFFFFFFFFu && acc != 0ull) atomicAdd(&hist[cur], acc);
            cur = d;
            acc = 0ull;
          }
          acc += counting ? 1ull : sample_mass(f[j], l0, temp);
        }
      }
    }
    if (cur != 0xFFFFFFFFu && acc != 0ull) atomicAdd(&hist[cur], acc);
  });
  __syncthreads();
  if (tid < 256 && hist[tid] != 0ull) atomicAdd(&g_exact_hist[row][stage][tid], hist[tid]);
}

template <int LDT>
__global__ __launch_bounds__(kSampleThreads) void exact_final_kernel(
    int64_t* __restrict__ out, const void* __restrict__ logits, int64_t stride, int vocab,
    const float* __restrict__ temperature, const int32_t* __restrict__ top_k, const float* __restrict__ top_p,
    const int64_t* __restrict__ seeds, const int64_t* __restrict__ steps, int chunk_vecs) {
  using L = ExactLevels<LDT>;
  const int row = blockIdx.x;
  if (row >= kExactRows || g_exact_flag[row] == 0u) return;
  const int k = MI355_LD(int32_t, top_k, top_k + row);
  const int K1 = (k > 0 && k < vocab) ? k : 0;
  const float tp = MI355_LD(float, top_p, top_p + row), temp = MI355_LD(float, temperature, temperature + row);
  const float l0 = g_exact_l0[row];
  __shared__ int shi[2];
  __shared__ unsigned long long shl[4];
  __shared__ int eq_base[kSampleThreads / 64 + 1];
  __shared__ unsigned long long part[kSampleThreads / 64];
  const int tid = threadIdx.x;
  const ExactState st = exact_replay<LDT>(row, 2 * L::NL, K1, tp, temp, l0, shi, shl);
  const void* base = LDT == MI355_F32 ? static_cast<const void*>(static_cast<const float*>(logits) + row * stride)
                                      : static_cast<const void*>(static_cast<const uint16_t*>(logits) + row * stride);
  // tokens at the cut key in the column ranges before this one (only when top-p cuts inside that tie): counted
  // here, from the row itself -- a compare per logit, cheaper than a launch of its own that would leave the counts
  unsigned long long before = 0ull;
  if (st.take != ~0ull && blockIdx.y != 0) {
    unsigned int mine = 0u;
    for_each_vector_in<LDT, 2>(base, vocab, 0, static_cast<int>(blockIdx.y) * chunk_vecs, false,
                               [&](const float (&f)[8], int nf, int) {
#pragma unroll
                                 for (int j = 0; j < 8; ++j) mine += (j < nf && sample_key(f[j]) == st.Tp) ? 1u : 0u;
                               });
    unsigned long long w = wave_sum64(static_cast<unsigned long long>(mine));
    if ((tid & 63) == 0) part[tid >> 6] = w;
    __syncthreads();
    for (int q = 0; q < kSampleThreads / 64; ++q) before += part[q];
    __syncthreads();
  }
  const uint32_t row_hash = sample_hash(static_cast<uint64_t>(MI355_LD(int64_t, seeds, seeds + row)),
                                        static_cast<uint64_t>(MI355_LD(int64_t, steps, steps + row)));
  unsigned long long best = 0ull;
  ordered_kept_pass<LDT>(base, vocab, static_cast<int>(blockIdx.y) * chunk_vecs, (static_cast<int>(blockIdx.y) + 1) * chunk_vecs,
                         blockIdx.y == gridDim.y - 1, st.Tp, st.take, before, eq_base, [&](float f, int i) {
                           best = umax64(best, gumbel_key(f, static_cast<uint32_t>(i), temp, row_hash));
                         });
  __syncthreads();
  best = sample_reduce_max(best, part);
  if (tid == 0 && best != 0ull && MI355_OK_AT(out, out + row, 8))
    atomicMax(reinterpret_cast<unsigned long long*>(out) + row, best);
}

// ---- the k most likely tokens of a row (the `logprobs` / `top_logprobs` lists of the completion routes): an
// exact selection, in (logit descending, index ascending) order, NaN on top as in the greedy pick.  One workgroup
// per row: radix selection by count over the order-preserving keys for the k-th largest key (the passes a 16-bit
// logit cannot differ in are skipped), then the ordered pass of the samplers' exact search, which keeps every key
// above it and the lowest-indexed of the tokens tied with it, then the k entries rank themselves.
constexpr int kTopkMax = 64;

template <int LDT>
__global__ __launch_bounds__(kSampleThreads) void topk_logits_kernel(
    int64_t* __restrict__ out_ids, float* __restrict__ out_vals, const void* __restrict__ logits, int64_t stride,
    int vocab, int k, const float* __restrict__ lse) {
  using L = ExactLevels<LDT>;
  const int row = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6;
  const void* base = LDT == MI355_F32 ? static_cast<const void*>(static_cast<const float*>(logits) + row * stride)
                                      : static_cast<const void*>(static_cast<const uint16_t*>(logits) + row * stride);
  __shared__ unsigned long long hist[256];
  __shared__ int shi[2];
  __shared__ unsigned long long shl[4];
  __shared__ int eq_base[kSampleThreads / 64 + 1];
  __shared__ unsigned long long found[kTopkMax];
  __shared__ int nfound;
  const int K = min(k, vocab);
  uint32_t prefix = 0u, mask = 0u;
  unsigned long long need = static_cast<unsigned long long>(K);
  for (int shift = 24; shift >= L::kLastShift; shift -= 8) {
    if (tid < 256) hist[tid] = 0ull;
    if (tid == 0) nfound = 0;
    __syncthreads();
    for_each_vector<LDT, 2>(base, vocab, [&](const float (&f)[8], int nf, int) {
      unsigned long long acc = 0ull;             // one LDS add per run of equal digits inside a vector
      uint32_t cur = 0xFFFFFFFFu;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < nf) {
          const uint32_t key = sample_key(f[j]);
          if ((key & mask) == prefix) {
            const uint32_t d = (key >> shift) & 255u;
            if (d != cur) {
              if (cur != 0xFFFFFFFFu) atomicAdd(&hist[cur], acc);
              cur = d;
              acc = 0ull;
            }
            acc += 1ull;
          }
        }
      }
      if (cur != 0xFFFFFFFFu) atomicAdd(&hist[cur], acc);
    });
    __syncthreads();
    if (wave == 0) radix_pick64(hist, need, &shi[0], &shl[0], &shl[1]);
    __syncthreads();
    prefix |= static_cast<uint32_t>(shi[0]) << shift;
    mask |= 255u << shift;
    need = shl[0];
    __syncthreads();
  }
  const uint32_t Tk = prefix | (L::kLastShift ? L::low_bits(prefix) : 0u);
  // every key above Tk and the first `need` tokens (by index) with key == Tk: K tokens in all
  ordered_kept_pass<LDT>(base, vocab, 0, vocab, true, Tk, need, 0ull, eq_base, [&](float f, int i) {
    const int at = atomicAdd(&nfound, 1);
    if (at < kTopkMax)
      found[at] = (static_cast<unsigned long long>(sample_key(f)) << 32) |
                  static_cast<unsigned long long>(0xFFFFFFFFu - static_cast<uint32_t>(i));
  });
  __syncthreads();
  const int n = min(nfound, K);
  const float sub = lse != nullptr ? MI355_LD(float, lse, lse + row) : 0.f;
  if (tid < k) {
    int64_t id = -1;
    float val = -INFINITY;
    int rank = tid;
    if (tid < n) {
      const unsigned long long mine = found[tid];
      rank = 0;
      for (int j = 0; j < n; ++j) rank += found[j] > mine ? 1 : 0;       // entries are distinct (their indices are)
      id = static_cast<int64_t>(0xFFFFFFFFu - static_cast<uint32_t>(mine & 0xFFFFFFFFull));
      val = load_logit<LDT>(base, static_cast<int>(id)) - sub;
    }
    MI355_ST(int64_t, out_ids, out_ids + (static_cast<int64_t>(row) * k + rank), id);
    MI355_ST(float, out_vals, out_vals + (static_cast<int64_t>(row) * k + rank), val);
  }
}

// Rows that need one key only: greedy ones and draws with neither top-k nor top-p.
__device__ __forceinline__ bool sample_one_key(float temp, int top_k, float top_p) {
  return !(temp > 0.0f) || (top_k <= 0 && !(top_p < 1.0f));
}

__global__ void sample_finish_kernel(int64_t* __restrict__ out, const float* __restrict__ temperature,
                                     const int32_t* __restrict__ top_k, const float* __restrict__ top_p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool spread = i < kExactRows && g_exact_flag[i] != 0u;      // its token came from the follow-up launches, as a key
  if (spread) g_exact_flag[i] = 0u;
  if (!spread && !sample_one_key(MI355_LD(float, temperature, temperature + i), MI355_LD(int32_t, top_k, top_k + i),
                                 MI355_LD(float, top_p, top_p + i)))
    return;
  const unsigned long long b = static_cast<unsigned long long>(MI355_LD(int64_t, out, out + i));
  MI355_ST(int64_t, out, out + i, static_cast<int64_t>(0xFFFFFFFFu - static_cast<uint32_t>(b & 0xFFFFFFFFull)));
}

template <int LDT>
__global__ __launch_bounds__(kSampleThreads) void sample_kernel(
    int64_t* __restrict__ out, const void* __restrict__ logits, int64_t stride, int vocab,
    const float* __restrict__ temperature, const int32_t* __restrict__ top_k, const float* __restrict__ top_p,
    const int64_t* __restrict__ seeds, const int64_t* __restrict__ steps, int chunk_vecs, int spread_exact) {
#pragma clang fp contract(off)
  const int row = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // a flag an aborted call may have left up must not reach this call's follow-up launches (the same thread
  // raises it again further down if this row needs them)
  if (blockIdx.y == 0 && tid == 0 && row < kExactRows) g_exact_flag[row] = 0u;
  const void* base = LDT == MI355_F32
                         ? static_cast<const void*>(static_cast<const float*>(logits) + row * stride)
                         : static_cast<const void*>(static_cast<const uint16_t*>(logits) + row * stride);
  __shared__ unsigned long long comp[kSampleCap];        // candidates: (key << 32) | inverted index
  __shared__ unsigned long long pre[kSampleCap];         // running sums of their masses
  __shared__ unsigned long long hist64[256];
  __shared__ unsigned long long part[kSampleThreads / 64];
  __shared__ unsigned long long shl[6];
  __shared__ int hist[256];
  __shared__ int eq_base[kSampleThreads / 64 + 1];
  __shared__ int sh[4];

  const float temp = MI355_LD(float, temperature, temperature + row);
  int k = MI355_LD(int32_t, top_k, top_k + row);
  const float tp = MI355_LD(float, top_p, top_p + row);
  const uint64_t seed = static_cast<uint64_t>(MI355_LD(int64_t, seeds, seeds + row));
  const uint64_t step = static_cast<uint64_t>(MI355_LD(int64_t, steps, steps + row));
  const bool greedy = !(temp > 0.0f);
  if (sample_one_key(temp, k, tp)) {
    // one pass, one key: the row is cut into gridDim.y column ranges, every workgroup merges the
    // best key of its range into out[row] (cleared by the launcher) with a 64-bit atomic max --
    // order-free, so the same on every launch -- and sample_finish_kernel turns it into the index
    const int v0 = blockIdx.y * chunk_vecs, v1 = v0 + chunk_vecs;
    const bool tail = blockIdx.y == gridDim.y - 1;
    unsigned long long best = 0ull;
    if (greedy)
      for_each_logit_in<LDT, 2>(base, vocab, v0, v1, tail,
                                [&](float f, int i) { best = umax64(best, argmax_key(f, static_cast<uint32_t>(i))); });
    else
      best = sample_full_vocab<LDT>(base, vocab, v0, v1, tail, temp, seed, step);
    best = sample_reduce_max(best, part);
    if (tid == 0 && best != 0ull && MI355_OK_AT(out, out + row, 8))
      atomicMax(reinterpret_cast<unsigned long long*>(out) + row, best);
    return;
  }
  if (blockIdx.y != 0) return;
  const int K1 = (k > 0 && k < vocab) ? k : 0;          // an active top-k
  const uint32_t row_hash = sample_hash(seed, step);

  // ---- 1. one pass: every thread's two largest logits (for the threshold guess), the row maximum,
  // and the first NaN (a row that holds a NaN yields its index, like the greedy pick)
  float f1 = -INFINITY, f2 = -INFINITY;
  uint32_t nan_at = 0xFFFFFFFFu;
  for_each_logit<LDT>(base, vocab, [&](float f, int i) {
    if (f != f) nan_at = min(nan_at, static_cast<uint32_t>(i));
    const float g = f != f ? -INFINITY : f;
    f2 = fmaxf(f2, fminf(f1, g));
    f1 = fmaxf(f1, g);
  });
  {
    // (max logit, inverted min NaN index) in one 64-bit max-reduction
    unsigned long long both = (static_cast<unsigned long long>(sample_key(f1)) << 32) |
                              static_cast<unsigned long long>(0xFFFFFFFFu - nan_at);
    unsigned long long hi = both & 0xFFFFFFFF00000000ull, lo = both & 0xFFFFFFFFull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      hi = umax64(hi, __shfl_xor(hi, off, 64));
      lo = umax64(lo, __shfl_xor(lo, off, 64));
    }
    if (lane == 0) part[wave] = hi | lo;
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < kSampleThreads / 64; ++w) {
        hi = umax64(hi, part[w] & 0xFFFFFFFF00000000ull);
        lo = umax64(lo, part[w] & 0xFFFFFFFFull);
      }
      shl[3] = hi | lo;
      sh[3] = 0;                                   // list length
      sh[2] = 0;                                   // ties beyond the k-th entry
      shl[4] = 0ull;                               // mass of the whole row
    }
    __syncthreads();
  }
  const uint32_t first_nan = 0xFFFFFFFFu - static_cast<uint32_t>(shl[3] & 0xFFFFFFFFull);
  if (first_nan != 0xFFFFFFFFu) {
    if (tid == 0) MI355_ST(int64_t, out, out + row, static_cast<int64_t>(first_nan));
    return;
  }
  const float l0 = key_to_logit(static_cast<uint32_t>(shl[3] >> 32));
  unsigned long long best = 0ull;
  bool exact = K1 > kSampleGuessRank;                   // the list is built around the 1024 largest: a deeper top-k goes straight to the exact path
  if (!exact) {
    // the R-th largest of the 2048 per-thread keys, to its upper 16 bits (two 8-bit radix passes in
    // LDS), lower bits clear: at least R logits of the row reach it (the 2048 keys are keys of distinct
    // logits; stand-ins of threads with fewer than two logits are -inf and can only lower the guess)
    const int Rk = min(K1 > 0 ? K1 : kSampleGuessRank, vocab);
    const uint32_t m1 = sample_key(f1), m2 = sample_key(f2);
    uint32_t guess = 0u;
    {
      int need = Rk;
      for (int shift = 24; shift >= 16; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const uint32_t above = shift == 24 ? 0u : 0xFF000000u;
        if ((m1 & above) == (guess & above)) atomicAdd(&hist[(m1 >> shift) & 255u], 1);
        if ((m2 & above) == (guess & above)) atomicAdd(&hist[(m2 >> shift) & 255u], 1);
        __syncthreads();
        if (wave == 0) radix_pick(hist, need, &sh[0], &sh[1]);
        __syncthreads();
        guess |= static_cast<uint32_t>(sh[0]) << shift;
        need = sh[1];
      }
    }
    // ---- 2. second pass: every logit at or above the guess goes to the LDS list (one counter bump per
    // lane and vector that has any); a row without top-k also adds up the mass of the WHOLE row, which
    // is what its top-p threshold is a share of
    const float guess_f = key_to_logit(guess);
    unsigned long long zsum = 0ull;
    // (a flat row overflows the list within its first tenth: the pass stops there, the row goes to the exact search)
    auto list_fits = [&]() { return *reinterpret_cast<volatile int*>(&sh[3]) <= kSampleCap; };
    for_each_vector_while<LDT>(base, vocab, list_fits, [&](const float (&f)[8], int nf, int i0) {
      int hits = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) hits += (j < nf && !(f[j] < guess_f)) ? 1 : 0;
      if (K1 == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < nf) zsum += sample_mass(f[j], l0, temp);
      }
      if (hits != 0) {
        int pos = atomicAdd(&sh[3], hits);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < nf && !(f[j] < guess_f)) {
            if (pos < kSampleCap)
              comp[pos] = (static_cast<unsigned long long>(sample_key(f[j])) << 32) |
                          static_cast<unsigned long long>(0xFFFFFFFFu - static_cast<uint32_t>(i0 + j));
            ++pos;
          }
      }
    });
    if (K1 == 0) {
      zsum = wave_sum64(zsum);
      if (lane == 0) atomicAdd(&shl[4], zsum);
    }
    __syncthreads();
    const int n = sh[3];
    exact = n < Rk || n > kSampleCap;                   // (n < Rk cannot happen; kept as a guard)
    if (!exact) {
      int P2 = 64;
      while (P2 < n) P2 <<= 1;
      for (int i = n + tid; i < P2; i += kSampleThreads) comp[i] = 0ull;
      __syncthreads();
      bitonic_desc(comp, P2);
      // the top-k survivors: the first K1 entries and every later one tied with the K1-th
      int kk = n;
      if (K1 > 0) {
        const uint32_t key_k = static_cast<uint32_t>(comp[K1 - 1] >> 32);
        int extra = 0;
        for (int j = K1 + tid; j < n; j += kSampleThreads) extra += static_cast<uint32_t>(comp[j] >> 32) == key_k ? 1 : 0;
        if (extra != 0) atomicAdd(&sh[2], extra);
        __syncthreads();
        kk = K1 + sh[2];
      }
      // masses and their running sums (exact 64-bit integers): thread t owns entries 4t .. 4t+3
      unsigned long long m4[4], run = 0ull;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = 4 * tid + q;
        m4[q] = j < kk ? sample_mass(key_to_logit(static_cast<uint32_t>(comp[j] >> 32)), l0, temp) : 0ull;
        run += m4[q];
        m4[q] = run;
      }
      unsigned long long incl = run;                     // inclusive scan of the threads' totals inside the wave
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
      }
      if (lane == 63) part[wave] = incl;
      __syncthreads();
      unsigned long long before = incl - run;
      for (int w = 0; w < wave; ++w) before += part[w];
#pragma unroll
      for (int q = 0; q < 4; ++q) pre[4 * tid + q] = before + m4[q];
      if (tid == 0) sh[0] = 0x7FFFFFFF;
      __syncthreads();
      int keep = kk;
      if (tp < 1.0f) {
        const unsigned long long Z = K1 > 0 ? pre[kk - 1] : shl[4];
        const unsigned long long P = mass_threshold(tp, Z);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = 4 * tid + q;
          if (j < kk && pre[j] >= P && (j == 0 || pre[j - 1] < P)) atomicMin(&sh[0], j + 1);
        }
        __syncthreads();
        keep = sh[0];
        exact = keep == 0x7FFFFFFF;                     // the nucleus reaches past the list: the exact path
      }
      if (!exact) {
        for (int j = tid; j < keep; j += kSampleThreads) {
          const uint32_t idx = 0xFFFFFFFFu - static_cast<uint32_t>(comp[j] & 0xFFFFFFFFull);
          best = umax64(best, gumbel_key(key_to_logit(static_cast<uint32_t>(comp[j] >> 32)), idx, temp, row_hash));
        }
        __syncthreads();
        best = sample_reduce_max(best, part);
      }
    }
  }
  if (exact && row < kExactRows && spread_exact) {
    // hand the row to the follow-up launches: clear its histograms, leave the row maximum, raise the flag
    __syncthreads();
    unsigned long long* h = &g_exact_hist[row][0][0];
    for (int j = tid; j < kExactStages * 256; j += kSampleThreads) h[j] = 0ull;
    if (tid == 0) {
      g_exact_l0[row] = l0;
      g_exact_flag[row] = 1u;
    }
    return;                                             // out[row] stays 0 (cleared by the launcher): the final launch max-merges into it
  }
  if (exact) {
    __syncthreads();
    best = sample_exact<LDT>(base, vocab, K1, tp, temp, l0, row_hash, hist64, part, eq_base, sh, shl);
  }
  if (tid == 0)
    MI355_ST(int64_t, out, out + row, static_cast<int64_t>(0xFFFFFFFFu - static_cast<uint32_t>(best & 0xFFFFFFFFull)));
}

}  // namespace mi355

namespace mi355 {
static int32_t check_logits(const char* name, const int64_t* out, const void* logits, int64_t logits_stride,
                            int32_t num_rows, int32_t vocab, int32_t logits_dtype, int64_t index_offset) {
  MI355_REQUIRE(out && logits, MI355_ERR_INVALID, "%s: null pointer", name);
  MI355_REQUIRE(num_rows >= 0 && vocab > 0, MI355_ERR_INVALID, "%s: num_rows/vocab out of range", name);
  MI355_REQUIRE(index_offset >= 0 && index_offset + vocab <= 0xFFFFFFFFll, MI355_ERR_INVALID,
                "%s: index_offset + vocab must fit 32 bits", name);
  MI355_REQUIRE(logits_dtype == MI355_BF16 || logits_dtype == MI355_F16 || logits_dtype == MI355_F32,
                MI355_ERR_UNSUPPORTED, "%s: logits dtype %d not supported", name, logits_dtype);
  const int per16 = logits_dtype == MI355_F32 ? 4 : 8;
  MI355_REQUIRE(logits_stride >= vocab && logits_stride % per16 == 0 && aligned16(logits),
                MI355_ERR_INVALID, "%s: logits rows must be 16-byte aligned (stride %lld)", name,
                (long long)logits_stride);
  return MI355_OK;
}

static int32_t argmax_entry(const char* name, int64_t* out, const void* logits, int64_t logits_stride,
                            int32_t num_rows, int32_t vocab, int32_t logits_dtype, bool keyed, int64_t index_offset,
                            void* stream) {
  const int32_t ok = check_logits(name, out, logits, logits_stride, num_rows, vocab, logits_dtype, index_offset);
  if (ok != MI355_OK || num_rows == 0) return ok;
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid(num_rows), block(kArgmaxThreads);
  const uint32_t off = static_cast<uint32_t>(index_offset);
#define MI355_ARGMAX(LDT)                                                                                         \
  do {                                                                                                            \
    if (keyed)                                                                                                    \
      hipLaunchKernelGGL((argmax_kernel<LDT, true>), grid, block, 0, s, out, logits, logits_stride, vocab, off);  \
    else                                                                                                          \
      hipLaunchKernelGGL((argmax_kernel<LDT, false>), grid, block, 0, s, out, logits, logits_stride, vocab, off); \
  } while (0)
  if (logits_dtype == MI355_F32) MI355_ARGMAX(MI355_F32);
  else if (logits_dtype == MI355_BF16) MI355_ARGMAX(MI355_BF16);
  else MI355_ARGMAX(MI355_F16);
#undef MI355_ARGMAX
  return check_launch(name);
}
}  // namespace mi355

extern "C" int32_t mi355_argmax(int64_t* out_ids, const void* logits, int64_t logits_stride,
                                int32_t num_rows, int32_t vocab, int32_t logits_dtype, void* stream) {
  return mi355::argmax_entry("mi355_argmax", out_ids, logits, logits_stride, num_rows, vocab, logits_dtype, false, 0,
                             stream);
}

extern "C" int32_t mi355_argmax_chunked(int64_t* out_ids, int64_t* out_ids2, int64_t* keys, const void* logits,
                                        int64_t logits_stride, int32_t num_rows, int32_t vocab,
                                        int32_t logits_dtype, void* stream) {
  using namespace mi355;
  const char* name = "mi355_argmax_chunked";
  MI355_REQUIRE(out_ids && keys && logits, MI355_ERR_INVALID, "%s: null pointer", name);
  MI355_REQUIRE(num_rows >= 0 && vocab > 0, MI355_ERR_INVALID, "%s: num_rows/vocab out of range", name);
  MI355_REQUIRE(logits_dtype == MI355_BF16 || logits_dtype == MI355_F16 || logits_dtype == MI355_F32,
                MI355_ERR_UNSUPPORTED, "%s: logits dtype %d not supported", name, logits_dtype);
  const int per16 = logits_dtype == MI355_F32 ? 4 : 8;
  MI355_REQUIRE(logits_stride >= vocab && logits_stride % per16 == 0 && aligned16(logits) &&
                    (reinterpret_cast<uintptr_t>(keys) & 7u) == 0,
                MI355_ERR_INVALID, "%s: logits rows must be 16-byte aligned (stride %lld), keys 8-byte aligned", name,
                (long long)logits_stride);
  if (num_rows == 0) return MI355_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // enough workgroups to draw from every HBM channel (>= ~512), each with at least 2 vectors per thread
  const int nvec = vocab / per16;
  int chunks = (512 + num_rows - 1) / num_rows;
  chunks = chunks < 1 ? 1 : chunks > 64 ? 64 : chunks;
  while (chunks > 1 && (nvec + chunks - 1) / chunks < 512) --chunks;
  const int chunk_vecs = (nvec + chunks - 1) / chunks;
  dim3 grid(num_rows, chunks), block(256);
  auto* kp = reinterpret_cast<unsigned long long*>(keys);
  if (logits_dtype == MI355_F32)
    hipLaunchKernelGGL((argmax_chunk_kernel<MI355_F32>), grid, block, 0, s, kp, logits, logits_stride, vocab, chunk_vecs);
  else if (logits_dtype == MI355_BF16)
    hipLaunchKernelGGL((argmax_chunk_kernel<MI355_BF16>), grid, block, 0, s, kp, logits, logits_stride, vocab, chunk_vecs);
  else
    hipLaunchKernelGGL((argmax_chunk_kernel<MI355_F16>), grid, block, 0, s, kp, logits, logits_stride, vocab, chunk_vecs);
  int32_t rc = check_launch(name);
  if (rc != MI355_OK) return rc;
  hipLaunchKernelGGL(argmax_finish_kernel, dim3((num_rows + 255) / 256), dim3(256), 0, s, out_ids, out_ids2, kp, num_rows);
  return check_launch(name);
}

extern "C" int32_t mi355_argmax_shard_key(int64_t* out_keys, const void* logits, int64_t logits_stride,
                                          int32_t num_rows, int32_t vocab, int32_t logits_dtype,
                                          int64_t index_offset, void* stream) {
  return mi355::argmax_entry("mi355_argmax_shard_key", out_keys, logits, logits_stride, num_rows, vocab, logits_dtype,
                             true, index_offset, stream);
}

extern "C" int32_t mi355_argmax_key_to_index(int64_t* out_ids, const int64_t* keys, int32_t num_rows, void* stream) {
  using namespace mi355;
  const char* name = "mi355_argmax_key_to_index";
  MI355_REQUIRE(out_ids && keys, MI355_ERR_INVALID, "%s: null pointer", name);
  MI355_REQUIRE(num_rows >= 0, MI355_ERR_INVALID, "%s: num_rows out of range", name);
  if (num_rows == 0) return MI355_OK;
  hipLaunchKernelGGL(argmax_key_to_index_kernel, dim3((num_rows + 255) / 256), dim3(256), 0,
                     static_cast<hipStream_t>(stream), out_ids, keys, num_rows);
  return check_launch(name);
}

namespace mi355 {
static int32_t sample_entry(const char* name, int64_t* out_ids, const void* logits, int64_t logits_stride,
                            int32_t num_rows, int32_t vocab, int32_t logits_dtype, const float* temperature,
                            const int32_t* top_k, const float* top_p, const int64_t* seeds, const int64_t* steps,
                            int32_t cut_rows, void* stream) {
  MI355_REQUIRE(temperature && top_k && top_p && seeds && steps, MI355_ERR_INVALID, "%s: null pointer", name);
  const int32_t rc = check_logits(name, out_ids, logits, logits_stride, num_rows, vocab, logits_dtype, 0);
  if (rc != MI355_OK || num_rows == 0) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // one-key rows (greedy, no cut) are spread over `chunks` workgroups each so that few rows still
  // occupy every CU: their pass is ALU-bound (two logarithms and two divisions per logit)
  const int nvec = vocab / (logits_dtype == MI355_F32 ? 4 : 8);
  int chunks = (512 + num_rows - 1) / num_rows;
  chunks = chunks < 1 ? 1 : chunks > 64 ? 64 : chunks;
  while (chunks > 1 && (nvec + chunks - 1) / chunks < kSampleThreads) --chunks;
  const int chunk_vecs = (nvec + chunks - 1) / chunks;
  const hipError_t cleared = hipMemsetAsync(out_ids, 0, sizeof(int64_t) * num_rows, s);
  MI355_REQUIRE(cleared == hipSuccess, MI355_ERR_LAUNCH, "%s: clearing the keys failed: %s", name, hipGetErrorString(cleared));
  dim3 grid(num_rows, chunks), block(kSampleThreads);
  // rows that need the exact kept-set search are spread over the launch's column ranges by follow-up launches
  // (a fixed sequence: they return at once when no row raised its flag); MI355_SAMPLE_SPREAD=0 keeps such rows
  // in their one workgroup
  static const int spread = [] {
    const char* e = getenv("MI355_SAMPLE_SPREAD");
    return (e != nullptr && e[0] == '0') ? 0 : 1;
  }();
  const int spread_exact = spread && cut_rows != 0 && chunks > 1 ? 1 : 0;
#define MI355_SAMPLE_LAUNCH(LDT)                                                                                      \
  do {                                                                                                                \
    hipLaunchKernelGGL((sample_kernel<LDT>), grid, block, 0, s, out_ids, logits, logits_stride, vocab, temperature,    \
                       top_k, top_p, seeds, steps, chunk_vecs, spread_exact);                                          \
    if (spread_exact) {                                                                                               \
      dim3 xgrid(num_rows < kExactRows ? num_rows : kExactRows, chunks);                                              \
      for (int stage = 0; stage < 2 * ExactLevels<LDT>::NL; ++stage)                                                   \
        hipLaunchKernelGGL((exact_hist_kernel<LDT>), xgrid, block, 0, s, logits, logits_stride, vocab, temperature,    \
                           top_k, top_p, chunk_vecs, stage);                                                          \
      hipLaunchKernelGGL((exact_final_kernel<LDT>), xgrid, block, 0, s, out_ids, logits, logits_stride, vocab,         \
                         temperature, top_k, top_p, seeds, steps, chunk_vecs);                                         \
    }                                                                                                                 \
  } while (0)
  if (logits_dtype == MI355_F32) MI355_SAMPLE_LAUNCH(MI355_F32);
  else if (logits_dtype == MI355_BF16) MI355_SAMPLE_LAUNCH(MI355_BF16);
  else MI355_SAMPLE_LAUNCH(MI355_F16);
#undef MI355_SAMPLE_LAUNCH
  const int32_t rc2 = check_launch(name);
  if (rc2 != MI355_OK) return rc2;
  hipLaunchKernelGGL(sample_finish_kernel, dim3((num_rows + 255) / 256), dim3(256), 0, s, out_ids, temperature, top_k,
                     top_p, num_rows);
  return check_launch(name);
}
}  // namespace mi355

extern "C" int32_t mi355_sample_topk_topp(int64_t* out_ids, const void* logits, int64_t logits_stride,
                                          int32_t num_rows, int32_t vocab, int32_t logits_dtype,
                                          const float* temperature, const int32_t* top_k, const float* top_p,
                                          const int64_t* seeds, const int64_t* steps, void* stream) {
  return mi355::sample_entry("mi355_sample_topk_topp", out_ids, logits, logits_stride, num_rows, vocab, logits_dtype,
                             temperature, top_k, top_p, seeds, steps, 1, stream);
}

extern "C" int32_t mi355_sample_topk_topp_hinted(int64_t* out_ids, const void* logits, int64_t logits_stride,
                                                 int32_t num_rows, int32_t vocab, int32_t logits_dtype,
                                                 const float* temperature, const int32_t* top_k, const float* top_p,
                                                 const int64_t* seeds, const int64_t* steps, int32_t cut_rows,
                                                 void* stream) {
  return mi355::sample_entry("mi355_sample_topk_topp_hinted", out_ids, logits, logits_stride, num_rows, vocab,
                             logits_dtype, temperature, top_k, top_p, seeds, steps, cut_rows, stream);
}

extern "C" int32_t mi355_topk_logits(int64_t* out_ids, float* out_vals, const void* logits, int64_t logits_stride,
                                     int32_t num_rows, int32_t vocab, int32_t logits_dtype, int32_t k,
                                     const float* lse, void* stream) {
  using namespace mi355;
  const char* name = "mi355_topk_logits";
  MI355_REQUIRE(out_vals != nullptr, MI355_ERR_INVALID, "%s: null pointer", name);
  MI355_REQUIRE(k >= 1 && k <= kTopkMax, MI355_ERR_UNSUPPORTED, "%s: k %d outside [1, %d]", name, k, kTopkMax);
  const int32_t rc = check_logits(name, out_ids, logits, logits_stride, num_rows, vocab, logits_dtype, 0);
  if (rc != MI355_OK || num_rows == 0) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid(num_rows), block(kSampleThreads);
  if (logits_dtype == MI355_F32)
    hipLaunchKernelGGL((topk_logits_kernel<MI355_F32>), grid, block, 0, s, out_ids, out_vals, logits, logits_stride, vocab, k, lse);
  else if (logits_dtype == MI355_BF16)
    hipLaunchKernelGGL((topk_logits_kernel<MI355_BF16>), grid, block, 0, s, out_ids, out_vals, logits, logits_stride, vocab, k, lse);
  else
    hipLaunchKernelGGL((topk_logits_kernel<MI355_F16>), grid, block, 0, s, out_ids, out_vals, logits, logits_stride, vocab, k, lse);
  return check_launch(name);
}
