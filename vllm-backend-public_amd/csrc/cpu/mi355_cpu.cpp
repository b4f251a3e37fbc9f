// CPU twin of the operator ABI (include/mi355_decode_cpu.h): the same argument checks, layouts
// and rounding points as the HIP kernels, in plain C++ (OpenMP over rows / sequences).  Not a
// fallback of the serving engine -- it exists so that the ABI can be exercised without a GPU and
// under sanitizers, and as the CPU baseline of the benchmark.
#include "mi355_decode_cpu.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "mi355_decode.h"

namespace {

thread_local char g_err[512] = "";

int32_t fail(int32_t code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define CPU_REQUIRE(cond, code, ...)              \
  do {                                            \
    if (!(cond)) return fail((code), __VA_ARGS__); \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline float bits_f(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
inline uint32_t f_bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}

// ---- 16-bit <-> float, round to nearest even, NaN stays NaN (as v_cvt_pk_bf16_f32 / v_cvt_f16_f32)
struct BF16 {
  static float to_f(uint16_t u) { return bits_f(static_cast<uint32_t>(u) << 16); }
  static uint16_t from_f(float f) {
    const uint32_t u = f_bits(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return static_cast<uint16_t>((u >> 16) | 0x40u);   // quiet NaN
    return static_cast<uint16_t>((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
  }
};

struct F16 {
  static float to_f(uint16_t h) {
    const uint32_t sign = static_cast<uint32_t>(h & 0x8000u) << 16;
    const uint32_t ex = (h >> 10) & 0x1Fu, man = h & 0x3FFu;
    if (ex == 0) {                                    // zero / subnormal: man * 2^-24
      const float v = static_cast<float>(man) * 5.9604644775390625e-08f;
      return bits_f(sign | f_bits(v));
    }
    if (ex == 31) return bits_f(sign | 0x7F800000u | (man << 13));
    return bits_f(sign | ((ex + 112u) << 23) | (man << 13));
  }
  static uint16_t from_f(float f) {
    const uint32_t u = f_bits(f);
    const uint16_t sign = static_cast<uint16_t>((u >> 16) & 0x8000u);
    const uint32_t a = u & 0x7FFFFFFFu;
    if (a > 0x7F800000u) return static_cast<uint16_t>(sign | 0x7E00u);           // NaN
    if (a >= 0x47800000u) return static_cast<uint16_t>(sign | 0x7C00u);          // >= 65536 (incl. inf) -> inf ...
    if (a >= 0x477FF000u) return static_cast<uint16_t>(sign | 0x7C00u);          // ... and what rounds up to it
    if (a < 0x33000001u) return sign;                                             // < 2^-25 (or == 2^-25: tie to even 0)
    const int e = static_cast<int>(a >> 23) - 127;
    if (e < -14) {                                                                // subnormal half: units of 2^-24
      const float scaled = bits_f(a) * 16777216.0f;                               // exact
      const float r = nearbyintf(scaled);                                         // default mode: to nearest even
      return static_cast<uint16_t>(sign | static_cast<uint16_t>(r));
    }
    uint32_t m = a & 0x7FFFFFu;
    uint32_t out = (static_cast<uint32_t>(e + 15) << 10) | (m >> 13);
    const uint32_t rem = m & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (out & 1u))) ++out;                   // carries into the exponent correctly
    return static_cast<uint16_t>(sign | out);
  }
};

template <typename T>
inline float rnd(float f) {
  return T::to_f(T::from_f(f));
}

template <typename T>
int32_t rms_norm_impl(uint16_t* out, const uint16_t* x, uint16_t* residual, const uint16_t* w, int64_t T_, int hidden,
                      int64_t xs, int64_t os, int64_t rs, float eps) {
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < T_; ++t) {
    std::vector<float> row(hidden);
    float ss = 0.f;
    for (int j = 0; j < hidden; ++j) {
      float f = T::to_f(x[t * xs + j]);
      if (residual) {
        f = rnd<T>(f + T::to_f(residual[t * rs + j]));
        residual[t * rs + j] = T::from_f(f);
      }
      row[j] = f;
      ss += f * f;
    }
    const float inv = 1.0f / sqrtf(ss / static_cast<float>(hidden) + eps);
    for (int j = 0; j < hidden; ++j) out[t * os + j] = T::from_f(rnd<T>(row[j] * inv) * T::to_f(w[j]));
  }
  return MI355_OK;
}

template <typename T>
int32_t layer_norm_impl(uint16_t* out, const uint16_t* x, uint16_t* residual, const uint16_t* w, const uint16_t* b,
                        int64_t T_, int hidden, int64_t xs, int64_t os, int64_t rs, float eps) {
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < T_; ++t) {
    std::vector<float> row(hidden);
    float s = 0.f;
    for (int j = 0; j < hidden; ++j) {
      float f = T::to_f(x[t * xs + j]);
      if (residual) {
        f = rnd<T>(f + T::to_f(residual[t * rs + j]));
        residual[t * rs + j] = T::from_f(f);
      }
      row[j] = f;
      s += f;
    }
    const float mean = s / static_cast<float>(hidden);
    float ss = 0.f;
    for (int j = 0; j < hidden; ++j) ss += (row[j] - mean) * (row[j] - mean);
    const float inv = 1.0f / sqrtf(ss / static_cast<float>(hidden) + eps);
    for (int j = 0; j < hidden; ++j) {
      const float n = (row[j] - mean) * inv;        // separate statements: no contraction across them with -ffp-contract=off
      const float m = n * T::to_f(w[j]);
      out[t * os + j] = T::from_f(m + T::to_f(b[j]));
    }
  }
  return MI355_OK;
}

int32_t norm_checks(const char* name, const void* out, const void* x, const void* residual, bool add, const void* w,
                    int64_t T_, int32_t hidden, int64_t xs, int64_t os, int64_t rs, int32_t dtype) {
  CPU_REQUIRE(out && x && w && (!add || residual), MI355_ERR_INVALID, "%s: null pointer", name);
  CPU_REQUIRE(T_ >= 0 && T_ <= 0x7FFFFFFF, MI355_ERR_INVALID, "%s: num_tokens %lld out of range", name, (long long)T_);
  CPU_REQUIRE(hidden > 0 && hidden % 8 == 0, MI355_ERR_UNSUPPORTED, "%s: hidden %d must be a positive multiple of 8",
              name, hidden);
  CPU_REQUIRE(xs >= hidden && os >= hidden && xs % 8 == 0 && os % 8 == 0 && (!add || (rs >= hidden && rs % 8 == 0)),
              MI355_ERR_INVALID, "%s: row strides must be >= hidden and multiples of 8", name);
  CPU_REQUIRE(aligned16(out) && aligned16(x) && aligned16(w) && (!add || aligned16(residual)), MI355_ERR_INVALID,
              "%s: pointers must be 16-byte aligned", name);
  CPU_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED, "%s: dtype %d not supported (bf16/fp16 only)",
              name, dtype);
  return MI355_OK;
}

int32_t cache_checks(const char* name, int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t dtype) {
  CPU_REQUIRE(num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0, MI355_ERR_INVALID,
              "%s: num_heads %d must be a positive multiple of num_kv_heads %d", name, num_heads, num_kv_heads);
  CPU_REQUIRE(num_heads / num_kv_heads <= 16, MI355_ERR_UNSUPPORTED, "%s: at most 16 query heads per kv head (got %d)", name,
              num_heads / num_kv_heads);
  CPU_REQUIRE(head_dim == 64 || head_dim == 128, MI355_ERR_UNSUPPORTED, "%s: head_dim %d not supported (64 or 128)", name,
              head_dim);
  CPU_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED, "%s: dtype %d not supported (bf16/fp16 only)",
              name, dtype);
  return MI355_OK;
}

// softmax(q . K^T * scale) V for one (sequence, head) over keys [0, n) of the paged cache
template <typename T>
void attend(uint16_t* out, const uint16_t* q, const uint16_t* kc, const uint16_t* vc, int num_blocks, const int32_t* bt,
            int n, int kvh, int num_kv_heads, int d, float scale) {
  constexpr int BS = MI355_BLOCK_SIZE;
  std::vector<float> s(n), qf(d), acc(d, 0.f);
  for (int j = 0; j < d; ++j) qf[j] = T::to_f(q[j]);
  float mx = -INFINITY;
  for (int t = 0; t < n; ++t) {
    int blk = bt[t / BS];
    blk = blk < 0 ? 0 : blk >= num_blocks ? num_blocks - 1 : blk;
    const uint16_t* kr = kc + ((static_cast<int64_t>(blk) * num_kv_heads + kvh) * BS + t % BS) * d;
    float dot = 0.f;
    for (int j = 0; j < d; ++j) dot += qf[j] * T::to_f(kr[j]);
    s[t] = dot * scale;
    mx = fmaxf(mx, s[t]);
  }
  float l = 0.f;
  for (int t = 0; t < n; ++t) {
    s[t] = expf(s[t] - mx);
    l += s[t];
  }
  for (int t = 0; t < n; ++t) {
    int blk = bt[t / BS];
    blk = blk < 0 ? 0 : blk >= num_blocks ? num_blocks - 1 : blk;
    const uint16_t* vr = vc + (static_cast<int64_t>(blk) * num_kv_heads + kvh) * d * BS + t % BS;
    const float p = rnd<T>(s[t]);                   // probabilities are rounded to the model type before P.V, as on the GPU
    for (int j = 0; j < d; ++j) acc[j] += p * T::to_f(vr[static_cast<int64_t>(j) * BS]);
  }
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  for (int j = 0; j < d; ++j) out[j] = T::from_f(acc[j] * inv);
}

template <typename T>
void linear_rows(uint16_t* y, int64_t ys, const uint16_t* x, int64_t xs, const uint16_t* w, const uint16_t* bias, int M, int N,
                 int K) {
  // x as fp32 once; every weight row is converted once and dotted with all M rows
  std::vector<float> xf(static_cast<size_t>(M) * K);
  for (int m = 0; m < M; ++m)
    for (int k = 0; k < K; ++k) xf[static_cast<size_t>(m) * K + k] = T::to_f(x[m * xs + k]);
#pragma omp parallel
  {
    std::vector<float> wr(K);
#pragma omp for schedule(static)
    for (int n = 0; n < N; ++n) {
      const uint16_t* wp = w + static_cast<int64_t>(n) * K;
      for (int k = 0; k < K; ++k) wr[k] = T::to_f(wp[k]);
      const float b = bias ? T::to_f(bias[n]) : 0.f;
      for (int m = 0; m < M; ++m) {
        const float* xr = &xf[static_cast<size_t>(m) * K];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
        int k = 0;
        for (; k + 8 <= K; k += 8) {
          a0 += xr[k] * wr[k];
          a1 += xr[k + 1] * wr[k + 1];
          a2 += xr[k + 2] * wr[k + 2];
          a3 += xr[k + 3] * wr[k + 3];
          a4 += xr[k + 4] * wr[k + 4];
          a5 += xr[k + 5] * wr[k + 5];
          a6 += xr[k + 6] * wr[k + 6];
          a7 += xr[k + 7] * wr[k + 7];
        }
        float acc = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
        for (; k < K; ++k) acc += xr[k] * wr[k];
        y[m * ys + n] = T::from_f(acc + b);
      }
    }
  }
}

}  // namespace

extern "C" const char* mi355_cpu_last_error(void) { return g_err; }

extern "C" int32_t mi355_cpu_rms_norm(void* out, const void* x, const void* weight, int64_t num_tokens, int32_t hidden,
                                      int64_t x_stride, int64_t out_stride, float eps, int32_t dtype) {
  const int32_t rc = norm_checks("mi355_cpu_rms_norm", out, x, nullptr, false, weight, num_tokens, hidden, x_stride, out_stride,
                                 0, dtype);
  if (rc != MI355_OK) return rc;
  auto* o = static_cast<uint16_t*>(out);
  auto* xi = static_cast<const uint16_t*>(x);
  auto* w = static_cast<const uint16_t*>(weight);
  return dtype == MI355_BF16 ? rms_norm_impl<BF16>(o, xi, nullptr, w, num_tokens, hidden, x_stride, out_stride, 0, eps)
                             : rms_norm_impl<F16>(o, xi, nullptr, w, num_tokens, hidden, x_stride, out_stride, 0, eps);
}

extern "C" int32_t mi355_cpu_fused_add_rms_norm(void* out, const void* x, void* residual, const void* weight,
                                                int64_t num_tokens, int32_t hidden, int64_t x_stride, int64_t out_stride,
                                                int64_t residual_stride, float eps, int32_t dtype) {
  const int32_t rc = norm_checks("mi355_cpu_fused_add_rms_norm", out, x, residual, true, weight, num_tokens, hidden, x_stride,
                                 out_stride, residual_stride, dtype);
  if (rc != MI355_OK) return rc;
  auto* o = static_cast<uint16_t*>(out);
  auto* xi = static_cast<const uint16_t*>(x);
  auto* r = static_cast<uint16_t*>(residual);
  auto* w = static_cast<const uint16_t*>(weight);
  return dtype == MI355_BF16
             ? rms_norm_impl<BF16>(o, xi, r, w, num_tokens, hidden, x_stride, out_stride, residual_stride, eps)
             : rms_norm_impl<F16>(o, xi, r, w, num_tokens, hidden, x_stride, out_stride, residual_stride, eps);
}

extern "C" int32_t mi355_cpu_layer_norm(void* out, const void* x, const void* weight, const void* bias, int64_t num_tokens,
                                        int32_t hidden, int64_t x_stride, int64_t out_stride, float eps, int32_t dtype) {
  const char* name = "mi355_cpu_layer_norm";
  const int32_t rc = norm_checks(name, out, x, nullptr, false, weight, num_tokens, hidden, x_stride, out_stride, 0, dtype);
  if (rc != MI355_OK) return rc;
  CPU_REQUIRE(bias && aligned16(bias), MI355_ERR_INVALID, "%s: bias missing or misaligned", name);
  auto* o = static_cast<uint16_t*>(out);
  auto* xi = static_cast<const uint16_t*>(x);
  auto* w = static_cast<const uint16_t*>(weight);
  auto* b = static_cast<const uint16_t*>(bias);
  return dtype == MI355_BF16 ? layer_norm_impl<BF16>(o, xi, nullptr, w, b, num_tokens, hidden, x_stride, out_stride, 0, eps)
                             : layer_norm_impl<F16>(o, xi, nullptr, w, b, num_tokens, hidden, x_stride, out_stride, 0, eps);
}

extern "C" int32_t mi355_cpu_fused_add_layer_norm(void* out, const void* x, void* residual, const void* weight,
                                                  const void* bias, int64_t num_tokens, int32_t hidden, int64_t x_stride,
                                                  int64_t out_stride, int64_t residual_stride, float eps, int32_t dtype) {
  const char* name = "mi355_cpu_fused_add_layer_norm";
  const int32_t rc = norm_checks(name, out, x, residual, true, weight, num_tokens, hidden, x_stride, out_stride,
                                 residual_stride, dtype);
  if (rc != MI355_OK) return rc;
  CPU_REQUIRE(bias && aligned16(bias), MI355_ERR_INVALID, "%s: bias missing or misaligned", name);
  auto* o = static_cast<uint16_t*>(out);
  auto* xi = static_cast<const uint16_t*>(x);
  auto* r = static_cast<uint16_t*>(residual);
  auto* w = static_cast<const uint16_t*>(weight);
  auto* b = static_cast<const uint16_t*>(bias);
  return dtype == MI355_BF16
             ? layer_norm_impl<BF16>(o, xi, r, w, b, num_tokens, hidden, x_stride, out_stride, residual_stride, eps)
             : layer_norm_impl<F16>(o, xi, r, w, b, num_tokens, hidden, x_stride, out_stride, residual_stride, eps);
}

extern "C" int32_t mi355_cpu_relu(void* out, const void* x, int64_t num_elements, int32_t dtype) {
  const char* name = "mi355_cpu_relu";
  CPU_REQUIRE(out && x, MI355_ERR_INVALID, "%s: null pointer", name);
  CPU_REQUIRE(num_elements >= 0 && num_elements % 8 == 0, MI355_ERR_UNSUPPORTED,
              "%s: the element count %lld must be a non-negative multiple of 8", name, (long long)num_elements);
  CPU_REQUIRE(aligned16(out) && aligned16(x), MI355_ERR_INVALID, "%s: pointers must be 16-byte aligned", name);
  CPU_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED, "%s: dtype %d not supported (bf16/fp16 only)",
              name, dtype);
  auto* o = static_cast<uint16_t*>(out);
  auto* xi = static_cast<const uint16_t*>(x);
  for (int64_t i = 0; i < num_elements; ++i) {
    const float f = dtype == MI355_BF16 ? BF16::to_f(xi[i]) : F16::to_f(xi[i]);
    o[i] = f < 0.f ? static_cast<uint16_t>(0) : xi[i];       // NaN and -0.0 pass through
  }
  return MI355_OK;
}

template <typename T>
static void rope_impl(uint16_t* qkv, int64_t qs, const int64_t* pos, const uint16_t* cs, int max_pos, const int64_t* slots,
                      uint16_t* kc, uint16_t* vc, int num_blocks, int64_t T_, int nh, int nkv, int d) {
  constexpr int BS = MI355_BLOCK_SIZE;
  const int half = d / 2;
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < T_; ++t) {
    int64_t p = pos[t];
    p = p < 0 ? 0 : p >= max_pos ? max_pos - 1 : p;
    const uint16_t* c = cs + p * d;
    uint16_t* row = qkv + t * qs;
    for (int hd = 0; hd < nh + nkv; ++hd) {               // q heads, then k heads
      uint16_t* v = row + static_cast<int64_t>(hd) * d;
      for (int j = 0; j < half; ++j) {
        const float x1 = T::to_f(v[j]), x2 = T::to_f(v[j + half]);
        const float co = T::to_f(c[j]), si = T::to_f(c[half + j]);
        v[j] = T::from_f(rnd<T>(x1 * co) - rnd<T>(x2 * si));
        v[j + half] = T::from_f(rnd<T>(x2 * co) + rnd<T>(x1 * si));
      }
    }
    const int64_t slot = slots[t];
    if (slot < 0 || slot >= static_cast<int64_t>(num_blocks) * BS) continue;
    const int64_t blk = slot / BS;
    const int off = static_cast<int>(slot % BS);
    for (int kh = 0; kh < nkv; ++kh) {
      const uint16_t* k = row + static_cast<int64_t>(nh + kh) * d;
      const uint16_t* v = row + static_cast<int64_t>(nh + nkv + kh) * d;
      uint16_t* kd = kc + ((blk * nkv + kh) * BS + off) * d;
      uint16_t* vd = vc + (blk * nkv + kh) * d * BS + off;
      for (int j = 0; j < d; ++j) {
        kd[j] = k[j];
        vd[static_cast<int64_t>(j) * BS] = v[j];
      }
    }
  }
}

extern "C" int32_t mi355_cpu_rope_kv_write(void* qkv, int64_t qkv_stride, const int64_t* positions, const void* cos_sin,
                                           int32_t max_position, const int64_t* slot_mapping, void* k_cache, void* v_cache,
                                           int32_t num_blocks, int64_t num_tokens, int32_t num_heads, int32_t num_kv_heads,
                                           int32_t head_dim, int32_t dtype) {
  const char* name = "mi355_cpu_rope_kv_write";
  CPU_REQUIRE(qkv && positions && cos_sin && slot_mapping && k_cache && v_cache, MI355_ERR_INVALID, "%s: null pointer", name);
  const int32_t rc = cache_checks(name, num_heads, num_kv_heads, head_dim, dtype);
  if (rc != MI355_OK) return rc;
  CPU_REQUIRE(num_tokens >= 0 && num_blocks > 0 && max_position > 0, MI355_ERR_INVALID,
              "%s: num_tokens/num_blocks/max_position out of range", name);
  const int64_t width = static_cast<int64_t>(num_heads + 2 * num_kv_heads) * head_dim;
  CPU_REQUIRE(qkv_stride >= width && qkv_stride % 8 == 0 && aligned16(qkv) && aligned16(k_cache) && aligned16(v_cache) &&
                  aligned16(cos_sin),
              MI355_ERR_INVALID, "%s: qkv_stride %lld invalid or pointers misaligned", name, (long long)qkv_stride);
  auto* q = static_cast<uint16_t*>(qkv);
  auto* cs = static_cast<const uint16_t*>(cos_sin);
  auto* kc = static_cast<uint16_t*>(k_cache);
  auto* vc = static_cast<uint16_t*>(v_cache);
  if (dtype == MI355_BF16)
    rope_impl<BF16>(q, qkv_stride, positions, cs, max_position, slot_mapping, kc, vc, num_blocks, num_tokens, num_heads,
                    num_kv_heads, head_dim);
  else
    rope_impl<F16>(q, qkv_stride, positions, cs, max_position, slot_mapping, kc, vc, num_blocks, num_tokens, num_heads,
                   num_kv_heads, head_dim);
  return MI355_OK;
}

extern "C" int32_t mi355_cpu_paged_attn_decode(void* out, const void* q, int64_t q_stride, const void* k_cache,
                                               const void* v_cache, int32_t num_blocks, const int32_t* block_tables,
                                               int32_t max_blocks_per_seq, const int32_t* context_lens, int32_t num_seqs,
                                               int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, float scale,
                                               int32_t dtype) {
  const char* name = "mi355_cpu_paged_attn_decode";
  CPU_REQUIRE(out && q && k_cache && v_cache && block_tables && context_lens, MI355_ERR_INVALID, "%s: null pointer", name);
  CPU_REQUIRE(num_seqs >= 0 && num_blocks > 0 && max_blocks_per_seq > 0, MI355_ERR_INVALID,
              "%s: num_seqs/num_blocks/max_blocks_per_seq out of range", name);
  const int32_t rc = cache_checks(name, num_heads, num_kv_heads, head_dim, dtype);
  if (rc != MI355_OK) return rc;
  CPU_REQUIRE(q_stride >= static_cast<int64_t>(num_heads) * head_dim && q_stride % 8 == 0 && aligned16(q) && aligned16(out) &&
                  aligned16(k_cache) && aligned16(v_cache),
              MI355_ERR_INVALID, "%s: q_stride %lld invalid or pointers misaligned", name, (long long)q_stride);
  const int group = num_heads / num_kv_heads;
  auto* o = static_cast<uint16_t*>(out);
  auto* qp = static_cast<const uint16_t*>(q);
  auto* kc = static_cast<const uint16_t*>(k_cache);
  auto* vc = static_cast<const uint16_t*>(v_cache);
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int b = 0; b < num_seqs; ++b)
    for (int h = 0; h < num_heads; ++h) {
      int ctx = context_lens[b];
      const int cap = max_blocks_per_seq * MI355_BLOCK_SIZE;
      ctx = ctx < 0 ? 0 : ctx > cap ? cap : ctx;
      uint16_t* dst = o + (static_cast<int64_t>(b) * num_heads + h) * head_dim;
      if (ctx == 0) {
        memset(dst, 0, sizeof(uint16_t) * head_dim);
        continue;
      }
      const uint16_t* qr = qp + b * q_stride + static_cast<int64_t>(h) * head_dim;
      const int32_t* bt = block_tables + static_cast<int64_t>(b) * max_blocks_per_seq;
      if (dtype == MI355_BF16)
        attend<BF16>(dst, qr, kc, vc, num_blocks, bt, ctx, h / group, num_kv_heads, head_dim, scale);
      else
        attend<F16>(dst, qr, kc, vc, num_blocks, bt, ctx, h / group, num_kv_heads, head_dim, scale);
    }
  return MI355_OK;
}

extern "C" int32_t mi355_cpu_paged_attn_prefill(void* out, int64_t out_stride, const void* q, int64_t q_stride,
                                                const void* k_cache, const void* v_cache, int32_t num_blocks,
                                                const int32_t* block_tables, int32_t max_blocks_per_seq,
                                                const int32_t* cu_seqlens, const int32_t* context_lens, int32_t num_seqs,
                                                int32_t max_q_len, int32_t num_heads, int32_t num_kv_heads, int32_t head_dim,
                                                float scale, int32_t dtype) {
  const char* name = "mi355_cpu_paged_attn_prefill";
  CPU_REQUIRE(out && q && k_cache && v_cache && block_tables && cu_seqlens && context_lens, MI355_ERR_INVALID,
              "%s: null pointer", name);
  CPU_REQUIRE(num_seqs >= 0 && num_seqs <= 65535 && num_blocks > 0 && max_blocks_per_seq > 0 && max_q_len >= 0,
              MI355_ERR_INVALID, "%s: num_seqs/num_blocks/max_blocks_per_seq/max_q_len out of range", name);
  const int32_t rc = cache_checks(name, num_heads, num_kv_heads, head_dim, dtype);
  if (rc != MI355_OK) return rc;
  const int64_t width = static_cast<int64_t>(num_heads) * head_dim;
  CPU_REQUIRE(q_stride >= width && q_stride % 8 == 0 && out_stride >= width && out_stride % 8 == 0, MI355_ERR_INVALID,
              "%s: q_stride/out_stride invalid", name);
  CPU_REQUIRE(aligned16(out) && aligned16(q) && aligned16(k_cache) && aligned16(v_cache), MI355_ERR_INVALID,
              "%s: pointers must be 16-byte aligned", name);
  const int group = num_heads / num_kv_heads;
  auto* o = static_cast<uint16_t*>(out);
  auto* qp = static_cast<const uint16_t*>(q);
  auto* kc = static_cast<const uint16_t*>(k_cache);
  auto* vc = static_cast<const uint16_t*>(v_cache);
  for (int s = 0; s < num_seqs; ++s) {
    const int q_begin = cu_seqlens[s], q_len = cu_seqlens[s + 1] - q_begin;
    int ctx = context_lens[s];
    const int cap = max_blocks_per_seq * MI355_BLOCK_SIZE;
    ctx = ctx > cap ? cap : ctx;
    const int pos0 = ctx - q_len;
    if (q_len <= 0 || pos0 < 0) continue;
    const int32_t* bt = block_tables + static_cast<int64_t>(s) * max_blocks_per_seq;
    const int rows = q_len < max_q_len ? q_len : max_q_len;      // rows beyond max_q_len are not computed, as on the GPU
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int r = 0; r < rows; ++r)
      for (int h = 0; h < num_heads; ++h) {
        uint16_t* dst = o + (static_cast<int64_t>(q_begin) + r) * out_stride + static_cast<int64_t>(h) * head_dim;
        const uint16_t* qr = qp + (static_cast<int64_t>(q_begin) + r) * q_stride + static_cast<int64_t>(h) * head_dim;
        if (dtype == MI355_BF16)
          attend<BF16>(dst, qr, kc, vc, num_blocks, bt, pos0 + r + 1, h / group, num_kv_heads, head_dim, scale);
        else
          attend<F16>(dst, qr, kc, vc, num_blocks, bt, pos0 + r + 1, h / group, num_kv_heads, head_dim, scale);
      }
  }
  return MI355_OK;
}

template <typename T>
static void silu_impl(uint16_t* out, const uint16_t* x, int64_t xs, int64_t T_, int inter) {
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < T_; ++t)
    for (int j = 0; j < inter; ++j) {
      const float g = T::to_f(x[t * xs + j]), u = T::to_f(x[t * xs + inter + j]);
      const float s = rnd<T>(g / (1.0f + expf(-g)));
      out[t * inter + j] = T::from_f(s * u);
    }
}

extern "C" int32_t mi355_cpu_silu_and_mul(void* out, const void* x, int64_t x_stride, int64_t num_tokens, int32_t inter,
                                          int32_t dtype) {
  const char* name = "mi355_cpu_silu_and_mul";
  CPU_REQUIRE(out && x, MI355_ERR_INVALID, "%s: null pointer", name);
  CPU_REQUIRE(num_tokens >= 0 && inter > 0 && inter % 8 == 0, MI355_ERR_UNSUPPORTED,
              "%s: inter %d must be a positive multiple of 8", name, inter);
  CPU_REQUIRE(x_stride >= 2 * static_cast<int64_t>(inter) && x_stride % 8 == 0 && aligned16(out) && aligned16(x),
              MI355_ERR_INVALID, "%s: x_stride %lld invalid or pointers misaligned", name, (long long)x_stride);
  CPU_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED, "%s: dtype %d not supported (bf16/fp16 only)",
              name, dtype);
  if (dtype == MI355_BF16)
    silu_impl<BF16>(static_cast<uint16_t*>(out), static_cast<const uint16_t*>(x), x_stride, num_tokens, inter);
  else
    silu_impl<F16>(static_cast<uint16_t*>(out), static_cast<const uint16_t*>(x), x_stride, num_tokens, inter);
  return MI355_OK;
}

extern "C" int32_t mi355_cpu_argmax(int64_t* out_ids, const void* logits, int64_t logits_stride, int32_t num_rows,
                                    int32_t vocab, int32_t logits_dtype) {
  const char* name = "mi355_cpu_argmax";
  CPU_REQUIRE(out_ids && logits, MI355_ERR_INVALID, "%s: null pointer", name);
  CPU_REQUIRE(num_rows >= 0 && vocab > 0, MI355_ERR_INVALID, "%s: num_rows/vocab out of range", name);
  CPU_REQUIRE(logits_dtype == MI355_BF16 || logits_dtype == MI355_F16 || logits_dtype == MI355_F32, MI355_ERR_UNSUPPORTED,
              "%s: logits dtype %d not supported", name, logits_dtype);
  const int per16 = logits_dtype == MI355_F32 ? 4 : 8;
  CPU_REQUIRE(logits_stride >= vocab && logits_stride % per16 == 0 && aligned16(logits), MI355_ERR_INVALID,
              "%s: logits rows must be 16-byte aligned (stride %lld)", name, (long long)logits_stride);
#pragma omp parallel for schedule(static)
  for (int r = 0; r < num_rows; ++r) {
    int best = 0;
    float bv = 0.f;
    for (int i = 0; i < vocab; ++i) {
      float v;
      if (logits_dtype == MI355_F32)
        v = static_cast<const float*>(logits)[r * logits_stride + i];
      else if (logits_dtype == MI355_BF16)
        v = BF16::to_f(static_cast<const uint16_t*>(logits)[r * logits_stride + i]);
      else
        v = F16::to_f(static_cast<const uint16_t*>(logits)[r * logits_stride + i]);
      if (v != v) {                                    // a NaN counts as the maximum: the first one wins
        best = i;
        break;
      }
      if (i == 0 || v > bv) {                          // strict: the lowest index among equals (-0.0 == +0.0)
        bv = v;
        best = i;
      }
    }
    out_ids[r] = best;
  }
  return MI355_OK;
}

extern "C" int32_t mi355_cpu_linear(void* y, int64_t y_stride, const void* x, int64_t x_stride, const void* weight,
                                    const void* bias, int32_t M, int32_t N, int32_t K, int32_t dtype) {
  const char* name = "mi355_cpu_linear";
  CPU_REQUIRE(y && x && weight, MI355_ERR_INVALID, "%s: null pointer", name);
  CPU_REQUIRE(M >= 0 && N > 0 && K > 0, MI355_ERR_INVALID, "%s: M/N/K out of range", name);
  CPU_REQUIRE(x_stride >= K && y_stride >= N, MI355_ERR_INVALID, "%s: strides too small", name);
  CPU_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED, "%s: dtype %d not supported (bf16/fp16 only)",
              name, dtype);
  if (M == 0) return MI355_OK;
  if (dtype == MI355_BF16)
    linear_rows<BF16>(static_cast<uint16_t*>(y), y_stride, static_cast<const uint16_t*>(x), x_stride,
                      static_cast<const uint16_t*>(weight), static_cast<const uint16_t*>(bias), M, N, K);
  else
    linear_rows<F16>(static_cast<uint16_t*>(y), y_stride, static_cast<const uint16_t*>(x), x_stride,
                     static_cast<const uint16_t*>(weight), static_cast<const uint16_t*>(bias), M, N, K);
  return MI355_OK;
}
