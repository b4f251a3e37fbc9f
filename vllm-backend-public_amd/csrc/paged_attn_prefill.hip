// Prefill / chunked-prefill attention: causal attention of packed query chunks
// over keys and values that are already in the paged KV cache.
//
// Work decomposition.  One 256-thread workgroup (4 waves) per
// (16-query tile, kv head, sequence).  The tile's (query, head-in-group) pairs
// -- 16 * group of them -- are laid on MFMA output COLUMNS, 16 per column tile;
// wave w owns column tiles w*NQ .. w*NQ+NQ-1 and walks ALL keys the tile can see
// in 32-token steps with an online softmax (no cross-wave merge; the four waves
// read the same K/V pages, which hit in L1/L2 after the first touch).
//
// MFMA mapping is the decode kernel's (paged_attn_decode.hip): S^T = K . Q^T
// with K rows loaded straight from the page into the A operand, and
// O^T += V^T . P^T with the transposed V page as the A operand and the lane's
// own probabilities as B.  The causal mask is applied per column
// (key <= position of the column's query).  MFMA-bound in principle; this
// version re-reads K/V through L1 per wave and is tuned for correctness first.
#include "common.h"

#include <math.h>

namespace mi355 {

constexpr int kPrefillThreads = 256;
constexpr int kQTile = 16;

template <typename T, int D, int NQ>
__global__ __launch_bounds__(kPrefillThreads) void paged_attn_prefill_kernel(
    uint16_t* __restrict__ out, int64_t out_stride, const uint16_t* __restrict__ q, int64_t q_stride,
    const uint16_t* __restrict__ k_cache, const uint16_t* __restrict__ v_cache, int num_blocks,
    const int32_t* __restrict__ block_tables, int max_blocks, const int32_t* __restrict__ cu_seqlens,
    const int32_t* __restrict__ context_lens, int num_heads, int num_kv_heads, float scale_log2e) {
  constexpr int KS = D / 32;
  constexpr int NT = D / 16;
  constexpr int BS = MI355_BLOCK_SIZE;

  const int qtile = blockIdx.x, kvh = blockIdx.y, seq = blockIdx.z;
  const int group = num_heads / num_kv_heads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, h = lane >> 4;

  const int q_begin = cu_seqlens[seq];
  const int q_len = cu_seqlens[seq + 1] - q_begin;
  const int q0 = qtile * kQTile;
  if (q0 >= q_len) return;
  int ctx = context_lens[seq];
  const int ctx_cap = max_blocks * BS;
  ctx = ctx > ctx_cap ? ctx_cap : ctx;
  const int pos0 = ctx - q_len;  // absolute position of the chunk's first query
  if (pos0 < 0) return;          // inconsistent lengths: nothing sensible to compute

  // this lane's columns: pair index p = (wave*NQ + t)*16 + c -> (query, head)
  int col_q[NQ], col_head[NQ], col_pos[NQ];
  bool col_ok[NQ];
  uint4 qf[NQ][KS];
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    const int p = (wave * NQ + t) * 16 + c;
    col_q[t] = q0 + p / group;
    col_head[t] = kvh * group + p % group;
    col_ok[t] = (p < kQTile * group) && (col_q[t] < q_len);
    col_pos[t] = col_ok[t] ? pos0 + col_q[t] : -1;
    if (col_ok[t]) {
      const uint16_t* qp =
          q + (q_begin + col_q[t]) * q_stride + static_cast<int64_t>(col_head[t]) * D + 8 * h;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) qf[t][ks] = *reinterpret_cast<const uint4*>(qp + 32 * ks);
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) qf[t][ks] = make_uint4(0, 0, 0, 0);
    }
  }
  // keys this wave needs: up to the largest position among its columns
  int kmax = -1;
#pragma unroll
  for (int t = 0; t < NQ; ++t) kmax = max(kmax, col_pos[t]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) kmax = max(kmax, __shfl_xor(kmax, off, 64));
  const int t_end = kmax + 1;  // wave-uniform; 0 when the wave has no valid column

  float m[NQ], l[NQ];
  f32x4 o[NQ][NT];
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    m[t] = -INFINITY;
    l[t] = 0.f;
#pragma unroll
    for (int n = 0; n < NT; ++n) o[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int32_t* bt = block_tables + static_cast<int64_t>(seq) * max_blocks;
  const int last_blk_idx = t_end > 0 ? (t_end - 1) / BS : 0;
  const int koff = ((c >> 2) & 1) * 8 + (c & 3);

  for (int tb = 0; tb < t_end; tb += 32) {
    const int bi = tb / BS;
    int blk0 = bt[bi];
    int blk1 = bt[min(bi + 1, last_blk_idx)];
    blk0 = min(max(blk0, 0), num_blocks - 1);
    blk1 = min(max(blk1, 0), num_blocks - 1);

    const int kblk = (c & 8) ? blk1 : blk0;
    const uint16_t* kp =
        k_cache + ((static_cast<int64_t>(kblk) * num_kv_heads + kvh) * BS + koff) * D + 8 * h;
    uint4 ka[KS], kb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      ka[ks] = *reinterpret_cast<const uint4*>(kp + 32 * ks);
      kb[ks] = *reinterpret_cast<const uint4*>(kp + 4 * D + 32 * ks);
    }
    const int vblk = (h & 2) ? blk1 : blk0;
    const uint16_t* vp =
        v_cache + ((static_cast<int64_t>(vblk) * num_kv_heads + kvh) * D + c) * BS + 8 * (h & 1);
    uint4 vf[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) vf[n] = *reinterpret_cast<const uint4*>(vp + n * 16 * BS);

    const int tok0 = tb + 8 * h;
    if (tb + 32 > t_end) {  // clear V columns of tokens no column can see (may be unwritten)
      uint32_t msk[4];
#pragma unroll
      for (int w = 0; w < 4; ++w)
        msk[w] = ((tok0 + 2 * w < t_end) ? 0x0000FFFFu : 0u) | ((tok0 + 2 * w + 1 < t_end) ? 0xFFFF0000u : 0u);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        vf[n].x &= msk[0];
        vf[n].y &= msk[1];
        vf[n].z &= msk[2];
        vf[n].w &= msk[3];
      }
    }

#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        sa = T::mfma(ka[ks], qf[t][ks], sa);
        sb = T::mfma(kb[ks], qf[t][ks], sb);
      }
      float s[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s[j] = (tok0 + j <= col_pos[t]) ? sa[j] * scale_log2e : -INFINITY;
        s[4 + j] = (tok0 + 4 + j <= col_pos[t]) ? sb[j] * scale_log2e : -INFINITY;
      }
      float mx = s[0];
#pragma unroll
      for (int j = 1; j < 8; ++j) mx = fmaxf(mx, s[j]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      // a column whose keys are all masked in this step keeps its state
      const float m_new = fmaxf(m[t], mx);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = exp2f(m[t] - m_use);
      float p[8], psum = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        p[j] = exp2f(s[j] - m_use);
        psum += p[j];
      }
      l[t] = l[t] * alpha + psum;
      m[t] = m_new;
      const uint4 pf = pack8<T>(p);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        o[t][n] *= alpha;
        o[t][n] = T::mfma(vf[n], pf, o[t][n]);
      }
    }
  }

#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    float lt = l[t];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    if (!col_ok[t]) continue;
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;
    uint16_t* op = out + (q_begin + col_q[t]) * out_stride + static_cast<int64_t>(col_head[t]) * D + 4 * h;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      // lane (pair c, h) holds d = 16n + 4h + r, r = 0..3: one 8-byte store
      const uint32_t w0 = static_cast<uint32_t>(T::from_f(o[t][n][0] * inv)) |
                          (static_cast<uint32_t>(T::from_f(o[t][n][1] * inv)) << 16);
      const uint32_t w1 = static_cast<uint32_t>(T::from_f(o[t][n][2] * inv)) |
                          (static_cast<uint32_t>(T::from_f(o[t][n][3] * inv)) << 16);
      *reinterpret_cast<uint2*>(op + 16 * n) = make_uint2(w0, w1);
    }
  }
}

}  // namespace mi355

extern "C" int32_t mi355_paged_attn_prefill(void* out, int64_t out_stride, const void* q, int64_t q_stride,
                                            const void* k_cache, const void* v_cache, int32_t num_blocks,
                                            const int32_t* block_tables, int32_t max_blocks_per_seq,
                                            const int32_t* cu_seqlens, const int32_t* context_lens,
                                            int32_t num_seqs, int32_t max_q_len, int32_t num_heads,
                                            int32_t num_kv_heads, int32_t head_dim, float scale,
                                            int32_t dtype, void* stream) {
  using namespace mi355;
  const char* name = "mi355_paged_attn_prefill";
  MI355_REQUIRE(out && q && k_cache && v_cache && block_tables && cu_seqlens && context_lens,
                MI355_ERR_INVALID, "%s: null pointer", name);
  MI355_REQUIRE(num_seqs >= 0 && num_seqs <= 65535 && num_blocks > 0 && max_blocks_per_seq > 0 &&
                    max_q_len >= 0,
                MI355_ERR_INVALID, "%s: num_seqs/num_blocks/max_blocks_per_seq/max_q_len out of range", name);
  MI355_REQUIRE(num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0, MI355_ERR_INVALID,
                "%s: num_heads %d must be a positive multiple of num_kv_heads %d", name, num_heads,
                num_kv_heads);
  const int group = num_heads / num_kv_heads;
  MI355_REQUIRE(group <= 16 && num_kv_heads <= 65535, MI355_ERR_UNSUPPORTED,
                "%s: at most 16 query heads per kv head (got %d)", name, group);
  MI355_REQUIRE(head_dim == 64 || head_dim == 128, MI355_ERR_UNSUPPORTED,
                "%s: head_dim %d not supported (64 or 128)", name, head_dim);
  const int64_t width = static_cast<int64_t>(num_heads) * head_dim;
  MI355_REQUIRE(q_stride >= width && q_stride % 8 == 0 && out_stride >= width && out_stride % 8 == 0,
                MI355_ERR_INVALID, "%s: q_stride/out_stride invalid", name);
  MI355_REQUIRE(aligned16(out) && aligned16(q) && aligned16(k_cache) && aligned16(v_cache),
                MI355_ERR_INVALID, "%s: pointers must be 16-byte aligned", name);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "%s: dtype %d not supported (bf16/fp16 only)", name, dtype);
  if (num_seqs == 0 || max_q_len == 0) return MI355_OK;

  const int nq = (kQTile * group + 63) / 64;  // column tiles per wave: 1..4
  dim3 grid((max_q_len + kQTile - 1) / kQTile, num_kv_heads, num_seqs), block(kPrefillThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define MI355_PRE(TT, DD, NN)                                                                           \
  hipLaunchKernelGGL((paged_attn_prefill_kernel<TT, DD, NN>), grid, block, 0, s,                        \
                     static_cast<uint16_t*>(out), out_stride, static_cast<const uint16_t*>(q), q_stride, \
                     static_cast<const uint16_t*>(k_cache), static_cast<const uint16_t*>(v_cache),      \
                     num_blocks, block_tables, max_blocks_per_seq, cu_seqlens, context_lens, num_heads, \
                     num_kv_heads, scale * 1.4426950408889634f)
#define MI355_PRE_NQ(TT, DD)                \
  switch (nq) {                             \
    case 1: MI355_PRE(TT, DD, 1); break;    \
    case 2: MI355_PRE(TT, DD, 2); break;    \
    case 3: MI355_PRE(TT, DD, 3); break;    \
    default: MI355_PRE(TT, DD, 4); break;   \
  }
  if (dtype == MI355_BF16) {
    if (head_dim == 64) { MI355_PRE_NQ(BF16, 64) } else { MI355_PRE_NQ(BF16, 128) }
  } else {
    if (head_dim == 64) { MI355_PRE_NQ(F16, 64) } else { MI355_PRE_NQ(F16, 128) }
  }
#undef MI355_PRE_NQ
#undef MI355_PRE
  return check_launch(name);
}
