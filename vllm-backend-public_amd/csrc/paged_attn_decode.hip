// Decode attention over the paged KV cache (one query token per sequence).
//
// Work decomposition.  One 256-thread workgroup (4 waves) per
// (sequence, kv head, context split).  The query heads that share the kv head
// (the GQA group, <= 16) are processed together so that every K/V page is read
// from HBM once per group.  Inside a workgroup the waves take 32-token steps of
// the split's token range round-robin; each wave runs an online softmax over
// its steps and the four partial results are merged through LDS.  With more
// than one split the merged partials go to a workspace and a second small
// kernel combines them.
//
// MFMA mapping (v_mfma_f32_16x16x32, heads on the 16 output COLUMNS throughout,
// so softmax statistics, rescaling and the accumulator all live on the same
// lane and only the column max/sum need cross-lane shuffles):
//   S^T[token, head] = K[token, :] . Q[head, :]
//        A = K tile: lane (i = lane & 15, h = lane >> 4) loads 16 bytes of token
//            row i at elements [32*ks + 8h, +8)  -- straight from HBM to VGPRs,
//            no LDS; tile "a" holds tokens 8*(i>>2) + (i&3), tile "b" those + 4,
//            so that afterwards lane (head, h) owns the 8 CONSECUTIVE tokens
//            8h .. 8h+7 of the step
//        B = Q^T, the group's heads zero-padded to 16 columns (registers)
//   O^T[d, head] += V^T[d, token] . P^T[token, head]
//        A = V page, which is stored [head_dim][16 tokens]: lane (i, h) loads
//            the 16 bytes of row d = 16n + i holding tokens 8h .. 8h+7
//        B = the lane's own 8 probabilities, packed to 16 bits
// HBM-bound: 2 * ctx * head_dim * 2 bytes per (sequence, kv head); MFMA use is
// a few percent.
#include "common.h"

#include <math.h>

namespace mi355 {

constexpr int kDecodeThreads = 256;
constexpr int kDecodeWaves = kDecodeThreads / 64;
constexpr int kStep = 32;  // tokens per wave step (two 16-token pages)

template <typename T, int D>
__global__ __launch_bounds__(kDecodeThreads) void paged_attn_decode_kernel(
    uint16_t* __restrict__ out, float* __restrict__ part_o, float* __restrict__ part_ml,
    const uint16_t* __restrict__ q, int64_t q_stride, const uint16_t* __restrict__ k_cache,
    const uint16_t* __restrict__ v_cache, int num_blocks, const int32_t* __restrict__ block_tables,
    int max_blocks, const int32_t* __restrict__ context_lens, int num_heads, int num_kv_heads,
    float scale_log2e, int num_splits) {
  constexpr int KS = D / 32;  // MFMA k-steps over head_dim for QK^T
  constexpr int NT = D / 16;  // 16-row tiles of head_dim for PV
  constexpr int BS = MI355_BLOCK_SIZE;

  const int seq = blockIdx.x, kvh = blockIdx.y, split = blockIdx.z;
  const int group = num_heads / num_kv_heads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, h = lane >> 4;

  int ctx = context_lens[seq];
  const int ctx_cap = max_blocks * BS;
  ctx = ctx < 0 ? 0 : (ctx > ctx_cap ? ctx_cap : ctx);
  // token range of this split: multiples of one workgroup pass (4 waves x 32)
  int per_split = (ctx + num_splits - 1) / num_splits;
  per_split = (per_split + kStep * kDecodeWaves - 1) / (kStep * kDecodeWaves) * (kStep * kDecodeWaves);
  const int t_begin = split * per_split;
  const int t_end = min(ctx, t_begin + per_split);

  // Q fragments (B operand): column c = head within the group
  uint4 qf[KS];
  if (c < group) {
    const uint16_t* qp = q + seq * q_stride + static_cast<int64_t>(kvh * group + c) * D + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const uint4*>(qp + 32 * ks);
  } else {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = make_uint4(0, 0, 0, 0);
  }

  float m = -INFINITY, l = 0.f;
  f32x4 o[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int32_t* bt = block_tables + static_cast<int64_t>(seq) * max_blocks;
  const int last_blk_idx = ctx > 0 ? (ctx - 1) / BS : 0;
  const int koff = ((c >> 2) & 1) * 8 + (c & 3);  // token row inside the page for tile "a"

  for (int tb = t_begin + kStep * wave; tb < t_end; tb += kStep * kDecodeWaves) {
    const int bi = tb / BS;
    int blk0 = bt[bi];
    int blk1 = bt[min(bi + 1, last_blk_idx)];
    blk0 = min(max(blk0, 0), num_blocks - 1);
    blk1 = min(max(blk1, 0), num_blocks - 1);

    // ---- issue all loads of the step: K tiles a/b, then the V tiles
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

    // ---- S^T = K . Q^T
    f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      sa = T::mfma(ka[ks], qf[ks], sa);
      sb = T::mfma(kb[ks], qf[ks], sb);
    }
    // lane (head c, h) now holds tokens tb + 8h + j, j = 0..7 (sa: j<4, sb: j>=4)
    float s[8];
    const int tok0 = tb + 8 * h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] = (tok0 + j < t_end) ? sa[j] * scale_log2e : -INFINITY;
      s[4 + j] = (tok0 + 4 + j < t_end) ? sb[j] * scale_log2e : -INFINITY;
    }
    float mx = s[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) mx = fmaxf(mx, s[j]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // token tb is always inside the range, so mx is finite
    const float m_new = fmaxf(m, mx);
    const float alpha = exp2f(m - m_new);
    float p[8], psum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      p[j] = exp2f(s[j] - m_new);
      psum += p[j];
    }
    l = l * alpha + psum;
    m = m_new;
#pragma unroll
    for (int n = 0; n < NT; ++n) o[n] *= alpha;
    const uint4 pf = pack8<T>(p);

    // last, partial step: V columns of tokens beyond the context may hold
    // anything (NaN included) -- clear them so 0 * garbage cannot reach O
    if (tb + kStep > t_end) {
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
    // ---- O^T += V^T . P^T
#pragma unroll
    for (int n = 0; n < NT; ++n) o[n] = T::mfma(vf[n], pf, o[n]);
  }

  // column sums live on the 4 lanes (c, h = 0..3)
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);

  // ---- merge the 4 waves through LDS: O as [wave][d][16 heads]
  __shared__ float lds_o[kDecodeWaves][D * 16];
  __shared__ float lds_m[kDecodeWaves][16];
  __shared__ float lds_l[kDecodeWaves][16];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) lds_o[wave][(16 * n + 4 * h + r) * 16 + c] = o[n][r];
  if (h == 0) {
    lds_m[wave][c] = m;
    lds_l[wave][c] = l;
  }
  __syncthreads();

  for (int idx = threadIdx.x; idx < group * D; idx += kDecodeThreads) {
    const int hh = idx / D, d = idx % D;
    float mg = lds_m[0][hh];
#pragma unroll
    for (int w = 1; w < kDecodeWaves; ++w) mg = fmaxf(mg, lds_m[w][hh]);
    float acc = 0.f, lsum = 0.f;
    if (mg != -INFINITY) {
#pragma unroll
      for (int w = 0; w < kDecodeWaves; ++w) {
        const float f = exp2f(lds_m[w][hh] - mg);
        acc += f * lds_o[w][d * 16 + hh];
        lsum += f * lds_l[w][hh];
      }
    }
    const int head = kvh * group + hh;
    if (num_splits == 1) {
      const float val = lsum > 0.f ? acc / lsum : 0.f;
      out[(static_cast<int64_t>(seq) * num_heads + head) * D + d] = T::from_f(val);
    } else {
      const int64_t pi = (static_cast<int64_t>(seq) * num_heads + head) * num_splits + split;
      part_o[pi * D + d] = acc;
      if (d == 0) {
        part_ml[pi * 2] = mg;
        part_ml[pi * 2 + 1] = lsum;
      }
    }
  }
}

// Combine the per-split partials: one workgroup per (sequence, head), D threads.
template <typename T, int D>
__global__ __launch_bounds__(D) void paged_attn_merge_kernel(uint16_t* __restrict__ out,
                                                             const float* __restrict__ part_o,
                                                             const float* __restrict__ part_ml,
                                                             int num_splits) {
  const int64_t row = blockIdx.x;  // seq * num_heads + head
  const int d = threadIdx.x;
  const float* ml = part_ml + row * num_splits * 2;
  float mg = -INFINITY;
  for (int s = 0; s < num_splits; ++s) mg = fmaxf(mg, ml[2 * s]);
  float acc = 0.f, lsum = 0.f;
  if (mg != -INFINITY) {
    for (int s = 0; s < num_splits; ++s) {
      const float f = exp2f(ml[2 * s] - mg);
      acc += f * part_o[(row * num_splits + s) * D + d];
      lsum += f * ml[2 * s + 1];
    }
  }
  out[row * D + d] = T::from_f(lsum > 0.f ? acc / lsum : 0.f);
}

template <typename T, int D>
static int32_t launch_decode(void* out, const void* q, int64_t q_stride, const void* k_cache,
                             const void* v_cache, int32_t num_blocks, const int32_t* block_tables,
                             int32_t max_blocks, const int32_t* context_lens, int32_t num_seqs,
                             int32_t num_heads, int32_t num_kv_heads, float scale, int32_t num_splits,
                             void* workspace, hipStream_t s) {
  float* part_o = nullptr;
  float* part_ml = nullptr;
  if (num_splits > 1) {
    part_o = static_cast<float*>(workspace);
    part_ml = part_o + static_cast<int64_t>(num_seqs) * num_heads * num_splits * D;
  }
  dim3 grid(num_seqs, num_kv_heads, num_splits), block(kDecodeThreads);
  hipLaunchKernelGGL((paged_attn_decode_kernel<T, D>), grid, block, 0, s, static_cast<uint16_t*>(out),
                     part_o, part_ml, static_cast<const uint16_t*>(q), q_stride,
                     static_cast<const uint16_t*>(k_cache), static_cast<const uint16_t*>(v_cache),
                     num_blocks, block_tables, max_blocks, context_lens, num_heads, num_kv_heads,
                     scale * 1.4426950408889634f, num_splits);
  int32_t rc = check_launch("mi355_paged_attn_decode");
  if (rc != MI355_OK || num_splits == 1) return rc;
  hipLaunchKernelGGL((paged_attn_merge_kernel<T, D>), dim3(num_seqs * num_heads), dim3(D), 0, s,
                     static_cast<uint16_t*>(out), part_o, part_ml, num_splits);
  return check_launch("mi355_paged_attn_decode(merge)");
}

}  // namespace mi355

extern "C" int64_t mi355_paged_attn_decode_workspace(int32_t num_seqs, int32_t num_heads,
                                                     int32_t head_dim, int32_t num_splits) {
  if (num_splits <= 1 || num_seqs <= 0 || num_heads <= 0 || head_dim <= 0) return 0;
  return static_cast<int64_t>(num_seqs) * num_heads * num_splits * (head_dim + 2) * sizeof(float);
}

extern "C" int32_t mi355_paged_attn_decode(void* out, const void* q, int64_t q_stride, const void* k_cache,
                                           const void* v_cache, int32_t num_blocks,
                                           const int32_t* block_tables, int32_t max_blocks_per_seq,
                                           const int32_t* context_lens, int32_t num_seqs,
                                           int32_t num_heads, int32_t num_kv_heads, int32_t head_dim,
                                           float scale, int32_t num_splits, void* workspace,
                                           int64_t workspace_bytes, int32_t dtype, void* stream) {
  using namespace mi355;
  const char* name = "mi355_paged_attn_decode";
  MI355_REQUIRE(out && q && k_cache && v_cache && block_tables && context_lens, MI355_ERR_INVALID,
                "%s: null pointer", name);
  MI355_REQUIRE(num_seqs >= 0 && num_blocks > 0 && max_blocks_per_seq > 0, MI355_ERR_INVALID,
                "%s: num_seqs/num_blocks/max_blocks_per_seq out of range", name);
  MI355_REQUIRE(num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0, MI355_ERR_INVALID,
                "%s: num_heads %d must be a positive multiple of num_kv_heads %d", name, num_heads,
                num_kv_heads);
  MI355_REQUIRE(num_heads / num_kv_heads <= 16, MI355_ERR_UNSUPPORTED,
                "%s: at most 16 query heads per kv head (got %d)", name, num_heads / num_kv_heads);
  MI355_REQUIRE(num_kv_heads <= 65535 && num_splits >= 1 && num_splits <= 64, MI355_ERR_INVALID,
                "%s: num_kv_heads <= 65535 and 1 <= num_splits <= 64 required", name);
  MI355_REQUIRE(head_dim == 64 || head_dim == 128, MI355_ERR_UNSUPPORTED,
                "%s: head_dim %d not supported (64 or 128)", name, head_dim);
  MI355_REQUIRE(q_stride >= static_cast<int64_t>(num_heads) * head_dim && q_stride % 8 == 0,
                MI355_ERR_INVALID, "%s: q_stride %lld invalid", name, (long long)q_stride);
  MI355_REQUIRE(aligned16(out) && aligned16(q) && aligned16(k_cache) && aligned16(v_cache),
                MI355_ERR_INVALID, "%s: pointers must be 16-byte aligned", name);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "%s: dtype %d not supported (bf16/fp16 only)", name, dtype);
  if (num_splits > 1) {
    const int64_t need = mi355_paged_attn_decode_workspace(num_seqs, num_heads, head_dim, num_splits);
    MI355_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= need, MI355_ERR_INVALID,
                  "%s: workspace of %lld bytes required for %d splits (got %lld)", name, (long long)need,
                  num_splits, (long long)workspace_bytes);
  }
  if (num_seqs == 0) return MI355_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define MI355_DEC(TT, DD)                                                                              \
  return launch_decode<TT, DD>(out, q, q_stride, k_cache, v_cache, num_blocks, block_tables,          \
                               max_blocks_per_seq, context_lens, num_seqs, num_heads, num_kv_heads,   \
                               scale, num_splits, workspace, s)
  if (dtype == MI355_BF16) {
    if (head_dim == 64) MI355_DEC(BF16, 64);
    MI355_DEC(BF16, 128);
  }
  if (head_dim == 64) MI355_DEC(F16, 64);
  MI355_DEC(F16, 128);
#undef MI355_DEC
}
