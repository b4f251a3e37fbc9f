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
//
// Fused form (FUSE != 0): the rotary embedding and the KV store of the new token
// happen here instead of in a kernel of their own.  The projection row of the
// (sequence, kv head) -- the group's q heads, the k head, the v head -- is formed
// once per workgroup: one thread per 8-element chunk (pair of chunks for the
// rotating rows) loads it with coalesced loads, rotates q and k, and hands the
// 16-bit row to the four waves through LDS, which read their MFMA operands
// from there.  The workgroup whose token range holds the new token stores k and
// v to their pages; the step that covers the new token takes its k row / v column
// from registers, whatever the pages held, so nothing waits for that store.  The
// projection comes either as packed 16-bit rows (FUSE == 1) or as the fp32
// partial-sum slabs of a K-split GEMM (FUSE == 2: slab sum + bias, rounded, first).
#include "common.h"

#include <math.h>
#include <stdlib.h>

namespace mi355 {

#ifdef MI355_ATTN_TRACE
// experiment build: per-workgroup time stamps (100 MHz wall clock and shader clock) of the phases
__device__ unsigned long long g_attn_trace[4096 * 16];
#define MI355_STAMP(slot)                                                                                  \
  do {                                                                                                     \
    if (threadIdx.x == 0) {                                                                                \
      const int wg_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                      \
      if (wg_ < 4096) {                                                                                    \
        g_attn_trace[wg_ * 16 + 2 * (slot)] = wall_clock64();                                              \
        g_attn_trace[wg_ * 16 + 2 * (slot) + 1] = clock64();                                               \
      }                                                                                                    \
    }                                                                                                      \
  } while (0)
#define MI355_WAIT_ALL() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define MI355_STAMP(slot)
#define MI355_WAIT_ALL()
#endif

constexpr int kDecodeThreads = 256;
constexpr int kStep = 32;  // tokens per wave step (two 16-token pages)
// Register sets of K/V tiles per wave: the loads of a wave's next step are in flight while it
// computes one (measured at contexts 129-255: 13.6 vs 15.2 us; three sets were slower again).
constexpr int kMaxStages = 2;

template <int KS, int NT>
struct KVRegs {
  uint4 ka[KS], kb[KS], vf[NT];
};

struct RopeArgs {             // FUSE != 0
  const float* partial;       // FUSE == 2: [num_slabs][num_seqs][width] fp32
  int num_slabs;
  int64_t slab_stride;
  const uint16_t* bias;       // FUSE == 2: [width] or null
  const uint16_t* cos_sin;    // [max_position][D]: cos | sin halves
  int max_position;
  const int64_t* slots;       // [num_seqs]; < 0 or beyond the pool: not stored
};

// 8 consecutive elements of the projection row of `seq` starting at column `col`: the loads
// (issue) are separate from the arithmetic (finish) so that a caller can put every load of a
// prologue in flight before the first value is used.  SLABS > 0: the slab count is a
// compile-time constant (straight-line loads); SLABS == 0: any count, loaded in finish().
template <typename T, int FUSE, int SLABS>
struct Proj8 {
  static constexpr int S = SLABS > 0 ? SLABS : 1;
  float4 a[S], b[S];
  uint4 raw;      // FUSE == 1: the 16-bit row; FUSE == 2: the bias (or a dummy)
  float has;
  const float* pr;

  __device__ __forceinline__ void issue(const uint16_t* q, int64_t q_stride, const RopeArgs& ra, int seq, int width,
                                        int col) {
    if (FUSE == 2) {
      pr = ra.partial + static_cast<int64_t>(seq) * width + col;
      if (SLABS > 0) {
#pragma unroll
        for (int s = 0; s < SLABS; ++s) {
          const float* p = pr + s * ra.slab_stride;
          a[s] = MI355_LD(float4, ra.partial, p);
          b[s] = MI355_LD(float4, ra.partial, p + 4);
        }
      }
      // branch-free (a branch around a load drains the load queue): without a bias the load
      // goes to the start of the cos/sin table and is multiplied by zero
      has = ra.bias != nullptr ? 1.f : 0.f;
      raw = MI355_LD(uint4, ra.bias != nullptr ? ra.bias : ra.cos_sin, ra.bias != nullptr ? ra.bias + col : ra.cos_sin);
    } else {
      raw = MI355_LD(uint4, q, q + static_cast<int64_t>(seq) * q_stride + col);
    }
  }
  __device__ __forceinline__ void finish(float (&f)[8], const RopeArgs& ra) const {
    if (FUSE == 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = 0.f;
      if (SLABS > 0) {
#pragma unroll
        for (int s = 0; s < SLABS; ++s) {
          f[0] += a[s].x; f[1] += a[s].y; f[2] += a[s].z; f[3] += a[s].w;
          f[4] += b[s].x; f[5] += b[s].y; f[6] += b[s].z; f[7] += b[s].w;
        }
      } else {
        for (int s = 0; s < ra.num_slabs; ++s) {
          const float* p = pr + s * ra.slab_stride;
          const float4 x = MI355_LD(float4, ra.partial, p), y = MI355_LD(float4, ra.partial, p + 4);
          f[0] += x.x; f[1] += x.y; f[2] += x.z; f[3] += x.w;
          f[4] += y.x; f[5] += y.y; f[6] += y.z; f[7] += y.w;
        }
      }
      float bb[8];
      unpack8<T>(raw, bb);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = round_f<T>(pin_f32(f[j] + has * bb[j]));   // fp32 sum first, then one rounding
    } else {
      unpack8<T>(raw, f);
    }
  }
};

// rotate-half rotation of 8 pairs; every product and the sum rounded to the 16-bit type
template <typename T>
__device__ __forceinline__ void rotate8(const float (&x1)[8], const float (&x2)[8], const float (&co)[8],
                                        const float (&si)[8], float (&o1)[8], float (&o2)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // contraction off: with fp16 the compiler would lower to half arithmetic and fuse mul + sub into one fma
#pragma clang fp contract(off)
    o1[j] = round_f<T>(x1[j] * co[j]) - round_f<T>(x2[j] * si[j]);
    o2[j] = round_f<T>(x2[j] * co[j]) + round_f<T>(x1[j] * si[j]);
  }
}

template <typename T, int D, int FUSE, int SLABS, bool SPLIT, int kStages, int NW>
__global__ __launch_bounds__(64 * NW) void paged_attn_decode_kernel(
    uint16_t* __restrict__ out, float* __restrict__ part_o, float* __restrict__ part_ml,
    const uint16_t* __restrict__ q, int64_t q_stride, uint16_t* k_cache,
    uint16_t* v_cache, int num_blocks, const int32_t* __restrict__ block_tables,
    int max_blocks, const int32_t* __restrict__ context_lens, int num_heads, int num_kv_heads,
    float scale_log2e, int num_splits, RopeArgs ra) {
  constexpr int KS = D / 32;  // MFMA k-steps over head_dim for QK^T
  constexpr int NT = D / 16;  // 16-row tiles of head_dim for PV
  constexpr int BS = MI355_BLOCK_SIZE;

  MI355_STAMP(0);
  const int seq = blockIdx.x, kvh = blockIdx.y, split = blockIdx.z;
  const int group = num_heads / num_kv_heads;
  const int lane = threadIdx.x & 63;
  // wave-uniform by construction; saying so keeps the block-table lookups of the main loop on the
  // scalar unit (a vector lookup would have to wait for vmcnt(0), i.e. for every K/V load in flight)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, h = lane >> 4;

  // Fused forms: the projection row of this (sequence, kv head) -- the group's q heads, the k head, the
  // v head -- is formed ONCE per workgroup, by as many threads as it has 8-element chunks, with
  // coalesced loads, and handed to the waves through LDS.  (Every wave loading and rotating its own
  // copy straight into the MFMA operand layout cost ~70 load instructions per wave, most lanes
  // fetching duplicates: 6 us of a 13 us kernel.)
  //   rotating rows (q heads, then k): thread (row, j) holds the pair chunks [8j, +8) and
  //   [D/2 + 8j, +8) of its row and rotates them; the v row: 8 elements per thread, no rotation.
  // The row's loads go out before the K/V page loads of the wave's first stages, so that the wait
  // for the row leaves those in flight (loads return in order).  Once the page loads of 256
  // workgroups are queued the row still takes ~4 us to arrive; measured alternatives: the row's
  // loads at the kernel's first instruction (no sooner, and the context length later), page loads
  // only after the row is in LDS (+0.8 us), only the first stage before it (+0.1 us).
  constexpr int CH = D / 16;                         // chunks of 8 rotation pairs per row
  const int width = (num_heads + 2 * num_kv_heads) * D;
  const int kcol = (num_heads + kvh) * D, vcol = (num_heads + num_kv_heads + kvh) * D;
  const int rot_items = (group + 1) * CH;            // <= 17 * 8
  const bool rot = FUSE != 0 && static_cast<int>(threadIdx.x) < rot_items;
  const bool isv = FUSE != 0 && !rot && static_cast<int>(threadIdx.x) < rot_items + D / 8;
  const int row = rot ? static_cast<int>(threadIdx.x) / CH : group + 1;
  const int j8 = 8 * (rot ? static_cast<int>(threadIdx.x) % CH : static_cast<int>(threadIdx.x) - rot_items);
  Proj8<T, FUSE, SLABS> p1, p2;

  // Without splits the block-table entries of a wave's first step do not depend on the context
  // length: their (vector) loads go out before the (scalar) context-length load, so the two
  // round trips overlap instead of the scalar wait holding the vector loads back.
  const int32_t* bt = block_tables + static_cast<int64_t>(seq) * max_blocks;
  int e0 = 0, e1 = 0;
  if (!SPLIT) {   // compile-time: a run-time branch around the loads would drain them at its join
    const int bi = min(2 * wave, max_blocks - 1);
    e0 = MI355_LD(int32_t, block_tables, bt + bi);
    e1 = MI355_LD(int32_t, block_tables, bt + min(bi + 1, max_blocks - 1));
  }
  __builtin_amdgcn_sched_barrier(0);
  int ctx = MI355_LD(int32_t, context_lens, context_lens + seq);
  const int ctx_cap = max_blocks * BS;
  ctx = ctx < 0 ? 0 : (ctx > ctx_cap ? ctx_cap : ctx);
#ifdef MI355_ATTN_TRACE
  if (ctx >= 0) MI355_STAMP(1);
#endif
  // token range of this split: multiples of one workgroup pass (NW waves x 32)
  int per_split = (ctx + num_splits - 1) / num_splits;
  per_split = (per_split + kStep * NW - 1) / (kStep * NW) * (kStep * NW);
  const int t_begin = split * per_split;
  const int t_end = min(ctx, t_begin + per_split);

  const int last_blk_idx = ctx > 0 ? (ctx - 1) / BS : 0;
  const int koff = ((c >> 2) & 1) * 8 + (c & 3);  // token row inside the page for tile "a"

  // ---- loads of one 32-token step: K tiles a/b, then the V tiles.  kStages register sets: a wave
  // keeps the loads of its next kStages - 1 steps in flight while it computes one.
  KVRegs<KS, NT> kv[kStages];
  // `live` (wave-uniform, 0 / 1): whether the step holds tokens of this wave's range.  The loads of a
  // step beyond it are still issued (see issue_loads) but with the lane-dependent part of the address
  // zeroed: all lanes then ask for the same 16 bytes and the instruction is ONE request to the cache
  // instead of 16 -- at contexts around 160, two thirds of a launch's load instructions are such
  // steps, and they queue in front of the loads that matter.
  auto issue_page_loads = [&](KVRegs<KS, NT>& r, int blk0, int blk1, int live) {
    blk0 = min(max(blk0, 0), num_blocks - 1);
    blk1 = min(max(blk1, 0), num_blocks - 1);
    const int kblk = (c & 8) ? blk1 : blk0;
    const uint16_t* kp =
        k_cache + (static_cast<int64_t>(kblk) * num_kv_heads + kvh) * (BS * D) + live * (koff * D + 8 * h);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      r.ka[ks] = MI355_LD(uint4, k_cache, kp + 32 * ks);
      r.kb[ks] = MI355_LD(uint4, k_cache, kp + 4 * D + 32 * ks);
    }
    const int vblk = (h & 2) ? blk1 : blk0;
    const uint16_t* vp =
        v_cache + (static_cast<int64_t>(vblk) * num_kv_heads + kvh) * (D * BS) + live * (c * BS + 8 * (h & 1));
#pragma unroll
    for (int n = 0; n < NT; ++n) r.vf[n] = MI355_LD(uint4, v_cache, vp + n * 16 * BS);
  };
  // Always issued, never inside a branch (a load inside a branch makes the compiler drain the load
  // queue at the join): a step beyond the wave's range reads the context's last pages, which the
  // cache holds, and is never used.
  auto issue_loads = [&](KVRegs<KS, NT>& r, int tb) {
    const int bi = min(tb / BS, last_blk_idx);
    issue_page_loads(r, MI355_LD(int32_t, block_tables, bt + bi),
                     MI355_LD(int32_t, block_tables, bt + min(bi + 1, last_blk_idx)), tb < t_end ? 1 : 0);
  };
  int tb = t_begin + kStep * wave;
  // the K/V loads of the wave's first kStages steps; in the fused forms they go out AFTER the loads of
  // the prologue, so that the wait for the prologue's values leaves them in flight (loads return in order)
  auto issue_first_stages = [&]() {
    if (!SPLIT) {
      issue_page_loads(kv[0], e0, e1, tb < t_end ? 1 : 0);
    } else {
      issue_loads(kv[0], tb);
    }
#pragma unroll
    for (int j = 1; j < kStages; ++j) issue_loads(kv[j], tb + j * kStep * NW);
  };

  // Q fragments (B operand): column c = head within the group
  uint4 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = make_uint4(0, 0, 0, 0);
  // fused form: the new token's rotated k row in fragment order, and v[16n + c]
  uint4 knew[FUSE ? KS : 1];
  uint16_t vnew[FUSE ? NT : 1];
  const int t_new = FUSE ? ctx - 1 : -1;       // patched into the step that covers it
  if (FUSE == 0) {
    if (c < group) {
      const uint16_t* qp = q + seq * q_stride + static_cast<int64_t>(kvh * group + c) * D + 8 * h;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) qf[ks] = MI355_LD(uint4, q, qp + 32 * ks);
    }
    issue_first_stages();
  } else {
    // rows: q heads (<= 16) | k | v, each padded by 16 bytes: the 8 rows a wave writes at once start 4 banks apart
    constexpr int DR = D + 8;
    __shared__ __attribute__((aligned(16))) uint16_t lds_new[(16 + 2) * DR];
    // the new token sits at position ctx - 1: no separate position array, no extra dependent load
    int64_t pos = ctx - 1;
    pos = pos < 0 ? 0 : (pos >= ra.max_position ? ra.max_position - 1 : pos);
    const uint16_t* cs = ra.cos_sin + pos * D;
    const bool owner = ctx > 0 && t_new >= t_begin && t_new < t_begin + per_split;
    const int64_t slot = MI355_LD(int64_t, ra.slots, ra.slots + seq);
    uint4 cosv = make_uint4(0, 0, 0, 0), sinv = make_uint4(0, 0, 0, 0);
    if (rot) {
      const int base = row < group ? (kvh * group + row) * D : kcol;
      p1.issue(q, q_stride, ra, seq, width, base + j8);
      p2.issue(q, q_stride, ra, seq, width, base + D / 2 + j8);
      cosv = MI355_LD(uint4, ra.cos_sin, cs + j8);            // the same rows for every layer of a step: L2 hits
      sinv = MI355_LD(uint4, ra.cos_sin, cs + D / 2 + j8);
    } else if (isv) {
      p1.issue(q, q_stride, ra, seq, width, vcol + j8);
    }
    __builtin_amdgcn_sched_barrier(0);
    issue_first_stages();
    __builtin_amdgcn_sched_barrier(0);
#ifdef MI355_ATTN_TRACE
    if (rot && p1.a[0].x == 12345.678f) MI355_STAMP(7);
    if (rot) MI355_STAMP(7);                       // after the first slab vector of thread 0 has arrived
#endif
    if (rot) {
      float co[8], si[8], x1[8], x2[8], o1[8], o2[8];
      unpack8<T>(cosv, co);
      unpack8<T>(sinv, si);
      p1.finish(x1, ra);
      p2.finish(x2, ra);
      rotate8<T>(x1, x2, co, si, o1, o2);
      *reinterpret_cast<uint4*>(&lds_new[row * DR + j8]) = pack8<T>(o1);
      *reinterpret_cast<uint4*>(&lds_new[row * DR + D / 2 + j8]) = pack8<T>(o2);
    } else if (isv) {
      float f[8];
      p1.finish(f, ra);
      *reinterpret_cast<uint4*>(&lds_new[(group + 1) * DR + j8]) = pack8<T>(f);
    }
    __syncthreads();
    if (c < group) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const uint4*>(&lds_new[c * DR + 32 * ks + 8 * h]);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) knew[ks] = *reinterpret_cast<const uint4*>(&lds_new[group * DR + 32 * ks + 8 * h]);
#pragma unroll
    for (int n = 0; n < NT; ++n) vnew[n] = lds_new[(group + 1) * DR + 16 * n + c];
    if (owner) {
      // the pages get the new token too (for the steps to come); nobody in this launch waits for it
      if (slot >= 0 && slot < static_cast<int64_t>(num_blocks) * BS && wave == 0) {
        const int64_t blk = slot / BS;
        const int off = static_cast<int>(slot % BS);
        if (c == 0) {                                // lanes h = 0..3 hold the whole k row between them
          uint16_t* dst = k_cache + ((blk * num_kv_heads + kvh) * BS + off) * D + 8 * h;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) MI355_ST(uint4, k_cache, dst + 32 * ks, knew[ks]);
        }
        if (h == 0) {                                // lanes c = 0..15 hold v[16n + c]
          uint16_t* dst = v_cache + (blk * num_kv_heads + kvh) * D * BS + off;
#pragma unroll
          for (int n = 0; n < NT; ++n) MI355_ST(uint16_t, v_cache, dst + static_cast<int64_t>(16 * n + c) * BS, vnew[n]);
        }
      }
    }
  }

#ifdef MI355_ATTN_TRACE
  if (qf[0].x != 0x12345678u) MI355_STAMP(2);        // depends on the rotated query: after the prologue's loads
#endif
  float m = -INFINITY, l = 0.f;
  f32x4 o[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto step = [&](KVRegs<KS, NT>& kvr, int tb) {
    if (FUSE != 0 && t_new >= tb && t_new < tb + kStep) {
      // the new token's k row / v column come from registers, whatever the pages held
      const int r = t_new - tb;
      const int ta = 16 * (c >> 3) + koff;           // token of this lane's row in tile a; tile b: + 4
      if (r == ta) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kvr.ka[ks] = knew[ks];
      }
      if (r == ta + 4) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kvr.kb[ks] = knew[ks];
      }
      if ((r >> 3) == h) {                           // this lane's 8 tokens are 8h .. 8h + 7
        const int j = r & 7;
        const uint32_t keep = (j & 1) ? 0x0000FFFFu : 0xFFFF0000u;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const uint32_t ins = static_cast<uint32_t>(vnew[n]) << ((j & 1) * 16);
          uint32_t w[4] = {kvr.vf[n].x, kvr.vf[n].y, kvr.vf[n].z, kvr.vf[n].w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e == (j >> 1)) w[e] = (w[e] & keep) | ins;
          kvr.vf[n] = make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
    }

    // ---- S^T = K . Q^T
    f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      sa = T::mfma(kvr.ka[ks], qf[ks], sa);
      sb = T::mfma(kvr.kb[ks], qf[ks], sb);
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
        kvr.vf[n].x &= msk[0];
        kvr.vf[n].y &= msk[1];
        kvr.vf[n].z &= msk[2];
        kvr.vf[n].w &= msk[3];
      }
    }
    // ---- O^T += V^T . P^T
#pragma unroll
    for (int n = 0; n < NT; ++n) o[n] = T::mfma(kvr.vf[n], pf, o[n]);

  };
  __builtin_amdgcn_sched_barrier(0);
#ifdef MI355_ATTN_TRACE
  MI355_WAIT_ALL();
  MI355_STAMP(3);                                     // every K/V load issued so far has arrived
#endif
  if (t_begin + kStages * kStep * NW >= t_end) {
    // Every step of the range is already in a first stage of some wave (contexts up to 256 tokens without
    // splits): nothing more to load.  Workgroup-uniform, and this side holds no load at all: the look-ups of
    // the block table (a scalar round trip each) and the one-address loads the general loop issues for steps
    // beyond the range cost a wave ~0.2 us a stage here.
#pragma unroll
    for (int j = 0; j < kStages; ++j) {
      if (tb < t_end) step(kv[j], tb);
      tb += kStep * NW;
    }
  } else {
    while (tb < t_end) {
#pragma unroll
      for (int j = 0; j < kStages; ++j) {
        if (tb < t_end) {
          step(kv[j], tb);
        }
        issue_loads(kv[j], tb + kStages * kStep * NW);
        tb += kStep * NW;
      }
    }
  }

#ifdef MI355_ATTN_TRACE
  if (l >= 0.f) MI355_STAMP(4);
#endif
  // column sums live on the 4 lanes (c, h = 0..3)
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);

  // ---- merge the waves through LDS: O as [wave][16 heads][d], a head's row padded by 4 words.  A lane's four
  // accumulator words of a tile are four consecutive d of ONE head: one 16-byte store, and the heads' rows start
  // 4 banks apart, so the lanes served together cover all banks once; the readers below take two adjacent d per
  // thread, consecutive threads consecutive pairs.  (As [d][16 heads] every lane of a reading wave asked the
  // same bank: +0.6 us a launch.)
  constexpr int DP = D + 4;
  __shared__ __attribute__((aligned(16))) float lds_o[NW][16 * DP];
  __shared__ float lds_m[NW][16];
  __shared__ float lds_l[NW][16];
#pragma unroll
  for (int n = 0; n < NT; ++n) *reinterpret_cast<f32x4*>(&lds_o[wave][c * DP + 16 * n + 4 * h]) = o[n];
  if (h == 0) {
    lds_m[wave][c] = m;
    lds_l[wave][c] = l;
  }
  __syncthreads();
  MI355_STAMP(5);

  // two adjacent head_dim elements per thread: one pass for a group of 4 heads of 128
  for (int idx = threadIdx.x; idx < group * (D / 2); idx += 64 * NW) {
    const int hh = idx / (D / 2), d = 2 * (idx % (D / 2));
    float mg = lds_m[0][hh];
#pragma unroll
    for (int w = 1; w < NW; ++w) mg = fmaxf(mg, lds_m[w][hh]);
    float acc0 = 0.f, acc1 = 0.f, lsum = 0.f;
    if (mg != -INFINITY) {
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float f = exp2f(lds_m[w][hh] - mg);
        const float2 ov = *reinterpret_cast<const float2*>(&lds_o[w][hh * DP + d]);
        acc0 += f * ov.x;
        acc1 += f * ov.y;
        lsum += f * lds_l[w][hh];
      }
    }
    const int head = kvh * group + hh;
    if (num_splits == 1) {
      const float v0 = lsum > 0.f ? acc0 / lsum : 0.f, v1 = lsum > 0.f ? acc1 / lsum : 0.f;
      MI355_ST(uint32_t, out, out + (static_cast<int64_t>(seq) * num_heads + head) * D + d,
               static_cast<uint32_t>(T::from_f(v0)) | (static_cast<uint32_t>(T::from_f(v1)) << 16));
    } else {
      // the partials are one buffer: [rows][splits][D] accumulators, then [rows][splits][2] statistics
      const int64_t pi = (static_cast<int64_t>(seq) * num_heads + head) * num_splits + split;
      MI355_ST(float2, part_o, part_o + pi * D + d, make_float2(acc0, acc1));
      if (d == 0) {
        MI355_ST(float, part_o, part_ml + pi * 2, mg);
        MI355_ST(float, part_o, part_ml + pi * 2 + 1, lsum);
      }
    }
  }
  MI355_STAMP(6);
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
  for (int s = 0; s < num_splits; ++s) mg = fmaxf(mg, MI355_LD(float, part_o, ml + 2 * s));
  float acc = 0.f, lsum = 0.f;
  if (mg != -INFINITY) {
    for (int s = 0; s < num_splits; ++s) {
      const float f = exp2f(MI355_LD(float, part_o, ml + 2 * s) - mg);
      acc += f * MI355_LD(float, part_o, part_o + (row * num_splits + s) * D + d);
      lsum += f * MI355_LD(float, part_o, ml + 2 * s + 1);
    }
  }
  MI355_ST(uint16_t, out, out + row * D + d, T::from_f(lsum > 0.f ? acc / lsum : 0.f));
}

template <typename T, int D, int FUSE, int SLABS = 0>
static int32_t launch_decode(void* out, const void* q, int64_t q_stride, void* k_cache,
                             void* v_cache, int32_t num_blocks, const int32_t* block_tables,
                             int32_t max_blocks, const int32_t* context_lens, int32_t num_seqs,
                             int32_t num_heads, int32_t num_kv_heads, float scale, int32_t num_splits,
                             void* workspace, hipStream_t s, const RopeArgs& ra) {
  float* part_o = nullptr;
  float* part_ml = nullptr;
  if (num_splits > 1) {
    part_o = static_cast<float*>(workspace);
    part_ml = part_o + static_cast<int64_t>(num_seqs) * num_heads * num_splits * D;
  }
  // Waves per (sequence, kv head, split) and register stages of K/V per wave.  Two stages (the next step's
  // loads in flight while one is computed) cost the registers that would let a second workgroup share the CU:
  // worth it while the launch has at most one workgroup per CU and each must hide its own latencies; with more
  // workgroups than CUs (128 sequences x 8 kv heads: four rounds) one stage and two workgroups per CU is
  // faster -- 30.9 vs 37.7 us at 128 x 192 tokens, 58.4 vs 70.9 us at 256 x 192.  From 1024 groups on there
  // is a wave for every SIMD without splitting a group's tokens over four: ONE wave per group walks all of
  // its steps, nothing is merged across waves and eight groups share a CU -- 29.6 vs 31.0 us at 128 x 192
  // tokens in the step, 54.2 vs 58.5 us at 256 x 192 (at 512 groups it is the slower form: 18.9 vs 17.2 us).
  // MI355_ATTN_FORM = 42 / 41 / 11 (waves * 10 + stages) forces one form, for measurements.
  const int64_t wgs = static_cast<int64_t>(num_seqs) * num_kv_heads * num_splits;
  const bool one_wave_fits = (num_heads / num_kv_heads + 1) * (D / 16) + D / 8 <= 64;   // the prologue's threads
  int form = wgs >= 1024 && one_wave_fits ? 11 : wgs > 256 ? 41 : 42;
  static const int forced = [] { const char* e = getenv("MI355_ATTN_FORM"); return e ? atoi(e) : 0; }();
  if (forced == 41 || forced == 42 || (forced == 11 && one_wave_fits)) form = forced;
  dim3 grid(num_seqs, num_kv_heads, num_splits), block(form / 10 == 1 ? 64 : kDecodeThreads);
#define MI355_PAD_LAUNCH(SPLIT_, STG_, NW_)                                                                        \
  hipLaunchKernelGGL((paged_attn_decode_kernel<T, D, FUSE, SLABS, SPLIT_, STG_, NW_>), grid, block, 0, s,          \
                     static_cast<uint16_t*>(out), part_o, part_ml, static_cast<const uint16_t*>(q), q_stride,      \
                     static_cast<uint16_t*>(k_cache), static_cast<uint16_t*>(v_cache), num_blocks, block_tables,   \
                     max_blocks, context_lens, num_heads, num_kv_heads, scale * 1.4426950408889634f, num_splits, ra)
#define MI355_PAD_FORMS(SPLIT_)                                     \
  switch (form) {                                                   \
    case 41: MI355_PAD_LAUNCH(SPLIT_, 1, 4); break;                 \
    case 11: MI355_PAD_LAUNCH(SPLIT_, 1, 1); break;                 \
    default: MI355_PAD_LAUNCH(SPLIT_, kMaxStages, 4); break;        \
  }
  if (num_splits == 1) {
    MI355_PAD_FORMS(false)
  } else {
    MI355_PAD_FORMS(true)
  }
#undef MI355_PAD_FORMS
#undef MI355_PAD_LAUNCH
  int32_t rc = check_launch("mi355_paged_attn_decode");
  if (rc != MI355_OK || num_splits == 1) return rc;
  hipLaunchKernelGGL((paged_attn_merge_kernel<T, D>), dim3(num_seqs * num_heads), dim3(D), 0, s,
                     static_cast<uint16_t*>(out), part_o, part_ml, num_splits);
  return check_launch("mi355_paged_attn_decode(merge)");
}

}  // namespace mi355

#ifdef MI355_ATTN_TRACE
extern "C" int32_t mi355_attn_trace_read(unsigned long long* host, int32_t words) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(mi355::g_attn_trace), sizeof(unsigned long long) * words) == hipSuccess ? 0 : -3;
}
#endif

extern "C" int64_t mi355_paged_attn_decode_workspace(int32_t num_seqs, int32_t num_heads,
                                                     int32_t head_dim, int32_t num_splits) {
  if (num_splits <= 1 || num_seqs <= 0 || num_heads <= 0 || head_dim <= 0) return 0;
  return static_cast<int64_t>(num_seqs) * num_heads * num_splits * (head_dim + 2) * sizeof(float);
}

namespace mi355 {

static int32_t decode_entry(const char* name, void* out, const void* q, int64_t q_stride, void* k_cache,
                            void* v_cache, int32_t num_blocks, const int32_t* block_tables,
                            int32_t max_blocks_per_seq, const int32_t* context_lens, int32_t num_seqs,
                            int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, float scale,
                            int32_t num_splits, void* workspace, int64_t workspace_bytes, int32_t dtype,
                            void* stream, int fuse, const RopeArgs& ra) {
  MI355_REQUIRE(out && k_cache && v_cache && block_tables && context_lens && (q || fuse == 2), MI355_ERR_INVALID,
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
  const int64_t width = static_cast<int64_t>(num_heads + (fuse ? 2 * num_kv_heads : 0)) * head_dim;
  if (fuse != 2)
    MI355_REQUIRE(q_stride >= width && q_stride % 8 == 0 && aligned16(q), MI355_ERR_INVALID,
                  "%s: q_stride %lld invalid or q misaligned", name, (long long)q_stride);
  MI355_REQUIRE(aligned16(out) && aligned16(k_cache) && aligned16(v_cache), MI355_ERR_INVALID,
                "%s: pointers must be 16-byte aligned", name);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "%s: dtype %d not supported (bf16/fp16 only)", name, dtype);
  if (fuse) {
    MI355_REQUIRE(ra.cos_sin && ra.slots && ra.max_position > 0 && aligned16(ra.cos_sin), MI355_ERR_INVALID,
                  "%s: cos_sin / slot_mapping missing or misaligned", name);
    if (fuse == 2) {
      MI355_REQUIRE(ra.partial && aligned16(ra.partial) && (ra.bias == nullptr || aligned16(ra.bias)),
                    MI355_ERR_INVALID, "%s: partial missing or partial/bias misaligned", name);
      MI355_REQUIRE(ra.num_slabs >= 1 && ra.num_slabs <= 64 && ra.slab_stride >= num_seqs * width &&
                        ra.slab_stride % 4 == 0,
                    MI355_ERR_INVALID, "%s: num_slabs in [1, 64] and slab_stride >= num_seqs * width (multiple of 4) required",
                    name);
    }
  }
  if (num_splits > 1) {
    const int64_t need = mi355_paged_attn_decode_workspace(num_seqs, num_heads, head_dim, num_splits);
    MI355_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= need, MI355_ERR_INVALID,
                  "%s: workspace of %lld bytes required for %d splits (got %lld)", name, (long long)need,
                  num_splits, (long long)workspace_bytes);
  }
  if (num_seqs == 0) return MI355_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define MI355_DEC(TT, DD, FF, SS)                                                                          \
  return launch_decode<TT, DD, FF, SS>(out, q, q_stride, k_cache, v_cache, num_blocks, block_tables,      \
                                       max_blocks_per_seq, context_lens, num_seqs, num_heads, num_kv_heads, \
                                       scale, num_splits, workspace, s, ra)
#define MI355_DEC_F(TT, DD)                                 \
  do {                                                      \
    if (fuse == 0) MI355_DEC(TT, DD, 0, 0);                 \
    if (fuse == 1) MI355_DEC(TT, DD, 1, 0);                 \
    if (ra.num_slabs == 1) MI355_DEC(TT, DD, 2, 1);         \
    if (ra.num_slabs == 2) MI355_DEC(TT, DD, 2, 2);         \
    if (ra.num_slabs == 4) MI355_DEC(TT, DD, 2, 4);         \
    MI355_DEC(TT, DD, 2, 0);                                \
  } while (0)
  if (dtype == MI355_BF16) {
    if (head_dim == 64) MI355_DEC_F(BF16, 64);
    MI355_DEC_F(BF16, 128);
  }
  if (head_dim == 64) MI355_DEC_F(F16, 64);
  MI355_DEC_F(F16, 128);
#undef MI355_DEC_F
#undef MI355_DEC
}

}  // namespace mi355

extern "C" int32_t mi355_paged_attn_decode(void* out, const void* q, int64_t q_stride, const void* k_cache,
                                           const void* v_cache, int32_t num_blocks,
                                           const int32_t* block_tables, int32_t max_blocks_per_seq,
                                           const int32_t* context_lens, int32_t num_seqs,
                                           int32_t num_heads, int32_t num_kv_heads, int32_t head_dim,
                                           float scale, int32_t num_splits, void* workspace,
                                           int64_t workspace_bytes, int32_t dtype, void* stream) {
  // the caches are only read in this form
  return mi355::decode_entry("mi355_paged_attn_decode", out, q, q_stride, const_cast<void*>(k_cache),
                             const_cast<void*>(v_cache), num_blocks, block_tables, max_blocks_per_seq, context_lens,
                             num_seqs, num_heads, num_kv_heads, head_dim, scale, num_splits, workspace,
                             workspace_bytes, dtype, stream, 0, mi355::RopeArgs{});
}

// ---- the entry point that allocates nothing.  The split-KV partials (fp32 accumulators and
// softmax statistics per (sequence, head, split), the only device memory the decode kernels use
// beyond their inputs and outputs) live in a file-scope device array instead of a caller
// workspace.  In host code the array's NAME is only a handle: the address the kernels get is
// looked up once per device with hipGetSymbolAddress (every GPU holds its own copy of the array
// at its own address) and kept.
namespace mi355 {

constexpr int64_t kStaticScratchBytes = MI355_PAGED_ATTN_DECODE_STATIC_MAX_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

constexpr int kMaxDevices = 64;
static void* g_scratch_ptr[kMaxDevices];          // per device, written once under the lock

static int32_t static_scratch(const char* name, void** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  MI355_REQUIRE(e == hipSuccess && dev >= 0 && dev < kMaxDevices, MI355_ERR_INVALID, "%s: no current device (%s)",
                name, hipGetErrorString(e));
  void* p = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
  if (p == nullptr) {
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    MI355_REQUIRE(e == hipSuccess && p != nullptr, MI355_ERR_LAUNCH,
                  "%s: hipGetSymbolAddress of the scratch array failed on device %d: %s", name, dev,
                  hipGetErrorString(e));
    __atomic_store_n(&g_scratch_ptr[dev], p, __ATOMIC_RELEASE);   // the same value from every thread
  }
  *out = p;
  return MI355_OK;
}

}  // namespace mi355

extern "C" int32_t mi355_paged_attn_decode_static(void* out, const void* q, int64_t q_stride, const void* k_cache,
                                                  const void* v_cache, int32_t num_blocks,
                                                  const int32_t* block_tables, int32_t max_blocks_per_seq,
                                                  const int32_t* context_lens, int32_t num_seqs, int32_t num_heads,
                                                  int32_t num_kv_heads, int32_t head_dim, float scale,
                                                  int32_t num_splits, int32_t dtype, void* stream) {
  const char* name = "mi355_paged_attn_decode_static";
  const int64_t need = mi355_paged_attn_decode_workspace(num_seqs, num_heads, head_dim, num_splits);
  MI355_REQUIRE(need <= mi355::kStaticScratchBytes, MI355_ERR_UNSUPPORTED,
                "%s: %d sequences x %d heads x %d splits need %lld bytes of partials, the built-in scratch holds %lld "
                "(use mi355_paged_attn_decode with a workspace)", name, num_seqs, num_heads, num_splits,
                (long long)need, (long long)mi355::kStaticScratchBytes);
  void* ws = nullptr;
  const int32_t rc = mi355::static_scratch(name, &ws);
  if (rc != MI355_OK) return rc;
  // every partial a launch reads was written by that launch (each workgroup stores all of its
  // entries, empty splits included): nothing has to start at zero
  return mi355::decode_entry(name, out, q, q_stride, const_cast<void*>(k_cache), const_cast<void*>(v_cache),
                             num_blocks, block_tables, max_blocks_per_seq, context_lens, num_seqs, num_heads,
                             num_kv_heads, head_dim, scale, num_splits, ws, mi355::kStaticScratchBytes, dtype, stream, 0,
                             mi355::RopeArgs{});
}

extern "C" int32_t mi355_paged_attn_decode_fused(void* out, const void* qkv, int64_t qkv_stride,
                                                 const float* partial, int32_t num_slabs, int64_t slab_stride,
                                                 const void* bias, const void* cos_sin,
                                                 int32_t max_position, const int64_t* slot_mapping, void* k_cache,
                                                 void* v_cache, int32_t num_blocks, const int32_t* block_tables,
                                                 int32_t max_blocks_per_seq, const int32_t* context_lens,
                                                 int32_t num_seqs, int32_t num_heads, int32_t num_kv_heads,
                                                 int32_t head_dim, float scale, int32_t num_splits, void* workspace,
                                                 int64_t workspace_bytes, int32_t dtype, void* stream) {
  const mi355::RopeArgs ra{partial, num_slabs, slab_stride, static_cast<const uint16_t*>(bias),
                           static_cast<const uint16_t*>(cos_sin), max_position, slot_mapping};
  return mi355::decode_entry("mi355_paged_attn_decode_fused", out, qkv, qkv_stride, k_cache, v_cache, num_blocks,
                             block_tables, max_blocks_per_seq, context_lens, num_seqs, num_heads, num_kv_heads,
                             head_dim, scale, num_splits, workspace, workspace_bytes, dtype, stream,
                             partial != nullptr ? 2 : 1, ra);
}
