// Prefill / chunked-prefill attention: causal attention of packed query chunks
// over keys and values that are already in the paged KV cache.
//
// Work decomposition.  One 256-thread workgroup (4 waves) per (query tile, kv head,
// sequence).  A query tile is QT = 64*NQ / group consecutive queries; its (query,
// head-in-group) pairs -- up to 64*NQ of them -- are laid on MFMA output COLUMNS, 16 per
// column tile, and wave w owns column tiles w*NQ .. w*NQ+NQ-1 (NQ = 1 or 2 by the chunk
// length).  The workgroup walks the keys its tile can see in 32-token steps.  The
// K and V pages of a step are STAGED THROUGH LDS ONCE PER WORKGROUP, double buffered: all 256
// threads copy the two pages (coalesced 16-byte pieces, whole 4 KB pages) into an LDS image
// in FRAGMENT ORDER -- piece (fragment, lane) at (fragment*64 + lane)*16 for V, the same with
// the slots of each 16-lane block XOR-swizzled for K (see KFRAG below) -- so every wave's
// `ds_read_b128` of an MFMA A operand and every staging `ds_write_b128` is conflict-free;
// two steps are staged per barrier interval while the waves compute on the other pair of
// images: one workgroup barrier per 64 tokens.  The query fragments (B
// operands) stay in registers.  Softmax in ONE pass over the keys with a LAZILY updated exponent
// reference: p = exp2(s*scale - m_ref), where m_ref is a column's maximum as of the last time
// it was brought up to date.  A step only checks, lane by lane and with no cross-lane traffic,
// whether any of its scores exceeds m_ref + 8; only then (a wave-uniform, rarely taken branch:
// the first step of a tile and the few steps in which a maximum jumps by more than 2^8) are the
// column maxima reduced across lanes and O and l rescaled.  In between p <= 2^8, harmless in the
// 16-bit P operand and in the fp32 sums, and the loop body is a straight line: scores, one fma +
// exp2 per score, PV.  (An online softmax that follows the true maximum moves it in most steps
// with 16 columns per tile, and every move rescales all of O; a two-pass form -- maxima first --
// pays the QK^T MFMAs, the K staging and the barriers twice: 1.5x slower than this one.)
//
// MFMA mapping is the decode kernel's (paged_attn_decode.hip): S^T = K . Q^T with the K
// rows as the A operand, and O^T += V^T . P^T with the transposed V page as the A operand
// and the lane's own probabilities as B.  The causal mask is applied per column (key <=
// position of the column's query).
#include "common.h"

#include <math.h>
#include <stdlib.h>

namespace mi355 {

constexpr int kPrefillThreads = 256;

typedef float f32x2 __attribute__((ext_vector_type(2)));

// max of three in one instruction (fmaxf chains go through a canonicalising v_max_f32 per input)
__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// Second launch bound: waves per SIMD the register allocation must leave room for.  Two workgroups
// per CU overlap each other's dependent chains (LDS reads -> MFMAs -> exp2 -> MFMAs -> barrier).
template <typename T, int D, int NQ>
#ifndef MI355_EXP_WG_PER_CU
#define MI355_EXP_WG_PER_CU 2
#endif
__global__ __launch_bounds__(kPrefillThreads, MI355_EXP_WG_PER_CU) void paged_attn_prefill_kernel(
    uint16_t* __restrict__ out, int64_t out_stride, const uint16_t* __restrict__ q, int64_t q_stride,
    const uint16_t* __restrict__ k_cache, const uint16_t* __restrict__ v_cache, int num_blocks,
    const int32_t* __restrict__ block_tables, int max_blocks, const int32_t* __restrict__ cu_seqlens,
    const int32_t* __restrict__ context_lens, int num_heads, int num_kv_heads, float scale_log2e, int qt) {
  constexpr int KS = D / 32;
  constexpr int NT = D / 16;
  constexpr int BS = MI355_BLOCK_SIZE;
  // K image of a step: [tile a/b][k-step][h = lane >> 4][c' = (lane & 15) ^ (2h + (k-step & 1))][16 B].
  // Reads (`ds_read_b128`: 16 lanes per LDS cycle, a bank per dword of 256 B) want the lanes of a
  // 16-lane block on consecutive 16-byte slots; the staging writes (`ds_write_b128`: 8 consecutive
  // threads per cycle, banks wrapping at 128 B) come as the eight (k-step, h) slots of one key row,
  // 256 B apart.  The XOR moves those eight onto different bank groups and only permutes the
  // slots of a block among the lanes that read it: both sides are conflict-free.
  constexpr int KFRAG = 1024;
  constexpr int KBYTES = 2 * KS * KFRAG;
  constexpr int VBYTES = NT * 1024;              // V image: [16-row tile of head_dim][lane][16 B]
  // LDS: K images [interval parity][image A/B], V images of the A half-steps [interval parity], V images of the B
  // half-steps in a ring of three (a B image is still read -- its PV product runs under the next interval's
  // first QK^T -- after the barrier that lets the interval after that be staged)
  constexpr int K_AT = 0, VA_AT = 4 * KBYTES, VB_AT = VA_AT + 2 * VBYTES, LDS_BYTES = VB_AT + 3 * VBYTES;
  constexpr int PIECES = 8 * D;                  // 16-byte pieces per step: two K pages + two V pages
  constexpr int PPT = PIECES / kPrefillThreads;  // per thread: 4 (D = 128) or 2 (D = 64)
  static_assert(PIECES % kPrefillThreads == 0, "whole pieces per thread");
  // HS 32-token images per barrier interval, two such sets (one computed on, one being staged):
  // half as many workgroup barriers per key as with one image per interval
  constexpr int HS = 2;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  // Workgroup -> (query tile, kv head, sequence).  The hardware deals consecutive workgroups to the
  // 8 XCDs in turn, and each XCD has its own L2: with the plain grid order every XCD would walk the
  // K/V pages of every sequence (16 x 2048 tokens: 16 MB through each 4 MB L2, the launch then runs
  // at the memory-side cache's rate).  Here XCD x takes a contiguous range of (sequence, kv head)
  // pairs instead, so the tiles that share pages share an L2; within a pair the LAST query tile
  // comes first -- it has the most keys to walk (causal), and the short ones fill the tail.
  const int gx = gridDim.x, gy = gridDim.y;
  const int total = gx * gy * static_cast<int>(gridDim.z);
  const int lin = blockIdx.x + gx * (blockIdx.y + gy * static_cast<int>(blockIdx.z));
  const int xcd = lin & 7;
  const int work = xcd * (total >> 3) + min(xcd, total & 7) + (lin >> 3);   // bijective on [0, total)
  const int qtile = gx - 1 - work % gx, kvh = (work / gx) % gy, seq = work / (gx * gy);
  const int group = num_heads / num_kv_heads;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, h = lane >> 4;

  const int q_begin = MI355_LD(int32_t, cu_seqlens, cu_seqlens + seq);
  const int q_len = MI355_LD(int32_t, cu_seqlens, cu_seqlens + seq + 1) - q_begin;
  const int q0 = qtile * qt;
  if (q0 >= q_len) return;                       // whole workgroup: no barrier is skipped by some threads only
  int ctx = MI355_LD(int32_t, context_lens, context_lens + seq);
  const int ctx_cap = max_blocks * BS;
  ctx = ctx > ctx_cap ? ctx_cap : ctx;
  const int pos0 = ctx - q_len;                  // absolute position of the chunk's first query
  if (pos0 < 0) return;                          // inconsistent lengths: nothing sensible to compute

  // this lane's columns: pair index p = (wave*NQ + t)*16 + c -> (query, head)
  int col_q[NQ], col_head[NQ], col_pos[NQ];
  bool col_ok[NQ];
  uint4 qf[NQ][KS];                              // the lane's query fragments (B operands), read once
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    const int p = (wave * NQ + t) * 16 + c;
    col_q[t] = q0 + p / group;
    col_head[t] = kvh * group + p % group;
    col_ok[t] = (p < qt * group) && (col_q[t] < q_len);
    col_pos[t] = col_ok[t] ? pos0 + col_q[t] : -1;
    const uint16_t* qp = q + (q_begin + (col_ok[t] ? col_q[t] : 0)) * q_stride +
                         static_cast<int64_t>(col_ok[t] ? col_head[t] : 0) * D + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const uint4 v = MI355_LD(uint4, q, qp + 32 * ks);     // always a valid row; dropped if not ours
      qf[t][ks] = col_ok[t] ? v : make_uint4(0, 0, 0, 0);
    }
  }
  int kmin = 0x7FFFFFFF;                                         // smallest position among the wave's valid columns
#pragma unroll
  for (int t = 0; t < NQ; ++t) kmin = min(kmin, col_ok[t] ? col_pos[t] : 0x7FFFFFFF);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) kmin = min(kmin, __shfl_xor(kmin, off, 64));
  const int t_min = __builtin_amdgcn_readfirstlane(kmin);
  const int t_end_wg = pos0 + min(q0 + qt, q_len);              // the tile's last query sees keys [0, t_end_wg)

  const int32_t* bt = block_tables + static_cast<int64_t>(seq) * max_blocks;
  const int last_blk_idx = (t_end_wg - 1) / BS;                 // t_end_wg >= 1

  // ---- staging: thread tid copies pieces g = j*256 + tid of a step (j < PPT/2: the two K pages,
  // the rest: the two V pages).  Source addresses walk a page front to back (coalesced);
  // destinations are in fragment order.
  // Where piece j of this thread comes from and goes: recomputed from the thread index at each use (a dozen
  // integer instructions per 64 tokens) instead of held in 14 registers through the loop -- the index goes
  // through an empty asm first so that the compiler cannot hoist the arithmetic back out
  struct Piece { int page, src, dst, tok; };
  auto piece = [&](int j, int tid_) {
    Piece pc;
    const int g = j * kPrefillThreads + tid_;
    if (j < PPT / 2) {                                           // K: [page][row 0..15][16-byte slot 0..D/8)
      const int pg = g / (2 * D), rem = g % (2 * D);
      const int r = rem / (D / 8), sl = rem % (D / 8);
      const int ks = sl >> 2, hh = sl & 3;
      const int tile = (r >> 2) & 1;
      const int cc = (r & 3) | (((r >> 3) & 1) << 2) | (pg << 3);
      pc.page = pg;
      pc.src = r * D + sl * 8;
      pc.tok = 0;
      pc.dst = (tile * KS + ks) * KFRAG + hh * 256 + (cc ^ (2 * hh + (ks & 1))) * 16;
    } else {                                                     // V: [page][head_dim row][half of the 16 tokens]
      const int g2 = g - PIECES / 2;
      const int pg = g2 / (2 * D), rem = g2 % (2 * D);
      const int dr = rem >> 1, half = rem & 1;
      pc.page = pg;
      pc.src = dr * BS + half * 8;
      pc.tok = pg * BS + half * 8;                               // token offset of the piece inside the step
      pc.dst = ((dr >> 4) * 64 + (pg * 2 + half) * 16 + (dr & 15)) * 16;                   // inside a V image
    }
    return pc;
  };
  typedef unsigned int pvec __attribute__((ext_vector_type(4)));
  auto stage_load = [&](pvec (&regs)[PPT], int tb) {
    constexpr int NP = PPT;
    const int bi = min(tb / BS, last_blk_idx);
    int blk[2] = {MI355_LD(int32_t, block_tables, bt + bi), MI355_LD(int32_t, block_tables, bt + min(bi + 1, last_blk_idx))};
    blk[0] = min(max(blk[0], 0), num_blocks - 1);
    blk[1] = min(max(blk[1], 0), num_blocks - 1);
    int tid_ = tid;
#ifndef MI355_EXP_HOIST_PIECES
    asm volatile("" : "+v"(tid_));
#endif
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const Piece pc = piece(j, tid_);
      const int64_t page = (static_cast<int64_t>(blk[pc.page]) * num_kv_heads + kvh) * (BS * D);
      const uint16_t* src = (j < PPT / 2 ? k_cache : v_cache) + page + pc.src;
      regs[j] = MI355_LD(pvec, j < PPT / 2 ? k_cache : v_cache, src);
    }
  };
  // In the image that reaches the tile's end the V columns of tokens beyond its last query may be
  // unwritten memory (their probabilities are 0, but 0 * NaN is NaN): cleared here, where no
  // accumulator is live across the branch, rather than in the MFMA loop.
  auto stage_store = [&](const pvec (&regs)[PPT], int k_at, int v_at, int tb) {   // byte offsets of the step's K and V image
    int tid_ = tid;
#ifndef MI355_EXP_HOIST_PIECES
    asm volatile("" : "+v"(tid_));
#endif
#pragma unroll
    for (int j = 0; j < PPT / 2; ++j) *reinterpret_cast<pvec*>(&lds[k_at + piece(j, tid_).dst]) = regs[j];
    if (tb + 32 > t_end_wg) {                                      // workgroup-uniform
#pragma unroll
      for (int j = PPT / 2; j < PPT; ++j) {
        const Piece pc = piece(j, tid_);
        const int tok = tb + pc.tok;                               // first of the piece's 8 tokens
        pvec v = regs[j];
#pragma unroll
        for (int w = 0; w < 4; ++w)
          v[w] &= ((tok + 2 * w < t_end_wg) ? 0x0000FFFFu : 0u) | ((tok + 2 * w + 1 < t_end_wg) ? 0xFFFF0000u : 0u);
        *reinterpret_cast<pvec*>(&lds[v_at + pc.dst]) = v;
      }
    } else {
#pragma unroll
      for (int j = PPT / 2; j < PPT; ++j) *reinterpret_cast<pvec*>(&lds[v_at + piece(j, tid_).dst]) = regs[j];
    }
  };

  const int steps = (t_end_wg + 31) / 32;                        // >= 1
  const int tok_h = 8 * h;

  // ---- softmax state: p = exp2(s * scale - m_ref) against a lazily raised reference
  float m_ref[NQ], m_thr[NQ], neg_m[NQ];          // in the exponent's units (score * scale * log2 e)
  f32x2 l[NQ];                                    // sum of the lane's probabilities, two partial sums
  f32x4 o[NQ][NT];
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    m_ref[t] = m_thr[t] = -INFINITY;
    neg_m[t] = 0.f;                               // a column that has seen no key yet: exponent reference 0
    l[t] = f32x2{0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NT; ++n) o[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  constexpr float kSlack = 8.f;                   // p <= 2^8 between two updates of the reference

  // The pieces of a 32-token step, as instruction streams that the main loop pairs up: a wave issues in
  // order, so a matrix product only runs under vector work that is INTERLEAVED with it in the program
  // (pair_sched below); with the products of a step following its exponentials back to back, as a plain
  // loop body has them, each waits for the other (two waves per SIMD: 30 % of the matrix pipe busy).
  struct Scores { f32x4 a[NQ], b[NQ]; };
  // (MFMA stream) S^T = K . Q^T for all column tiles; a K fragment is read from LDS once and used NQ times
  auto qk = [&](int k_at, Scores& S) {
    const unsigned char* base[2] = {&lds[k_at + h * 256 + (c ^ (2 * h)) * 16], &lds[k_at + h * 256 + (c ^ (2 * h + 1)) * 16]};
#pragma unroll
    for (int t = 0; t < NQ; ++t) S.a[t] = S.b[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const uint4 ka = *reinterpret_cast<const uint4*>(base[ks & 1] + ks * KFRAG);
      const uint4 kb = *reinterpret_cast<const uint4*>(base[ks & 1] + (KS + ks) * KFRAG);
#pragma unroll
      for (int t = 0; t < NQ; ++t) {
        S.a[t] = T::mfma(ka, qf[t][ks], S.a[t]);
        S.b[t] = T::mfma(kb, qf[t][ks], S.b[t]);
      }
    }
  };
  // causal mask, only where the step reaches the diagonal of some column of this wave (wave-uniform, rare)
  auto mask = [&](int s, Scores& S) {
    if (s * 32 + 32 > t_min + 1) {
      const int tok0 = s * 32 + tok_h;
#pragma unroll
      for (int t = 0; t < NQ; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          S.a[t][j] = (tok0 + j <= col_pos[t]) ? S.a[t][j] : -INFINITY;
          S.b[t][j] = (tok0 + 4 + j <= col_pos[t]) ? S.b[t][j] : -INFINITY;
        }
      }
    }
  };
  // (vector stream) the lane's largest score per column tile, in the exponent's units (scale > 0: the maximum
  // commutes with the scaling; a masked score is -inf)
  auto lane_max = [&](const Scores& S, float (&lm)[NQ]) {
#pragma unroll
    for (int t = 0; t < NQ; ++t)
      lm[t] = scale_log2e * max3(max3(max3(S.a[t][0], S.a[t][1], S.a[t][2]), S.a[t][3], S.b[t][0]),
                                 max3(S.b[t][1], S.b[t][2], S.b[t][3]), -INFINITY);
  };
  // does any score leave the slack above its column's reference?  Only then (wave-uniform; the first step of a
  // tile and the few in which a maximum jumps by more than 2^8) are the maxima reduced across lanes and O, l rescaled
  auto grow = [&](const float (&lm)[NQ]) {
    bool g = false;
#pragma unroll
    for (int t = 0; t < NQ; ++t) g |= lm[t] > m_thr[t];
    if (__any(g)) {
#pragma unroll
      for (int t = 0; t < NQ; ++t) {
        float v = lm[t];
        v = fmaxf(v, __shfl_xor(v, 16, 64));                     // the column's keys of this step sit on 4 lanes
        v = fmaxf(v, __shfl_xor(v, 32, 64));
        const float m_new = fmaxf(m_ref[t], v);
        // -inf -> finite: alpha = 0 (O and l are still zero); no key seen yet: nothing to rescale
        const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_ref[t] - m_new);
        l[t] *= alpha;                                           // both halves
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          o[t][n][0] *= alpha;
          o[t][n][1] *= alpha;
          o[t][n][2] *= alpha;
          o[t][n][3] *= alpha;
        }
        m_ref[t] = m_new;
        m_thr[t] = m_new + kSlack;                               // -inf stays -inf
        neg_m[t] = (m_new == -INFINITY) ? 0.f : -m_new;
      }
    }
  };
  // (vector stream) probabilities of a step's scores as PV operands; the softmax scale is folded into the
  // exponent, one fma per score
  auto probs = [&](const Scores& S, uint4 (&pf)[NQ]) {
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      const f32x2 sc = {scale_log2e, scale_log2e}, nm = {neg_m[t], neg_m[t]};
      const f32x2 e[4] = {__builtin_elementwise_fma(f32x2{S.a[t][0], S.a[t][1]}, sc, nm),   // v_pk_fma_f32
                          __builtin_elementwise_fma(f32x2{S.a[t][2], S.a[t][3]}, sc, nm),
                          __builtin_elementwise_fma(f32x2{S.b[t][0], S.b[t][1]}, sc, nm),
                          __builtin_elementwise_fma(f32x2{S.b[t][2], S.b[t][3]}, sc, nm)};
      float p[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        p[2 * j] = __builtin_amdgcn_exp2f(e[j][0]);
        p[2 * j + 1] = __builtin_amdgcn_exp2f(e[j][1]);
      }
      l[t] += (f32x2{p[0], p[1]} + f32x2{p[2], p[3]}) + (f32x2{p[4], p[5]} + f32x2{p[6], p[7]});   // v_pk_add_f32
      pf[t] = pack8<T>(p);
    }
  };
  // (MFMA stream) O^T += V^T . P^T: a V fragment is read once and used NQ times (the V columns of tokens beyond
  // the tile's last query were cleared when the image was staged)
  auto pv = [&](int v_at, const uint4 (&pf)[NQ]) {
    const unsigned char* vbase = &lds[v_at + lane * 16];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const uint4 vf = *reinterpret_cast<const uint4*>(vbase + n * 1024);
#pragma unroll
      for (int t = 0; t < NQ; ++t) o[t][n] = T::mfma(vf, pf[t], o[t][n]);
    }
  };
  // The paired forms.  A wave issues in program order, so the pairing is written out: one scheduling fence per
  // slot, each slot a piece of the vector stream followed by one or two matrix instructions (the piece issues
  // while the previous slot's product runs; the compiler's scheduler, left alone, puts a block's vector work
  // first and its 16 products back to back).
  static_assert(2 * KS == NT, "QK^T and PV issue the same number of matrix instructions");
  static_assert(KS == 4 || KS == 2, "head_dim 128 or 64");
  constexpr int VOPS = 4 * NQ;                     // pieces of the probability stream: one per pair of scores
  // S_next = QK^T of the image at k_at, under pf = probabilities of S_cur
  auto qk_probs = [&](int k_at, Scores& Sn, const Scores& Sc, uint4 (&pf)[NQ]) {
    int lane_ = lane;                              // fragment addresses from the lane index at each use: three integer
    asm volatile("" : "+v"(lane_));                // instructions instead of registers held (spilled, as it turned out) through the loop
    const int c_ = lane_ & 15, h_ = lane_ >> 4;
    const unsigned char* base[2] = {&lds[k_at + h_ * 256 + (c_ ^ (2 * h_)) * 16], &lds[k_at + h_ * 256 + (c_ ^ (2 * h_ + 1)) * 16]};
    uint4 ka[2], kb[2];                            // two k-steps of K fragments in flight
#pragma unroll
    for (int ks = 0; ks < 2 && ks < KS; ++ks) {
      ka[ks] = *reinterpret_cast<const uint4*>(base[ks & 1] + ks * KFRAG);
      kb[ks] = *reinterpret_cast<const uint4*>(base[ks & 1] + (KS + ks) * KFRAG);
    }
    // The probability stream, one pair of scores a piece, as a three-stage pipeline over the slots: slot s
    // forms the exponent of piece s, the exponentials of piece s - 1 and packs piece s - 2 -- no instruction
    // of a slot waits for another of the same slot (a wave issues in order: a piece done start to end in one
    // slot is a chain of dependent instructions, each waiting out the latency of the one before)
    f32x2 ex[VOPS], pp[VOPS];
    f32x2 part[NQ];                                // the row sum keeps its order: (p01 + p23) + (p45 + p67)
    uint32_t pw[NQ][4];
    auto vslot = [&](int s_) {
      if (s_ < VOPS) {
        const int t = s_ / 4, j = s_ % 4;
        const f32x2 sc = {scale_log2e, scale_log2e}, nm = {neg_m[t], neg_m[t]};
        const f32x2 sv = j == 0 ? f32x2{Sc.a[t][0], Sc.a[t][1]} : j == 1 ? f32x2{Sc.a[t][2], Sc.a[t][3]}
                         : j == 2 ? f32x2{Sc.b[t][0], Sc.b[t][1]} : f32x2{Sc.b[t][2], Sc.b[t][3]};
        ex[s_] = __builtin_elementwise_fma(sv, sc, nm);                                    // v_pk_fma_f32
      }
      if (s_ >= 1 && s_ - 1 < VOPS) {
        const int m = s_ - 1;
        pp[m] = f32x2{__builtin_amdgcn_exp2f(ex[m][0]), __builtin_amdgcn_exp2f(ex[m][1])};
        asm volatile("" : "+v"(pp[m]));            // computed HERE: the optimiser otherwise sinks it to its use, a block later
      }
      if (s_ >= 2 && s_ - 2 < VOPS) {
        const int m = s_ - 2, t = m / 4, j = m % 4;
        pw[t][j] = T::pack2(pp[m][0], pp[m][1]);
        asm volatile("" : "+v"(pw[t][j]));
        if (j == 1) part[t] = pp[m - 1] + pp[m];                                           // v_pk_add_f32
        if (j == 3) {
          l[t] += part[t] + (pp[m - 1] + pp[m]);
          pf[t] = make_uint4(pw[t][0], pw[t][1], pw[t][2], pw[t][3]);
        }
      }
    };
#pragma unroll
    for (int t = 0; t < NQ; ++t) Sn.a[t] = Sn.b[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int t = 0; t < NQ; ++t) {
        const int s_ = 2 * (ks * NQ + t);          // slots of this pair of matrix instructions
        vslot(s_);
        Sn.a[t] = T::mfma(ka[ks & 1], qf[t][ks], Sn.a[t]);
        __builtin_amdgcn_sched_barrier(0);
        vslot(s_ + 1);
        Sn.b[t] = T::mfma(kb[ks & 1], qf[t][ks], Sn.b[t]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ks + 2 < KS) {                           // the fragments of k-step ks + 2 into the registers k-step ks has freed
        ka[ks & 1] = *reinterpret_cast<const uint4*>(base[ks & 1] + (ks + 2) * KFRAG);
        kb[ks & 1] = *reinterpret_cast<const uint4*>(base[ks & 1] + (KS + ks + 2) * KFRAG);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int s_ = 2 * KS * NQ; s_ < VOPS + 2; ++s_) vslot(s_);   // what the matrix instructions were too few for (head_dim 64)
  };
  // O^T += V^T . P^T from the image at v_at, under the lane maxima of S_next
  auto pv_max = [&](int v_at, const uint4 (&pf)[NQ], const Scores& Sn, float (&lm)[NQ]) {
    int lane_ = lane;
    asm volatile("" : "+v"(lane_));
    const unsigned char* vbase = &lds[v_at + lane_ * 16];
    uint4 vf[2];
    vf[0] = *reinterpret_cast<const uint4*>(vbase);
    vf[1] = *reinterpret_cast<const uint4*>(vbase + 1024);
    float m1[NQ] = {}, m2[NQ] = {};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
#pragma unroll
      for (int t = 0; t < NQ; ++t) {
        const int q = n * NQ + t;                  // 0 .. NM - 1; the four steps of a column tile's maximum on slots 4t .. 4t + 3
        if (q / 4 < NQ) {
          const int tt = q / 4, st = q % 4;
          if (st == 0) m1[tt] = max3(Sn.a[tt][0], Sn.a[tt][1], Sn.a[tt][2]);
          if (st == 1) m2[tt] = max3(Sn.b[tt][1], Sn.b[tt][2], Sn.b[tt][3]);
          if (st == 2) m1[tt] = max3(m1[tt], Sn.a[tt][3], Sn.b[tt][0]);
          if (st == 3) lm[tt] = scale_log2e * max3(m1[tt], m2[tt], -INFINITY);
          if (st < 3) asm volatile("" : "+v"(m1[tt]), "+v"(m2[tt]));
          else asm volatile("" : "+v"(lm[tt]));
        }
        o[t][n] = T::mfma(vf[n & 1], pf[t], o[t][n]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (n + 2 < NT) {
        vf[n & 1] = *reinterpret_cast<const uint4*>(vbase + (n + 2) * 1024);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  const int intervals = (steps + 1) / 2;

  // ---- interval 0 staged, its first scores computed
  {
#pragma unroll
    for (int hs = 0; hs < HS; ++hs) {              // an image beyond the last step re-reads the last pages (clamped), unused
      pvec r0[PPT];
      stage_load(r0, hs * 32);
      stage_store(r0, K_AT + hs * KBYTES, hs == 0 ? VA_AT : VB_AT, hs * 32);
    }
  }
  __syncthreads();
  Scores SA, SB;
  float lm[NQ];
  uint4 pf[NQ];
  qk(K_AT, SA);
  mask(0, SA);
  lane_max(SA, lm);
  grow(lm);
  int vb_slot = 0;                                 // ring slot of this interval's B image
  for (int i = 0; i < intervals; ++i) {
    const bool more = i + 1 < intervals;
    const int par = i & 1;
    const int k_at = K_AT + par * 2 * KBYTES;
#ifdef MI355_EXP_EARLY_LOADS
    pvec nxt[HS][PPT];
    if (more) {
#pragma unroll
      for (int hs = 0; hs < HS; ++hs) stage_load(nxt[hs], (2 * i + 2 + hs) * 32);
    }
#endif
    if (2 * i + 1 >= steps) {                                    // the tile ends on an A image (then `more` is false)
      probs(SA, pf);
      pv(VA_AT + par * VBYTES, pf);
      break;
    }
    // A half-step: its exponentials under the B image's QK^T, its PV product over the B scores' maxima
    qk_probs(k_at + KBYTES, SB, SA, pf);
    mask(2 * i + 1, SB);
    // the next interval's pages: asked for here, not at the top of the interval -- their 32 registers then
    // live through one block instead of two (the kernel spilled otherwise); the pages are cache hits for all
    // but the first tile that walks a (sequence, kv head)
#ifndef MI355_EXP_EARLY_LOADS
    pvec nxt[HS][PPT];
    if (more) {                                                  // workgroup-uniform branch
#pragma unroll
      for (int hs = 0; hs < HS; ++hs) stage_load(nxt[hs], (2 * i + 2 + hs) * 32);
    }
#endif
    pv_max(VA_AT + par * VBYTES, pf, SB, lm);
    grow(lm);
    if (!more) {                                                 // last interval: the B half-step has nothing to hide under
      probs(SB, pf);
      pv(VB_AT + vb_slot * VBYTES, pf);
      break;
    }
    const int nslot = vb_slot == 2 ? 0 : vb_slot + 1;
    stage_store(nxt[0], K_AT + (par ^ 1) * 2 * KBYTES, VA_AT + (par ^ 1) * VBYTES, (2 * i + 2) * 32);
    stage_store(nxt[1], K_AT + (par ^ 1) * 2 * KBYTES + KBYTES, VB_AT + nslot * VBYTES, (2 * i + 3) * 32);
    __syncthreads();
    // B half-step, after the barrier: its exponentials under the next interval's first QK^T, its PV product (from
    // the ring slot no staging store touches before the NEXT barrier) over those scores' maxima
    qk_probs(K_AT + (par ^ 1) * 2 * KBYTES, SA, SB, pf);
    mask(2 * i + 2, SA);
    pv_max(VB_AT + vb_slot * VBYTES, pf, SA, lm);
    grow(lm);
    vb_slot = nslot;
  }

#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    float lt = l[t][0] + l[t][1];
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
      MI355_ST(uint2, out, op + 16 * n, make_uint2(w0, w1));
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

  // column tiles per wave: a workgroup covers 64*nq (query, head) pairs.  Two beyond short chunks:
  // two workgroups then share a CU (64 KB of LDS and ~250 registers each) and overlap each other's
  // dependent chains; four tiles per wave (one workgroup per CU) measured slower on every shape.
  // MI355_PREFILL_NQ=1 selects the narrow form for experiments.
  const int64_t pairs = static_cast<int64_t>(max_q_len) * group;
  int nq = pairs <= 64 ? 1 : 2;
  if (const char* e = getenv("MI355_PREFILL_NQ")) {
    if (atoi(e) == 1) nq = 1;
  }
  const int qt = 64 * nq / group;                      // queries per workgroup (>= 4: group <= 16)
  dim3 grid((max_q_len + qt - 1) / qt, num_kv_heads, num_seqs), block(kPrefillThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define MI355_PRE(TT, DD, NN)                                                                           \
  hipLaunchKernelGGL((paged_attn_prefill_kernel<TT, DD, NN>), grid, block, 0, s,                        \
                     static_cast<uint16_t*>(out), out_stride, static_cast<const uint16_t*>(q), q_stride, \
                     static_cast<const uint16_t*>(k_cache), static_cast<const uint16_t*>(v_cache),      \
                     num_blocks, block_tables, max_blocks_per_seq, cu_seqlens, context_lens, num_heads, \
                     num_kv_heads, scale * 1.4426950408889634f, qt)
#define MI355_PRE_NQ(TT, DD)                \
  switch (nq) {                             \
    case 1: MI355_PRE(TT, DD, 1); break;    \
    default: MI355_PRE(TT, DD, 2); break;   \
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
