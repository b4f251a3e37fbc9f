/*
 * mi355_decode.h -- C ABI of libmi355_decode.so: the batched-decode hot path
 * (paged-attention prefill/decode over a block-table KV cache, RMSNorm, RoPE +
 * KV-cache write, SwiGLU gate, sampling) as hand-written HIP for gfx950.
 *
 * What this boundary replaces.  The reference harness never calls these
 * operators itself: it starts an engine container and talks HTTP to it.  The
 * engine it starts is the third-party image vllm/vllm-openai:v0.9.1
 * (benchmark-vllm/vllm_manager.py:371-375; local variant vllm-cpu-env:latest,
 * benchmark-vllm/vllm_templates.py:124), configured only through CLI flags
 * (benchmark-vllm/vllm_templates.py:50-87, vllm_manager.py:399-416).  Each entry
 * point below stands in for the operator of that engine's model step named in
 * its comment, and cites the reference flag that parameterises it.  The Python
 * host (vllm-backend-public_amd/mi355_decode) binds this file with ctypes.
 *
 * Conventions
 *  - plain C: raw device pointers, explicit sizes/strides (in ELEMENTS), no
 *    framework types.  The caller owns every buffer; nothing is allocated here.
 *  - `stream` is a hipStream_t passed as void*; every call only enqueues work on
 *    it (no synchronisation, safe under HIP graph capture).
 *  - return value: MI355_OK or a negative MI355_ERR_*; the reason is kept per
 *    thread and returned by mi355_last_error().  Arguments are validated on the
 *    host before any launch; a rejected call launches nothing.
 *  - 16-bit tensors are bf16 or fp16 as selected by `dtype`; all pointers to
 *    them must be 16-byte aligned and row strides multiples of 8 elements.
 *  - one process drives one GPU (tensor parallel = one process per GPU); calls
 *    are re-entrant across streams.
 *
 * Paged KV cache layout, per layer (block_size is fixed at 16 tokens):
 *    K cache  [num_blocks][kv_heads][16][head_dim]     token rows
 *    V cache  [num_blocks][kv_heads][head_dim][16]     token columns (transposed
 *             so a PV MFMA operand is one 16-byte load)
 *    slot = block_id * 16 + offset.
 */
#ifndef MI355_DECODE_H
#define MI355_DECODE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_ABI_VERSION 1
#define MI355_BLOCK_SIZE 16

enum mi355_dtype { MI355_BF16 = 0, MI355_F16 = 1, MI355_F32 = 2 };

enum mi355_status {
  MI355_OK = 0,
  MI355_ERR_INVALID = -1,     /* bad argument (null, misaligned, size) */
  MI355_ERR_UNSUPPORTED = -2, /* shape/dtype outside what the kernels cover */
  MI355_ERR_LAUNCH = -3       /* HIP reported an error at launch */
};

/* ABI version of the loaded library (== MI355_ABI_VERSION at build time). */
int32_t mi355_abi_version(void);

/* Message for the last non-OK status returned on the calling thread. */
const char* mi355_last_error(void);

/* Name of the GPU architecture the kernels were compiled for ("gfx950"). */
const char* mi355_target_arch(void);

/*
 * Bounds-checking build (no counterpart in the reference: a debugging aid of this library).
 * libmi355_decode_bounds.so is the same sources compiled with -DMI355_BOUNDS (`make bounds`):
 * every global-memory access of every kernel is first compared with the device allocation its
 * base pointer lies in; an access that leaves it is skipped (a load yields zeros) and the first
 * one is recorded, so a faulty index is named instead of faulting the GPU.  Many times slower.
 *   mi355_bounds_enabled      1 in that build, 0 in the library as shipped (where the other two
 *                             return MI355_ERR_UNSUPPORTED).
 *   mi355_bounds_set_ranges   the live device allocations as `count` (<= 4096) pairs {lo, hi} of
 *                             byte addresses, [lo, hi), sorted and disjoint; count 0 switches the
 *                             checks off (every access passes).  Synchronous; not during capture.
 *   mi355_bounds_report       out[0..5]: out[0] = refused accesses since the last report; of the
 *                             first one out[1] = source line, out[2] = base, out[3] = address,
 *                             out[4] = size, `file` (if not null) = its source file; out[5] =
 *                             accesses whose base lies in no range of the list (memory allocated
 *                             after the list was set: performed, only counted).  Clears the
 *                             record.  The caller synchronises the streams first.
 */
int32_t mi355_bounds_enabled(void);
int32_t mi355_bounds_set_ranges(const uint64_t* lo_hi, int32_t count);
int32_t mi355_bounds_report(uint64_t* out, char* file, int32_t file_len);

/*
 * RMSNorm:  out = round(round(x * rsqrt(mean(x^2) + eps)) * weight), fp32
 * reduction.  x, out [num_tokens, hidden] with the given row strides; weight
 * [hidden].  Replaces the engine's rms_norm (input_layernorm / final norm of the
 * model selected by --model, vllm_templates.py:51; arithmetic type from --dtype,
 * vllm_templates.py:57).
 */
int32_t mi355_rms_norm(void* out, const void* x, const void* weight,
                       int64_t num_tokens, int32_t hidden,
                       int64_t x_stride, int64_t out_stride,
                       float eps, int32_t dtype, void* stream);

/*
 * Residual add fused with RMSNorm:  residual = round(x + residual);
 * out = rms_norm(residual).  `out` may alias `x`.  Replaces the engine's
 * fused_add_rms_norm (same flags as above).
 */
int32_t mi355_fused_add_rms_norm(void* out, const void* x, void* residual,
                                 const void* weight, int64_t num_tokens,
                                 int32_t hidden, int64_t x_stride,
                                 int64_t out_stride, int64_t residual_stride,
                                 float eps, int32_t dtype, void* stream);

/*
 * First kernel of a decode step: residual[t, :] = embed[token_ids[t], :] and
 * out[t, :] = rms_norm(residual[t, :]) in one launch (the embedding lookup of
 * the served model + the first layer's input norm).  embed [vocab, hidden], row
 * stride embed_stride; an id outside [0, vocab) reads row 0.
 */
int32_t mi355_embed_rms_norm(void* out, void* residual, const void* embed,
                             const int64_t* token_ids, int64_t vocab,
                             const void* weight, int64_t num_tokens, int32_t hidden,
                             int64_t embed_stride, int64_t out_stride,
                             int64_t residual_stride, float eps, int32_t dtype,
                             void* stream);

/*
 * LayerNorm with weight and bias, the norm of OPT-style decoders (the model the harness
 * launches with its defaults, benchmark-vllm/models.py:7 model_name "facebook/opt-125m"):
 *   y = round(((x - mean) * rsqrt(var + eps)) * weight + bias),  var = mean((x - mean)^2),
 * statistics and the affine map in fp32, one rounding -- what the engine's LayerNorm computes on
 * a 16-bit tensor.  x, out [num_tokens, hidden] with the given row strides (out may alias x);
 * weight, bias [hidden].
 */
int32_t mi355_layer_norm(void* out, const void* x, const void* weight, const void* bias, int64_t num_tokens,
                         int32_t hidden, int64_t x_stride, int64_t out_stride, float eps, int32_t dtype,
                         void* stream);

/*
 * residual = round(x + residual) (in place), out = layer_norm(residual): the residual add of a
 * decoder layer and the norm that follows it, one pass.
 */
int32_t mi355_fused_add_layer_norm(void* out, const void* x, void* residual, const void* weight,
                                   const void* bias, int64_t num_tokens, int32_t hidden, int64_t x_stride,
                                   int64_t out_stride, int64_t residual_stride, float eps, int32_t dtype,
                                   void* stream);

/*
 * First kernel of an OPT-style step: residual[t] = round(embed[token_ids[t]] + pos_embed[positions[t]
 * + position_offset]) (learned positions; OPT stores them with an offset of 2), out[t] =
 * layer_norm(residual[t]) with the first layer's norm.  embed [vocab, hidden], pos_embed
 * [num_positions, hidden] with the given row strides.  An id outside [0, vocab) reads row 0, a
 * position outside the table its nearest row (padded rows of a batch bucket may hold anything).
 */
int32_t mi355_embed_pos_layer_norm(void* out, void* residual, const void* embed, const int64_t* token_ids,
                                   int64_t vocab, int64_t embed_stride, const void* pos_embed,
                                   const int64_t* positions, int64_t num_positions, int64_t position_offset,
                                   int64_t pos_stride, const void* weight, const void* bias,
                                   int64_t num_tokens, int32_t hidden, int64_t out_stride,
                                   int64_t residual_stride, float eps, int32_t dtype, void* stream);

/*
 * out = x < 0 ? 0 : x over num_elements (a multiple of 8) contiguous 16-bit values; NaN and -0.0
 * pass through.  The activation of an OPT-style MLP (fc1 -> ReLU -> fc2).  out may alias x.
 */
int32_t mi355_relu(void* out, const void* x, int64_t num_elements, int32_t dtype, void* stream);

/*
 * Rotary embedding (NeoX / rotate-half pairing (i, i + head_dim/2)) applied in
 * place to the q and k parts of the packed QKV projection output, fused with the
 * paged KV-cache store of k (rotated) and v.
 *   qkv          [num_tokens, (num_heads + 2*num_kv_heads) * head_dim]
 *   positions    int64 [num_tokens]   (clamped to max_position - 1)
 *   cos_sin      [max_position, head_dim] in `dtype`: cos | sin halves
 *   slot_mapping int64 [num_tokens]; slot < 0 or >= num_blocks*16: not stored
 * Rounds at the points a 16-bit model does: each product and the sum.
 * Replaces the engine's rotary_embedding + reshape_and_cache; the page size is
 * the harness's --block-size (vllm_templates.py:54; benchmark-vllm/models.py:10).
 */
int32_t mi355_rope_kv_write(void* qkv, int64_t qkv_stride,
                            const int64_t* positions, const void* cos_sin,
                            int32_t max_position, const int64_t* slot_mapping,
                            void* k_cache, void* v_cache, int32_t num_blocks,
                            int64_t num_tokens, int32_t num_heads,
                            int32_t num_kv_heads, int32_t head_dim,
                            int32_t dtype, void* stream);

/*
 * Workspace (bytes) mi355_paged_attn_decode needs for a given split count.
 * 0 when num_splits <= 1.
 */
int64_t mi355_paged_attn_decode_workspace(int32_t num_seqs, int32_t num_heads,
                                          int32_t head_dim, int32_t num_splits);

/*
 * Decode attention, one new query token per sequence, keys/values read from the
 * paged cache through the block table:
 *   out[b, h, :] = softmax(q[b, h, :] . K_b^T * scale) V_b   over context_lens[b]
 * tokens (the new token's k/v already stored).  fp32 softmax and accumulation,
 * one rounding of the result.  Grouped-query: num_heads / num_kv_heads (<= 16)
 * query heads share a kv head and its pages are read once for the group.
 *   q            [num_seqs, num_heads, head_dim], token stride q_stride
 *   out          [num_seqs, num_heads, head_dim] contiguous
 *   block_tables int32 [num_seqs, max_blocks_per_seq]
 *   context_lens int32 [num_seqs]  (0 -> zeros)
 *   num_splits   >= 1: the context of a sequence is cut into this many ranges
 *                handled by separate workgroups and merged by a second kernel;
 *                needs `workspace` of mi355_paged_attn_decode_workspace() bytes.
 * Replaces the engine's paged_attention_v1/v2 (batch = --max-num-seqs,
 * vllm_templates.py:53; context bound = --max-model-len, vllm_templates.py:72-73).
 */
int32_t mi355_paged_attn_decode(void* out, const void* q, int64_t q_stride,
                                const void* k_cache, const void* v_cache,
                                int32_t num_blocks, const int32_t* block_tables,
                                int32_t max_blocks_per_seq,
                                const int32_t* context_lens, int32_t num_seqs,
                                int32_t num_heads, int32_t num_kv_heads,
                                int32_t head_dim, float scale,
                                int32_t num_splits, void* workspace,
                                int64_t workspace_bytes, int32_t dtype,
                                void* stream);

/*
 * mi355_paged_attn_decode for callers that cannot allocate device memory (code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc):
 * the same kernels, with the split-KV partials in a device array built into the
 * library (MI355_PAGED_ATTN_DECODE_STATIC_MAX_BYTES per GPU the library runs on,
 * held until the process ends) instead of a caller workspace.  No counterpart in
 * the reference.  Calls no hipMalloc, hipFree or other allocating routine; the first
 * call on a device looks the array's address up (hipGetSymbolAddress) -- make it
 * outside stream capture.  MI355_ERR_UNSUPPORTED when
 * mi355_paged_attn_decode_workspace(num_seqs, num_heads, head_dim, num_splits)
 * exceeds the array.  All calls on one GPU share the array: calls that overlap on
 * one GPU (different streams) race; calls on one stream are ordered.  Nothing in
 * the array has to be cleared between calls (a launch reads only what it wrote).
 */
#define MI355_PAGED_ATTN_DECODE_STATIC_MAX_BYTES (16 * 1024 * 1024)
int32_t mi355_paged_attn_decode_static(void* out, const void* q, int64_t q_stride,
                                       const void* k_cache, const void* v_cache,
                                       int32_t num_blocks, const int32_t* block_tables,
                                       int32_t max_blocks_per_seq,
                                       const int32_t* context_lens, int32_t num_seqs,
                                       int32_t num_heads, int32_t num_kv_heads,
                                       int32_t head_dim, float scale,
                                       int32_t num_splits, int32_t dtype,
                                       void* stream);

/*
 * Decode attention with the rotary embedding and the KV store of the new token
 * fused in: the projection of the step (packed q|k|v rows) is rotated and stored
 * exactly as mi355_rope_kv_write would, then attended to exactly as
 * mi355_paged_attn_decode would -- in one launch, and the rotated query never
 * leaves the registers.  The projection comes either as 16-bit rows (qkv,
 * qkv_stride; partial == NULL) or as the fp32 partial-sum slabs of a K-split
 * GEMM (partial fp32 [num_slabs][num_seqs][width] + bias, as in
 * mi355_rope_kv_write_partial; qkv may then be NULL).  The new token of
 * sequence b sits at position context_lens[b] - 1 (that is its rotary position),
 * and its slot must be the page slot of that token in the block table, as in
 * every decode step; slot < 0: nothing stored.  Replaces the engine's
 * rotary_embedding + reshape_and_cache + paged_attention_v1/v2 at decode.
 */
int32_t mi355_paged_attn_decode_fused(void* out, const void* qkv, int64_t qkv_stride,
                                      const float* partial, int32_t num_slabs,
                                      int64_t slab_stride, const void* bias,
                                      const void* cos_sin, int32_t max_position,
                                      const int64_t* slot_mapping, void* k_cache,
                                      void* v_cache, int32_t num_blocks,
                                      const int32_t* block_tables,
                                      int32_t max_blocks_per_seq,
                                      const int32_t* context_lens, int32_t num_seqs,
                                      int32_t num_heads, int32_t num_kv_heads,
                                      int32_t head_dim, float scale,
                                      int32_t num_splits, void* workspace,
                                      int64_t workspace_bytes, int32_t dtype,
                                      void* stream);

/*
 * Prefill (and chunked-prefill) attention: causal attention of packed
 * variable-length query chunks over keys/values already stored in the paged
 * cache.  Sequence i owns query rows [cu_seqlens[i], cu_seqlens[i+1]) sitting at
 * absolute positions [context_lens[i] - q_len_i, context_lens[i]); the row at
 * position p attends keys [0, p].
 *   q, out       [num_tokens, num_heads, head_dim], token strides q_stride /
 *                out_stride
 *   cu_seqlens   int32 [num_seqs + 1]
 *   max_q_len    upper bound of q_len_i known to the host (sizes the grid)
 * Replaces the engine's varlen flash-attention prefill.
 */
int32_t mi355_paged_attn_prefill(void* out, int64_t out_stride, const void* q,
                                 int64_t q_stride, const void* k_cache,
                                 const void* v_cache, int32_t num_blocks,
                                 const int32_t* block_tables,
                                 int32_t max_blocks_per_seq,
                                 const int32_t* cu_seqlens,
                                 const int32_t* context_lens, int32_t num_seqs,
                                 int32_t max_q_len, int32_t num_heads,
                                 int32_t num_kv_heads, int32_t head_dim,
                                 float scale, int32_t dtype, void* stream);

/*
 * SwiGLU gate:  out[t, :] = round(round(silu(x[t, :F])) * x[t, F:2F]).
 * x [num_tokens, 2F] (row stride x_stride), out [num_tokens, F] contiguous.
 * Replaces the engine's silu_and_mul.
 */
int32_t mi355_silu_and_mul(void* out, const void* x, int64_t x_stride,
                           int64_t num_tokens, int32_t inter, int32_t dtype,
                           void* stream);

/*
 * Consumers of the fp32 partial-sum slabs partial[num_slabs][M][N] of a K-split
 * mi355_linear_tiled (epilogue 2) for the call sites whose next step needs a 16-bit
 * tensor.  The slabs are added in slab order, so results do not depend on scheduling.
 *   mi355_sum_slabs:           y[M, N] = round(sum_s partial[s] + bias)   (bias may be NULL)
 *     -- the attention-output / down projection of a tensor-parallel rank before its
 *     all-reduce (--tensor-parallel-size, benchmark-vllm/vllm_templates.py:55)
 *   mi355_silu_and_mul_slabs:  slabs of a gate_up projection tiled with interleave_halves
 *     (16-column tiles alternate gate / up): out[M, N/2] =
 *     round(round(silu(round(sum gate))) * round(sum up)), what epilogue 1 computes
 * num_slabs 1..8; N a multiple of 8 (SwiGLU: 32); y / out row strides in elements.
 * Same role as the engine's decode GEMM epilogues (see mi355_linear_tiled).
 */
int32_t mi355_sum_slabs(void* y, int64_t y_stride, const float* partial, int32_t num_slabs,
                        const void* bias, int32_t M, int32_t N, int32_t dtype, void* stream);
int32_t mi355_silu_and_mul_slabs(void* out, int64_t out_stride, const float* partial,
                                 int32_t num_slabs, int32_t M, int32_t N, int32_t dtype,
                                 void* stream);

/*
 * Greedy sampling: out_ids[b] = argmax_v logits[b, v]; lowest index on ties, a
 * NaN counts as the maximum (torch.argmax semantics).  logits may be bf16, fp16
 * or fp32 (logits_dtype); row stride in elements.  Serves requests with
 * temperature == 0 on POST /v1/completions
 * (benchmark-deployer/docs/deployment/post_run.md:46).
 */
int32_t mi355_argmax(int64_t* out_ids, const void* logits,
                     int64_t logits_stride, int32_t num_rows, int32_t vocab,
                     int32_t logits_dtype, void* stream);

/*
 * mi355_argmax for few rows (a decode batch): every row is cut into column ranges
 * handled by separate workgroups whose best keys are merged with a 64-bit atomic
 * max into keys[row], then turned into token ids (same ids as mi355_argmax, on
 * every launch).  `keys` int64 [num_rows]: zero before the first call, left zero
 * by every call.  out_ids2 (may be NULL) receives the same ids a second time (the
 * next step's input tokens, so that they never leave the GPU).
 */
int32_t mi355_argmax_chunked(int64_t* out_ids, int64_t* out_ids2, int64_t* keys,
                             const void* logits, int64_t logits_stride,
                             int32_t num_rows, int32_t vocab, int32_t logits_dtype,
                             void* stream);

/*
 * Greedy sampling over a vocabulary that is split across tensor-parallel ranks
 * (--tensor-parallel-size > 1, benchmark-vllm/vllm_templates.py:102-103): each
 * rank holds the logits of vocabulary rows [index_offset, index_offset + vocab).
 * out_keys[b] = the 64-bit key of the rank's best token,
 *     ((order-preserving image of the logit) << 32 | (0xFFFFFFFF - global index))
 *     XOR 0x8000000000000000,
 * so that a SIGNED 64-bit max over the ranks' keys (one all-reduce of num_rows
 * integers instead of a gather of the logits) is the key of the global argmax
 * with mi355_argmax's semantics (lowest index on ties, NaN as the maximum);
 * mi355_argmax_key_to_index turns the reduced keys into token ids.
 */
int32_t mi355_argmax_shard_key(int64_t* out_keys, const void* logits,
                               int64_t logits_stride, int32_t num_rows,
                               int32_t vocab, int32_t logits_dtype,
                               int64_t index_offset, void* stream);

int32_t mi355_argmax_key_to_index(int64_t* out_ids, const int64_t* keys,
                                  int32_t num_rows, void* stream);

/*
 * Random sampling with temperature, top-k and top-p, one row per sequence, with the kept set
 * the published samplers define (top-k, then top-p over what top-k kept, both on
 * logits / temperature -- HF's TopKLogitsWarper + TopPLogitsWarper, the stock engine's mask):
 *   top_k[b] in (0, vocab)  keeps every token whose logit reaches the k-th largest logit
 *                           (ties at that value all stay); any k up to the vocabulary
 *   top_p[b] < 1            orders the survivors by (logit desc, index asc) and keeps the
 *                           shortest prefix whose probability mass reaches top_p of the
 *                           survivors' mass -- of the WHOLE vocabulary when top-k is off --
 *                           the crossing token included, never fewer than one
 * Masses are exact integers, F_i = floor(exp((logit_i - logit_max) / temperature[b]) * 2^40)
 * summed in 64 bits, the threshold max(1, ceil(double(top_p) * double(total))): no rounding, so
 * the set does not depend on any order of additions.  The token is drawn from the kept set
 * (the whole row when neither cut is set) in the Gumbel-max form,
 * token = argmax_i(logit_i / temperature[b] - ln(-ln(u_i))), one uniform u_i per
 * (seeds[b], steps[b], i) from a counter hash, ties to the lowest index; exp and ln are fixed
 * polynomial forms in fp32.  A (logits, parameters, seed, step) tuple therefore always yields
 * the same token, on any launch.  Rows with temperature[b] <= 0 get the greedy token
 * (mi355_argmax semantics); a row that holds a NaN yields the index of its first NaN from
 * either.  Serves temperature/top_k/top_p/seed of POST /v1/completions
 * (benchmark-deployer/docs/deployment/post_run.md:46).
 */
int32_t mi355_sample_topk_topp(int64_t* out_ids, const void* logits,
                               int64_t logits_stride, int32_t num_rows,
                               int32_t vocab, int32_t logits_dtype,
                               const float* temperature, const int32_t* top_k,
                               const float* top_p, const int64_t* seeds,
                               const int64_t* steps, void* stream);

/*
 * The same, told by the caller whether any row of the batch can have an active cut
 * (temperature > 0 and top_k in (0, vocab) or top_p < 1).  A row whose kept set cannot be
 * found among its 4096 largest logits (top_k > 1024, a nucleus or a tie reaching further) needs
 * several passes over the row; with cut_rows != 0 a fixed sequence of follow-up launches
 * spreads those passes over many workgroups (rows 0 .. 255 of a batch of fewer than 512 rows;
 * the launches return at once when no row needs them, but cost their launch gaps, a few
 * microseconds together), with cut_rows == 0 they are left out and such a row, should there be one
 * after all, is served by its one workgroup.  The tokens are the same either way.
 * mi355_sample_topk_topp is this with cut_rows = 1.  The follow-up launches keep their
 * histograms in one scratch area per process and GPU: calls that may overlap on one GPU
 * (different streams) must pass cut_rows = 0.
 */
int32_t mi355_sample_topk_topp_hinted(int64_t* out_ids, const void* logits,
                                      int64_t logits_stride, int32_t num_rows,
                                      int32_t vocab, int32_t logits_dtype,
                                      const float* temperature, const int32_t* top_k,
                                      const float* top_p, const int64_t* seeds,
                                      const int64_t* steps, int32_t cut_rows, void* stream);

/*
 * Log-probabilities for the `logprobs` field of POST /v1/completions
 * (benchmark-deployer/docs/deployment/post_run.md:46):
 *   out_lse[b]     = log(sum_v exp(logits[b, v]))            (fp32, fixed-order sum)
 *   out_logprob[b] = logits[b, token_ids[b]] - out_lse[b]    (-inf for an id outside the vocabulary)
 * Either output may be NULL.  Any other logit of the row becomes a log-probability
 * by subtracting out_lse[b].  logits may be bf16, fp16 or fp32; row stride in elements.
 */
int32_t mi355_token_logprobs(float* out_logprob, float* out_lse, const void* logits,
                             int64_t logits_stride, int32_t num_rows, int32_t vocab,
                             int32_t logits_dtype, const int64_t* token_ids,
                             void* stream);

/*
 * The k most likely tokens of every row, for the `top_logprobs` lists of the completion routes
 * (benchmark-deployer/docs/deployment/post_run.md:46): out_ids[b, j] / out_vals[b, j], j < k, in
 * (logit descending, token id ascending) order -- an exact selection, ties to the lower id, a NaN
 * logit on top as in mi355_argmax.  out_vals holds the logit as fp32, minus lse[b] when `lse`
 * (mi355_token_logprobs' log-sum-exp) is given, i.e. the log-probability.  1 <= k <= 64; with
 * k > vocab the tail is (-1, -inf).
 */
int32_t mi355_topk_logits(int64_t* out_ids, float* out_vals, const void* logits,
                          int64_t logits_stride, int32_t num_rows, int32_t vocab,
                          int32_t logits_dtype, int32_t k, const float* lse, void* stream);

/*
 * Decode-time linear layer  y[M, N] = x[M, K] . W[N, K]^T (+ bias[N]),  M <= 128:
 * the weight matrix is streamed from HBM exactly once (fp32 accumulation on the
 * matrix cores, one rounding of the result).  W row-major [N, K] contiguous;
 * x, y row-major with the given row strides; bias may be NULL.  K must be a
 * multiple of 32.  k_slices: 0 = choose, or 1/2/4 waves splitting K per 16-row
 * tile of W (partial sums are combined in a fixed order: results do not depend
 * on scheduling).  Replaces the engine's qkv/o/gate_up/down/lm_head GEMMs at
 * decode batch sizes (--max-num-seqs, vllm_templates.py:53), which carry ~95 %
 * of the bytes a decode step moves.
 */
int32_t mi355_linear_decode(void* y, int64_t y_stride, const void* x,
                            int64_t x_stride, const void* weight,
                            const void* bias, int32_t M, int32_t N, int32_t K,
                            int32_t k_slices, int32_t dtype, void* stream);

/*
 * Fragment-tiled weights for mi355_linear_tiled.  A row-major [N, K] matrix is
 * re-laid once (at model load) into 1 KB blocks of 16 rows x 32 k stored in the
 * register order of the MFMA operand, blocks ordered [k-step / 4][tile][k-step % 4]
 * (K a multiple of 128): the waves of a launch, one per tile and all near the
 * same k, then read one contiguous region together; rows are zero-padded to a
 * multiple of 16.  mi355_tiled_weight_elems gives the element
 * count of the destination.  With interleave_halves != 0 the 16-row tiles
 * alternate between the first and the second half of the rows (gate / up of a
 * fused SwiGLU projection), which lets the GEMM epilogue apply the gate.
 */
int64_t mi355_tiled_weight_elems(int32_t N, int32_t K);
int32_t mi355_tile_weights(void* dst, const void* src, int32_t N, int32_t K,
                           int32_t interleave_halves, void* stream);

/*
 * The inverse re-layouts: fragment-tiled weights back to a row-major [N, K]
 * matrix in the model's 16-bit type (rows the tiling padded are dropped).  The
 * fp8 form decodes, dst = round(fp8 * row scale): the matrix the tiled GEMM
 * multiplies by.  An engine that keeps ONLY the tiled copy of its decode weights
 * (half the weight memory) rebuilds a row-major matrix into a scratch buffer
 * when a library GEMM (prefill, batches beyond 128 rows) needs one.  No
 * counterpart in the reference: memory layout of this engine only.
 */
int32_t mi355_untile_weights(void* dst, const void* src_tiled, int32_t N, int32_t K,
                             int32_t interleave_halves, void* stream);
int32_t mi355_untile_weights_fp8(void* dst, const void* src_tiled, const float* scale_tiled,
                                 int32_t N, int32_t K, int32_t interleave_halves,
                                 int32_t dtype, void* stream);

/*
 * Decode-time linear layer on fragment-tiled weights, M <= 256 (a workgroup covers
 * up to 128 rows; 129 .. 256 rows -- the step of a tensor-parallel rank that serves
 * --max-num-seqs 256, benchmark-vllm/models.py:9,11 -- run as two row blocks of one
 * launch, every geometry of the 65 .. 128-row class, epilogues 0-2):
 * y[M, N] = x[M, K] . W[N, K]^T.  Every weight load is 1 KB of contiguous memory
 * and each wave streams one 16-row tile of W; x is staged through LDS once per
 * workgroup, one K slice at a time.  tiles_per_wg = 4 or 8 tiles (= waves) per
 * workgroup with wide x slices, or 104 / 108 for the same with narrow slices (two
 * workgroups per CU); 208 / 216 (33..128 rows, epilogues 0-2): every wave streams two
 * adjacent tiles and uses each x fragment it reads from LDS for both -- 4 waves
 * with narrow slices / 8 waves with wide ones; 210 / 212 / 214: 5 / 6 / 7 waves with wide slices
 * (10 / 12 / 14 tiles per workgroup, for tile counts that 16 would spread over too few CUs).  Slice width: 1024 (narrow 512) for M <= 32, halved for
 * M <= 64 and again for M <= 128; K must be a multiple of it.  302 / 304 / 308 (epilogues 0-2,
 * 16-bit weights): workgroups of 8 waves in which 2 / 4 / 8 waves share a tile and
 * split its K range between them (partial sums added through LDS in wave order), for matrices
 * with few tiles -- the slices of a tensor-parallel rank -- whose output must be 16-bit or gated
 * and so cannot be split across workgroups; 402 / 404: the same with 512-wide slices (the SwiGLU
 * epilogue needs at most 4 waves per tile, and a wave at least 128 columns of the slice).  Built
 * geometries: M <= 32: 4, 8, 104, 108, 302, 304, 308, 402, 404; M <= 64: 8, 104, 108, 208, 210,
 * 212, 214, 216, 302, 304, 402; beyond 64 rows: 8, 210, 212, 214, 216, 302.  tiles_per_wg + 1000 /
 * + 2000 (epilogues 0-2): rows beyond 64 / 32 run as row blocks of 64 / 32 rows of one launch, each
 * block by the geometry (and slice width) of that row class -- a wave then reads half / a quarter
 * of the x fragments from LDS per weight fragment, and the blocks of a column block read the same
 * weights, once from HBM; for the compute-shaped slices of a tensor-parallel rank at 128 / 256
 * rows.  k_groups (at most K / slice)
 * splits K over workgroups; the slices are dealt to the groups as evenly as they
 * go, K / slice need not be a multiple of k_groups.
 *   epilogue 0: y = round(sum + bias)                       (k_groups == 1)
 *   epilogue 1: weights tiled with interleave_halves; y[M, N/2] =
 *               round(round(silu(round(gate))) * round(up))  (k_groups == 1)
 *   epilogue 2: fp32 partial sums to partial[k_groups][M][N]; the consumers
 *               (mi355_fused_add_rms_norm_partial, mi355_rope_kv_write_partial,
 *               mi355_sum_slabs, mi355_silu_and_mul_slabs)
 *               add the slabs in order, so results do not depend on scheduling.
 * Same role as mi355_linear_decode (the engine's decode GEMMs, --max-num-seqs,
 * vllm_templates.py:53) with the weight layout chosen for this chip.
 */
int32_t mi355_linear_tiled(void* y, int64_t y_stride, void* partial,
                           const void* x, int64_t x_stride, const void* w_tiled,
                           const void* bias, int32_t M, int32_t N, int32_t K,
                           int32_t tiles_per_wg, int32_t k_groups,
                           int32_t epilogue, int32_t dtype, void* stream);

/*
 * mi355_fused_add_rms_norm whose x is the output of a K-split linear layer still
 * in partial-sum form:  x = round(partial[0] + ... + partial[num_slabs-1] + bias)
 * (fp32, slab order), then residual = round(x + residual); out =
 * rms_norm(residual).  partial is fp32 [num_slabs][num_tokens][hidden] with
 * slab_stride elements between slabs; bias [hidden] may be NULL.  Finishes the
 * o_proj / down_proj GEMM, the residual add and the following norm in one pass
 * (same engine operators as mi355_fused_add_rms_norm).
 */
int32_t mi355_fused_add_rms_norm_partial(void* out, const float* partial,
                                         int32_t num_slabs, int64_t slab_stride,
                                         const void* bias, void* residual,
                                         const void* weight, int64_t num_tokens,
                                         int32_t hidden, int64_t out_stride,
                                         int64_t residual_stride, float eps,
                                         int32_t dtype, void* stream);

/*
 * mi355_rope_kv_write whose packed QKV rows are still in partial-sum form:
 * qkv = round(partial[0] + ... + partial[num_slabs-1] + bias) is formed first
 * (fp32, slab order; partial fp32 [num_slabs][num_tokens][width], width =
 * (num_heads + 2*num_kv_heads) * head_dim, bias [width] or NULL), then q and k
 * are rotated, and q, k, v are written to `qkv` (16-bit rows) and k, v to the
 * cache exactly as mi355_rope_kv_write leaves them.
 */
int32_t mi355_rope_kv_write_partial(void* qkv, int64_t qkv_stride,
                                    const float* partial, int32_t num_slabs,
                                    int64_t slab_stride, const void* bias,
                                    const int64_t* positions, const void* cos_sin,
                                    int32_t max_position,
                                    const int64_t* slot_mapping, void* k_cache,
                                    void* v_cache, int32_t num_blocks,
                                    int64_t num_tokens, int32_t num_heads,
                                    int32_t num_kv_heads, int32_t head_dim,
                                    int32_t dtype, void* stream);

/*
 * mi355_linear_tiled with the neighbouring RMSNorm / residual-add / K-split
 * reduction fused into the GEMM launch (tiles_per_wg 8 or 104 for M <= 64, 8
 * beyond; norm prologue and epilogue 4: M <= 32):
 *   norm_weight != NULL (epilogues 0-2): x is formed while it is staged,
 *     x = round(round(x_in * inv_rms[row]) * norm_weight), inv_rms[row] =
 *     rsqrt(sum_q norm_sumsq[q][row] / K + eps) with norm_sumsq fp32
 *     [norm_parts][32] (what epilogue 4 of the previous GEMM wrote), added in a
 *     fixed order.  x_in is the residual stream.
 *   epilogue 3: K split over k_groups workgroups and reduced in the same launch:
 *     y = round(partial[0] + ... + partial[k_groups-1] + bias).  Every workgroup
 *     writes its slab and draws a ticket from counters[column group]; the one
 *     drawing the last ticket adds the slabs in slab order, so the result does
 *     not depend on scheduling, and nobody waits.  counters: uint32
 *     [ceil(N / (16 * tiles))], zero before the first launch, left zero.
 *   epilogue 4: as 3, then residual = round(y + residual) in place (y is not
 *     stored) and sumsq_out[column group][row] = sum of residual^2 over the
 *     column group (fp32 [ceil(N / (16 * tiles))][32]): the input of the next
 *     GEMM's norm prologue.
 * partial (fp32 [k_groups][M][N]) is scratch for epilogues 3 and 4.  Launches
 * that share counters / partial must be ordered on one stream.  Replaces
 * fused_add_rms_norm + the GEMM on either side of it (same engine operators as
 * mi355_fused_add_rms_norm and mi355_linear_decode).
 */
int32_t mi355_linear_tiled_fused(void* y, int64_t y_stride, float* partial,
                                 const void* x, int64_t x_stride,
                                 const void* w_tiled, const void* bias,
                                 int32_t M, int32_t N, int32_t K,
                                 int32_t tiles_per_wg, int32_t k_groups,
                                 int32_t epilogue, const void* norm_weight,
                                 const float* norm_sumsq, int32_t norm_parts,
                                 float eps, void* residual,
                                 int64_t residual_stride, float* sumsq_out,
                                 uint32_t* counters, int32_t dtype, void* stream);

/*
 * fp8 weight copies for the decode GEMM (--quantization fp8, vllm_manager.py:405-407,
 * vllm_templates.py:75-76): OCP e4m3fn weights with one fp32 scale per weight row,
 * scale = max|row| / 448, q = fp8_rne(w / scale), laid out in 1 KB blocks of 16
 * rows x 64 k in MFMA operand order (a lane's 16 bytes are its elements of two
 * consecutive k-steps), blocks ordered [k / 128][tile][(k / 64) % 2] like the 16-bit
 * tiling.  mi355_tile_weights_fp8 quantises and tiles a row-major
 * 16-bit [N, K] matrix (K % 128 == 0) into dst (mi355_tiled_weight_fp8_bytes(N, K)
 * bytes) and writes the scales to scale_tiled (fp32 [ceil(N/16) * 16], in the row
 * order of the tiles, so with interleave_halves gate and up rows alternate tile
 * by tile).
 */
int64_t mi355_tiled_weight_fp8_bytes(int32_t N, int32_t K);
int32_t mi355_tile_weights_fp8(void* dst, float* scale_tiled, const void* src,
                               int32_t N, int32_t K, int32_t interleave_halves,
                               int32_t dtype, void* stream);

/*
 * mi355_linear_tiled on fp8 weights (weight-only quantisation: x stays 16-bit).
 * The weight stream is half the bytes; each lane turns its fp8 elements into the
 * 16-bit MFMA operand in registers (exact), the products accumulate in fp32, and
 * the row scale is applied to the accumulator before the epilogue:
 *   y = round(sum_k x[m,k] q[n,k] * scale[n] (+ bias[n])).
 * Epilogues 0-2 and geometries as mi355_linear_tiled_fused (tiles_per_wg 8 or 104
 * up to 64 rows, 8 beyond).
 */
int32_t mi355_linear_tiled_w8(void* y, int64_t y_stride, void* partial,
                              const void* x, int64_t x_stride,
                              const void* w_tiled_fp8, const float* w_scale,
                              const void* bias, int32_t M, int32_t N, int32_t K,
                              int32_t tiles_per_wg, int32_t k_groups,
                              int32_t epilogue, int32_t dtype, void* stream);

/*
 * W8A8: the same fp8 weight copies multiplied with fp8 ACTIVATIONS on the fp8 matrix
 * instruction (--quantization fp8 in the engine this replaces quantises activations per
 * token too; vllm_manager.py:405-407, vllm_templates.py:75-76 pass the flag through).
 * mi355_quant_rows_fp8 turns 16-bit rows x[M, K] (K % 8 == 0) into row-major e4m3fn rows
 * xq (xq_stride bytes per row, a multiple of 8) and one fp32 scale per row with the recipe
 * of the weight copies: scale = max|row| / 448 (1 for an all-zero row), q = fp8_rne(x / scale).
 * mi355_linear_tiled_w8a8 computes
 *   y[m, n] = round(((sum_k xq[m, k] q[n, k]) * w_scale[n]) * x_scale[m] (+ bias[n]))
 * with fp32 accumulation; dtype is the type of y and bias.  M <= 128; xq 16-byte aligned with
 * xq_stride a multiple of 16; tiles_per_wg 208 / 210 / 212 / 214 / 216: every wave streams two
 * adjacent weight tiles, 4 (narrow x slices) / 5 / 6 / 7 / 8 waves per workgroup; x slice = 1024
 * (208: 512) columns for M <= 32, halved for M <= 64 and again for M <= 128, K a multiple of
 * it.  Built geometries: M <= 32: 208, 216; M <= 64: all five; M <= 128: 210 .. 216.
 * Epilogues 0-2 and k_groups as mi355_linear_tiled (the slabs of epilogue 2 carry both scales).
 * Not used by the decode step of the engine, which keeps 16-bit activations.
 */
int32_t mi355_quant_rows_fp8(void* xq, int64_t xq_stride, float* scale, const void* x,
                             int64_t x_stride, int32_t M, int32_t K, int32_t dtype,
                             void* stream);
int32_t mi355_linear_tiled_w8a8(void* y, int64_t y_stride, void* partial, const void* xq,
                                int64_t xq_stride, const float* x_scale,
                                const void* w_tiled_fp8, const float* w_scale,
                                const void* bias, int32_t M, int32_t N, int32_t K,
                                int32_t tiles_per_wg, int32_t k_groups, int32_t epilogue,
                                int32_t dtype, void* stream);

/*
 * One-shot all-reduce (sum, in place) of the small 16-bit messages of a tensor-parallel decode
 * step: the [tokens, hidden] partial sums after the attention output projection (and after the
 * down projection when the MLP is sharded), one process per GPU.  Every rank writes its message
 * into a buffer of every peer (mapped through HIP IPC), flags it, waits for the peers' flags and
 * adds the world messages in rank order in fp32 -- one kernel per call, no host round trip,
 * results bit-identical on all ranks, capturable into a HIP graph.  Replaces what the engine
 * image does with its custom all-reduce for these sizes (the harness switches that one off:
 * --disable-custom-all-reduce, benchmark-vllm/vllm_templates.py:62, VLLM_DISABLE_CUSTOM_ALL_REDUCE
 * :157-160, so the collective is this engine's own business).
 *   init     allocates this rank's buffer (uncached device memory: 2 * world * max_bytes + flags)
 *            and writes its IPC handle (mi355_allreduce_oneshot_handle_bytes() bytes) to
 *            handle_out; the caller gathers the handles of all ranks (any transport)
 *   connect  maps the peers' buffers; all_handles = world handles in rank order
 *   run      data: count 16-bit elements on this rank's GPU (count % 8 == 0, count * 2 <=
 *            max_bytes), summed over the ranks in place.  Every rank must issue the same calls
 *            with the same counts in the same order.  A peer that does not arrive within 10 s
 *            makes the kernel give up and leave data as it was:
 *   run_slabs  the same, but this rank's message is round(slabs[0] + ... + slabs[num_slabs - 1]):
 *            fp32 partial sums [num_slabs][count] of a K-split mi355_linear_tiled (epilogue 2),
 *            added in slab order and rounded once on the way into the peers' buffers -- what
 *            mi355_sum_slabs followed by run computes, without the launch in between.  `out`
 *            (count 16-bit elements) receives the sum over the ranks; on a timeout it is left as it was
 *   run_add_norm  the all-reduce of an o_proj / down_proj result together with the kernel that
 *            follows it in a decoder layer: the message is [rows][hidden] (data: this rank's 16-bit
 *            partial rows, contiguous; or slabs as in run_slabs, data then null), a workgroup owns a
 *            row, sums it over the ranks and goes on with residual = round(round(sum) + residual),
 *            out = round(residual * inv_rms) * norm_weight -- bit-identical to run (or run_slabs)
 *            followed by mi355_fused_add_rms_norm, one launch instead of two; the summed projection
 *            itself is not stored.  rows <= 256, hidden <= 8192.  On a timeout residual and out are
 *            left as they were
 *   mi355_allreduce_twoshot_run_add_norm  the same call (same communicator, same arguments, bit-identical
 *            results) in two hops for messages whose bytes set the time: the rows are dealt to the ranks in
 *            contiguous slices, every rank sends each row only to the rank that owns it, the owner sums the
 *            world rows in rank order and sends the finished 16-bit row to everybody -- 2 (world - 1) / world
 *            message sizes over a rank's links instead of world - 1, one flag wait more.  Every rank must
 *            choose the same form for the same call
 *   run_max_keys  the greedy pick of a step whose vocabulary is split over the ranks: keys[rows] are the
 *            (logit, index) keys of this rank's slice (mi355_argmax_shard_key); they are max-reduced
 *            over the ranks (signed 64-bit) and the winners' token ids written to out_ids (and to
 *            out_ids2 unless null) -- the process group's MAX all-reduce and mi355_argmax_key_to_index
 *            in one single-workgroup launch.  rows <= 256
 *   status   synchronises with the queued work and returns MI355_ERR_LAUNCH if that happened
 *   status_async  queues a copy of the communicator's error word (0: every wait so far was met;
 *            1 + r: rank r once failed to arrive) to *host_word (pinned host memory) on the
 *            stream: a serving loop reads it with the step's other results, no extra
 *            synchronisation, and stops serving in the step that hit it
 *   destroy  unmaps and frees
 * world <= 8, max_bytes <= 16 MiB.  Larger (prefill-sized) messages belong on RCCL.
 */
int64_t mi355_allreduce_oneshot_handle_bytes(void);
int32_t mi355_allreduce_oneshot_init(void** comm_out, int32_t rank, int32_t world,
                                     int64_t max_bytes, void* handle_out);
int32_t mi355_allreduce_oneshot_connect(void* comm, const void* all_handles);
int32_t mi355_allreduce_oneshot_run(void* comm, void* data, int64_t count,
                                    int32_t dtype, void* stream);
int32_t mi355_allreduce_oneshot_run_slabs(void* comm, void* out, const float* slabs,
                                          int32_t num_slabs, int64_t count, int32_t dtype,
                                          void* stream);
int32_t mi355_allreduce_oneshot_run_add_norm(void* comm, const void* data, const float* slabs,
                                             int32_t num_slabs, void* residual,
                                             int64_t residual_stride, const void* norm_weight,
                                             void* out, int64_t out_stride, int32_t rows,
                                             int32_t hidden, float eps, int32_t dtype, void* stream);
int32_t mi355_allreduce_twoshot_run_add_norm(void* comm, const void* data, const float* slabs,
                                             int32_t num_slabs, void* residual,
                                             int64_t residual_stride, const void* norm_weight,
                                             void* out, int64_t out_stride, int32_t rows,
                                             int32_t hidden, float eps, int32_t dtype, void* stream);
int32_t mi355_allreduce_oneshot_run_max_keys(void* comm, const int64_t* keys, int64_t* out_ids,
                                             int64_t* out_ids2, int32_t rows, void* stream);
int32_t mi355_allreduce_oneshot_status(void* comm);
int32_t mi355_allreduce_oneshot_status_async(void* comm, uint32_t* host_word, void* stream);
int32_t mi355_allreduce_oneshot_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* MI355_DECODE_H */
