/*
 * mi355_decode_cpu.h -- the CPU twin of the operator ABI in mi355_decode.h
 * (libmi355_decode_cpu.so, plain C++, no GPU, no HIP): mi355_cpu_<op> takes the arguments of
 * mi355_<op> with HOST pointers and without the stream (and, for decode attention, without the
 * split / workspace arguments: one pass on the CPU), rounds at the same points, and returns the
 * same status codes with the reason in mi355_cpu_last_error().
 *
 * What it is for: the argument contract of the ABI can be exercised where there is no GPU and
 * under AddressSanitizer / UBSan (tests/test_cpu_twin.py builds it with
 * -fsanitize=address,undefined), and bench.py times a decode step built from it on the GPU
 * box's host cores as the CPU baseline -- the reference's own CPU path is the stock engine's CPU
 * backend behind the same flags (--device cpu, benchmark-vllm/vllm_templates.py:60; image
 * vllm-cpu-env:latest, :124), which is not installable here.
 *
 * What it is NOT: a fallback.  The serving engine (mi355_decode) never loads this library; it
 * raises at import when libmi355_decode.so is missing.
 *
 * Layouts, dtypes (MI355_BF16 / MI355_F16 / MI355_F32 for logits) and semantics: mi355_decode.h.
 */
#ifndef MI355_DECODE_CPU_H
#define MI355_DECODE_CPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* mi355_cpu_last_error(void);

/* twin of mi355_rms_norm */
int32_t mi355_cpu_rms_norm(void* out, const void* x, const void* weight, int64_t num_tokens,
                           int32_t hidden, int64_t x_stride, int64_t out_stride, float eps,
                           int32_t dtype);

/* twin of mi355_fused_add_rms_norm */
int32_t mi355_cpu_fused_add_rms_norm(void* out, const void* x, void* residual, const void* weight,
                                     int64_t num_tokens, int32_t hidden, int64_t x_stride,
                                     int64_t out_stride, int64_t residual_stride, float eps,
                                     int32_t dtype);

/* twin of mi355_layer_norm */
int32_t mi355_cpu_layer_norm(void* out, const void* x, const void* weight, const void* bias, int64_t num_tokens,
                             int32_t hidden, int64_t x_stride, int64_t out_stride, float eps, int32_t dtype);

/* twin of mi355_fused_add_layer_norm */
int32_t mi355_cpu_fused_add_layer_norm(void* out, const void* x, void* residual, const void* weight,
                                       const void* bias, int64_t num_tokens, int32_t hidden, int64_t x_stride,
                                       int64_t out_stride, int64_t residual_stride, float eps, int32_t dtype);

/* twin of mi355_relu */
int32_t mi355_cpu_relu(void* out, const void* x, int64_t num_elements, int32_t dtype);

/* twin of mi355_rope_kv_write */
int32_t mi355_cpu_rope_kv_write(void* qkv, int64_t qkv_stride, const int64_t* positions,
                                const void* cos_sin, int32_t max_position,
                                const int64_t* slot_mapping, void* k_cache, void* v_cache,
                                int32_t num_blocks, int64_t num_tokens, int32_t num_heads,
                                int32_t num_kv_heads, int32_t head_dim, int32_t dtype);

/* twin of mi355_paged_attn_decode (no context splits on the CPU) */
int32_t mi355_cpu_paged_attn_decode(void* out, const void* q, int64_t q_stride, const void* k_cache,
                                    const void* v_cache, int32_t num_blocks,
                                    const int32_t* block_tables, int32_t max_blocks_per_seq,
                                    const int32_t* context_lens, int32_t num_seqs,
                                    int32_t num_heads, int32_t num_kv_heads, int32_t head_dim,
                                    float scale, int32_t dtype);

/* twin of mi355_paged_attn_prefill */
int32_t mi355_cpu_paged_attn_prefill(void* out, int64_t out_stride, const void* q, int64_t q_stride,
                                     const void* k_cache, const void* v_cache, int32_t num_blocks,
                                     const int32_t* block_tables, int32_t max_blocks_per_seq,
                                     const int32_t* cu_seqlens, const int32_t* context_lens,
                                     int32_t num_seqs, int32_t max_q_len, int32_t num_heads,
                                     int32_t num_kv_heads, int32_t head_dim, float scale,
                                     int32_t dtype);

/* twin of mi355_silu_and_mul */
int32_t mi355_cpu_silu_and_mul(void* out, const void* x, int64_t x_stride, int64_t num_tokens,
                               int32_t inter, int32_t dtype);

/* twin of mi355_argmax */
int32_t mi355_cpu_argmax(int64_t* out_ids, const void* logits, int64_t logits_stride,
                         int32_t num_rows, int32_t vocab, int32_t logits_dtype);

/* twin of mi355_linear_decode: y = round(x . W^T + bias), W row-major [N, K], fp32 accumulation.
 * Any M (the GPU entry point stops at 128 rows).  Threads: OpenMP, OMP_NUM_THREADS. */
int32_t mi355_cpu_linear(void* y, int64_t y_stride, const void* x, int64_t x_stride, const void* weight,
                         const void* bias, int32_t M, int32_t N, int32_t K, int32_t dtype);

#ifdef __cplusplus
}
#endif
#endif /* MI355_DECODE_CPU_H */
