// Rotary embedding (rotate-half pairing) on the q and k parts of the packed QKV
// rows, fused with the paged KV-cache store.
//
// Work item = (token, head, chunk): 8 rotation pairs, i.e. elements
// [8c, 8c+8) of the first half and the matching 8 of the second half of one
// head.  q heads are rotated in place; k heads are rotated in place and stored
// to the K page (token rows, two 16-byte stores); v heads are copied to the V
// page, which keeps tokens along the fastest axis (16 two-byte stores at a
// 32-byte stride -- the page is laid out for the PV MFMA operand, see
// paged_attn_decode.hip).  All loads are 16-byte vectors; cos/sin come from a
// precomputed table in the model dtype.  HBM-bound and small: 2 bytes in, 2 out
// per q element, 2 in / 4 out per k element, 2 in / 2 out per v element.
//
// Prefill-sized calls (>= 64 tokens, plain form) store V with a second kernel instead
// (v_pages_kernel): a wave takes 16 consecutive packed tokens of one kv head, and when
// they fill one page in order -- the usual case, a prompt's tokens go to consecutive
// slots -- it transposes the 16 x head_dim tile through LDS and writes the page as
// one contiguous 32 * head_dim bytes (16-byte stores) instead of 16 * head_dim two-byte
// stores at a 32-byte stride, which the cache then has to merge sector by sector.
// Any other group (page boundary inside, padding slots, a chunk that starts mid-page)
// takes the element-wise stores.
//
// The PART form reads the projection from the fp32 partial-sum slabs of a
// K-split GEMM ([slab][token][width]) instead: qkv = round(slab_0 + slab_1 + ...
// + bias) is formed on the fly, and q, k (rotated) and v are all written to the
// packed 16-bit rows, which then hold what the plain form leaves there.
#include "common.h"

namespace mi355 {

template <typename T, bool PART>
__global__ __launch_bounds__(256) void rope_kv_write_kernel(
    uint16_t* __restrict__ qkv, int64_t qkv_stride, const int64_t* __restrict__ positions,
    const uint16_t* __restrict__ cos_sin, int max_position, const int64_t* __restrict__ slot_mapping,
    uint16_t* __restrict__ k_cache, uint16_t* __restrict__ v_cache, int num_blocks, int64_t num_tokens,
    int num_heads, int num_kv_heads, int head_dim, const float* __restrict__ partial, int num_slabs,
    int64_t slab_stride, const uint16_t* __restrict__ bias, int skip_v) {
  const int half = head_dim >> 1;
  const int chunks = half >> 3;  // 8 pairs per chunk
  const int heads_total = num_heads + 2 * num_kv_heads;
  const int64_t items = num_tokens * heads_total * chunks;
  constexpr int BS = MI355_BLOCK_SIZE;

  for (int64_t it = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; it < items;
       it += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int c = static_cast<int>(it % chunks);
    const int head = static_cast<int>((it / chunks) % heads_total);
    const int64_t tok = it / (static_cast<int64_t>(chunks) * heads_total);
    if (skip_v && head >= num_heads + num_kv_heads) continue;   // v_pages_kernel stores the V heads

    uint16_t* base = qkv + tok * qkv_stride + static_cast<int64_t>(head) * head_dim;
    uint4* p1 = reinterpret_cast<uint4*>(base + 8 * c);
    uint4* p2 = reinterpret_cast<uint4*>(base + half + 8 * c);
    uint4 r1, r2;
    if (PART) {
      const int col = head * head_dim + 8 * c;
      const float* pr = partial + tok * (static_cast<int64_t>(heads_total) * head_dim) + col;
      float f1[8], f2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f1[j] = f2[j] = 0.f;
      for (int s = 0; s < num_slabs; ++s) {
        const float4* a = reinterpret_cast<const float4*>(pr + s * slab_stride);
        const float4* b = reinterpret_cast<const float4*>(pr + s * slab_stride + half);
        const float4 a0 = MI355_LD(float4, partial, a), a1 = MI355_LD(float4, partial, a + 1),
                     b0 = MI355_LD(float4, partial, b), b1 = MI355_LD(float4, partial, b + 1);
        f1[0] += a0.x; f1[1] += a0.y; f1[2] += a0.z; f1[3] += a0.w;
        f1[4] += a1.x; f1[5] += a1.y; f1[6] += a1.z; f1[7] += a1.w;
        f2[0] += b0.x; f2[1] += b0.y; f2[2] += b0.z; f2[3] += b0.w;
        f2[4] += b1.x; f2[5] += b1.y; f2[6] += b1.z; f2[7] += b1.w;
      }
      if (bias != nullptr) {
        float b1[8], b2[8];
        unpack8<T>(MI355_LD(uint4, bias, bias + col), b1);
        unpack8<T>(MI355_LD(uint4, bias, bias + col + half), b2);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          f1[j] += b1[j];
          f2[j] += b2[j];
        }
      }
      r1 = pack8<T>(f1);
      r2 = pack8<T>(f2);
    } else {
      r1 = MI355_LD(uint4, qkv, p1);
      r2 = MI355_LD(uint4, qkv, p2);
    }

    const bool is_v = head >= num_heads + num_kv_heads;
    if (!is_v) {
      int64_t pos = MI355_LD(int64_t, positions, positions + tok);
      pos = pos < 0 ? 0 : (pos >= max_position ? max_position - 1 : pos);
      const uint16_t* cs = cos_sin + pos * head_dim;
      float x1[8], x2[8], co[8], si[8];
      unpack8<T>(r1, x1);
      unpack8<T>(r2, x2);
      unpack8<T>(MI355_LD(uint4, cos_sin, cs + 8 * c), co);
      unpack8<T>(MI355_LD(uint4, cos_sin, cs + half + 8 * c), si);
      float o1[8], o2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // Each product and the sum are rounded to the model dtype.  Contraction
        // must stay off here: with fp16 the compiler lowers these to half
        // arithmetic and would fuse mul + sub into one fma, dropping a rounding.
#pragma clang fp contract(off)
        o1[j] = round_f<T>(x1[j] * co[j]) - round_f<T>(x2[j] * si[j]);
        o2[j] = round_f<T>(x2[j] * co[j]) + round_f<T>(x1[j] * si[j]);
      }
      r1 = pack8<T>(o1);
      r2 = pack8<T>(o2);
      MI355_ST(uint4, qkv, p1, r1);
      MI355_ST(uint4, qkv, p2, r2);
    } else if (PART) {
      MI355_ST(uint4, qkv, p1, r1);
      MI355_ST(uint4, qkv, p2, r2);
    }
    if (head < num_heads) continue;

    const int64_t slot = MI355_LD(int64_t, slot_mapping, slot_mapping + tok);
    if (slot < 0 || slot >= static_cast<int64_t>(num_blocks) * BS) continue;
    const int64_t blk = slot / BS;
    const int off = static_cast<int>(slot % BS);
    if (!is_v) {
      const int kvh = head - num_heads;
      uint16_t* dst = k_cache + ((blk * num_kv_heads + kvh) * BS + off) * head_dim;
      MI355_ST(uint4, k_cache, dst + 8 * c, r1);
      MI355_ST(uint4, k_cache, dst + half + 8 * c, r2);
    } else {
      const int kvh = head - num_heads - num_kv_heads;
      uint16_t* dst = v_cache + (blk * num_kv_heads + kvh) * head_dim * BS + off;
      const uint32_t w1[4] = {r1.x, r1.y, r1.z, r1.w};
      const uint32_t w2[4] = {r2.x, r2.y, r2.z, r2.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint16_t e1 = static_cast<uint16_t>(j & 1 ? w1[j >> 1] >> 16 : w1[j >> 1] & 0xFFFFu);
        const uint16_t e2 = static_cast<uint16_t>(j & 1 ? w2[j >> 1] >> 16 : w2[j >> 1] & 0xFFFFu);
        MI355_ST(uint16_t, v_cache, dst + static_cast<int64_t>(8 * c + j) * BS, e1);
        MI355_ST(uint16_t, v_cache, dst + static_cast<int64_t>(half + 8 * c + j) * BS, e2);
      }
    }
  }
}

// V heads of a prefill-sized call: one wave per (16 packed tokens, kv head).
constexpr int kVPitch = 48;   // LDS bytes per head_dim row of the transposed tile: 16 tokens x 2 B + padding, 16-byte aligned

__global__ __launch_bounds__(256) void v_pages_kernel(const uint16_t* __restrict__ qkv, int64_t qkv_stride,
                                                      const int64_t* __restrict__ slot_mapping,
                                                      uint16_t* __restrict__ v_cache, int num_blocks, int64_t num_tokens,
                                                      int num_heads, int num_kv_heads, int head_dim) {
  constexpr int BS = MI355_BLOCK_SIZE;
  __shared__ __attribute__((aligned(16))) unsigned char lds_all[4 * 256 * kVPitch];   // per wave: head_dim (<= 256) rows
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  unsigned char* lds = lds_all + wave * (256 * kVPitch);
  const int t = lane & 15, p0 = lane >> 4;
  const int pieces = head_dim >> 3;                       // 16-byte pieces per token row
  const int64_t groups = (num_tokens + 15) >> 4;
  const int64_t tiles = groups * num_kv_heads;
  const int64_t slots_total = static_cast<int64_t>(num_blocks) * BS;
  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * 4 + wave; tile < tiles; tile += static_cast<int64_t>(gridDim.x) * 4) {
    const int kvh = static_cast<int>(tile % num_kv_heads);
    const int64_t tok = (tile / num_kv_heads) * 16 + t;
    const bool valid = tok < num_tokens;
    const int64_t slot = valid ? MI355_LD(int64_t, slot_mapping, slot_mapping + tok) : -1;
    const int64_t slot0 = __shfl(slot, 0, 64);
    const bool in_order = valid && slot == slot0 + t;
    const bool full = __all(in_order) && slot0 >= 0 && (slot0 % BS) == 0 && slot0 + BS <= slots_total;   // wave-uniform
    const uint16_t* src = qkv + (valid ? tok : 0) * qkv_stride + static_cast<int64_t>(num_heads + num_kv_heads + kvh) * head_dim;
    if (full) {
      for (int piece = p0; piece < pieces; piece += 4) {
        const uint4 v = MI355_LD(uint4, qkv, src + 8 * piece);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 8; ++e)
          *reinterpret_cast<uint16_t*>(lds + (8 * piece + e) * kVPitch + 2 * t) =
              static_cast<uint16_t>(e & 1 ? w[e >> 1] >> 16 : w[e >> 1] & 0xFFFFu);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // the page of this kv head: head_dim rows of 16 tokens, 32 * head_dim contiguous bytes
      uint16_t* page = v_cache + ((slot0 / BS) * num_kv_heads + kvh) * static_cast<int64_t>(head_dim) * BS;
      for (int q = lane; q < 2 * head_dim; q += 64) {    // 16-byte piece q: row q / 2, tokens 8 * (q % 2) .. + 8
        const uint4 v = *reinterpret_cast<const uint4*>(lds + (q >> 1) * kVPitch + (q & 1) * 16);
        MI355_ST(uint4, v_cache, page + static_cast<int64_t>(q) * 8, v);
      }
      __builtin_amdgcn_wave_barrier();                   // the next tile of this wave overwrites the LDS rows
    } else if (valid && slot >= 0 && slot < slots_total) {
      uint16_t* dst = v_cache + ((slot / BS) * num_kv_heads + kvh) * static_cast<int64_t>(head_dim) * BS + (slot % BS);
      for (int piece = p0; piece < pieces; piece += 4) {
        const uint4 v = MI355_LD(uint4, qkv, src + 8 * piece);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 8; ++e)
          MI355_ST(uint16_t, v_cache, dst + static_cast<int64_t>(8 * piece + e) * BS,
                   static_cast<uint16_t>(e & 1 ? w[e >> 1] >> 16 : w[e >> 1] & 0xFFFFu));
      }
    }
  }
}

template <bool PART>
static int32_t launch_rope(const char* name, void* qkv, int64_t qkv_stride, const int64_t* positions,
                           const void* cos_sin, int32_t max_position, const int64_t* slot_mapping, void* k_cache,
                           void* v_cache, int32_t num_blocks, int64_t num_tokens, int32_t num_heads,
                           int32_t num_kv_heads, int32_t head_dim, int32_t dtype, void* stream,
                           const float* partial, int32_t num_slabs, int64_t slab_stride, const void* bias) {
  MI355_REQUIRE(qkv && positions && cos_sin && slot_mapping && k_cache && v_cache, MI355_ERR_INVALID,
                "%s: null pointer", name);
  MI355_REQUIRE(num_tokens >= 0 && num_tokens <= 0x7FFFFFFF, MI355_ERR_INVALID,
                "%s: num_tokens %lld out of range", name, (long long)num_tokens);
  MI355_REQUIRE(head_dim > 0 && head_dim % 16 == 0 && head_dim <= 256, MI355_ERR_UNSUPPORTED,
                "%s: head_dim %d must be a multiple of 16, at most 256", name, head_dim);
  MI355_REQUIRE(num_heads > 0 && num_kv_heads > 0 && num_blocks > 0 && max_position > 0, MI355_ERR_INVALID,
                "%s: num_heads/num_kv_heads/num_blocks/max_position must be positive", name);
  const int64_t width = static_cast<int64_t>(num_heads + 2 * num_kv_heads) * head_dim;
  MI355_REQUIRE(qkv_stride >= width && qkv_stride % 8 == 0, MI355_ERR_INVALID,
                "%s: qkv_stride %lld must be >= %lld and a multiple of 8", name, (long long)qkv_stride,
                (long long)width);
  MI355_REQUIRE(aligned16(qkv) && aligned16(cos_sin) && aligned16(k_cache) && aligned16(v_cache),
                MI355_ERR_INVALID, "%s: pointers must be 16-byte aligned", name);
  MI355_REQUIRE(dtype == MI355_BF16 || dtype == MI355_F16, MI355_ERR_UNSUPPORTED,
                "%s: dtype %d not supported (bf16/fp16 only)", name, dtype);
  if (PART) {
    MI355_REQUIRE(partial && aligned16(partial) && (bias == nullptr || aligned16(bias)), MI355_ERR_INVALID,
                  "%s: partial/bias must be non-null (partial) and 16-byte aligned", name);
    MI355_REQUIRE(num_slabs >= 1 && num_slabs <= 64, MI355_ERR_INVALID, "%s: num_slabs %d outside [1, 64]", name,
                  num_slabs);
    MI355_REQUIRE(slab_stride >= num_tokens * width && slab_stride % 4 == 0, MI355_ERR_INVALID,
                  "%s: slab_stride %lld must be >= num_tokens * width and a multiple of 4", name,
                  (long long)slab_stride);
  }
  if (num_tokens == 0) return MI355_OK;

  const int64_t items = num_tokens * (num_heads + 2 * num_kv_heads) * (head_dim / 16);
  int64_t blocks = (items + 255) / 256;
  if (blocks > 8192) blocks = 8192;  // grid-stride beyond 32 workgroups per CU
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 grid(static_cast<unsigned>(blocks)), block(256);
  const int skip_v = (!PART && num_tokens >= 64) ? 1 : 0;   // prefill-sized: the V heads go through v_pages_kernel
  if (dtype == MI355_BF16)
    hipLaunchKernelGGL((rope_kv_write_kernel<BF16, PART>), grid, block, 0, s, static_cast<uint16_t*>(qkv),
                       qkv_stride, positions, static_cast<const uint16_t*>(cos_sin), max_position,
                       slot_mapping, static_cast<uint16_t*>(k_cache), static_cast<uint16_t*>(v_cache),
                       num_blocks, num_tokens, num_heads, num_kv_heads, head_dim, partial, num_slabs,
                       slab_stride, static_cast<const uint16_t*>(bias), skip_v);
  else
    hipLaunchKernelGGL((rope_kv_write_kernel<F16, PART>), grid, block, 0, s, static_cast<uint16_t*>(qkv),
                       qkv_stride, positions, static_cast<const uint16_t*>(cos_sin), max_position,
                       slot_mapping, static_cast<uint16_t*>(k_cache), static_cast<uint16_t*>(v_cache),
                       num_blocks, num_tokens, num_heads, num_kv_heads, head_dim, partial, num_slabs,
                       slab_stride, static_cast<const uint16_t*>(bias), skip_v);
  if (skip_v) {
    const int64_t tiles = ((num_tokens + 15) / 16) * num_kv_heads;
    int64_t vblocks = (tiles + 3) / 4;
    if (vblocks > 8192) vblocks = 8192;
    hipLaunchKernelGGL(v_pages_kernel, dim3(static_cast<unsigned>(vblocks)), dim3(256), 0, s,
                       static_cast<const uint16_t*>(qkv), qkv_stride, slot_mapping, static_cast<uint16_t*>(v_cache),
                       num_blocks, num_tokens, num_heads, num_kv_heads, head_dim);
  }
  return check_launch(name);
}

}  // namespace mi355

extern "C" int32_t mi355_rope_kv_write(void* qkv, int64_t qkv_stride, const int64_t* positions,
                                       const void* cos_sin, int32_t max_position,
                                       const int64_t* slot_mapping, void* k_cache, void* v_cache,
                                       int32_t num_blocks, int64_t num_tokens, int32_t num_heads,
                                       int32_t num_kv_heads, int32_t head_dim, int32_t dtype,
                                       void* stream) {
  return mi355::launch_rope<false>("mi355_rope_kv_write", qkv, qkv_stride, positions, cos_sin, max_position,
                                   slot_mapping, k_cache, v_cache, num_blocks, num_tokens, num_heads,
                                   num_kv_heads, head_dim, dtype, stream, nullptr, 0, 0, nullptr);
}

extern "C" int32_t mi355_rope_kv_write_partial(void* qkv, int64_t qkv_stride, const float* partial,
                                               int32_t num_slabs, int64_t slab_stride, const void* bias,
                                               const int64_t* positions, const void* cos_sin,
                                               int32_t max_position, const int64_t* slot_mapping, void* k_cache,
                                               void* v_cache, int32_t num_blocks, int64_t num_tokens,
                                               int32_t num_heads, int32_t num_kv_heads, int32_t head_dim,
                                               int32_t dtype, void* stream) {
  return mi355::launch_rope<true>("mi355_rope_kv_write_partial", qkv, qkv_stride, positions, cos_sin,
                                  max_position, slot_mapping, k_cache, v_cache, num_blocks, num_tokens, num_heads,
                                  num_kv_heads, head_dim, dtype, stream, partial, num_slabs, slab_stride, bias);
}
