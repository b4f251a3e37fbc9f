// Sanitizer driver for the CPU twin of the operator ABI: every entry point on exactly-sized heap
// buffers (so that AddressSanitizer sees any access past an end), ragged and edge shapes, and the
// rejected-argument paths.  Built by tests/test_cpu_twin.py with -fsanitize=address,undefined;
// exit status 0 = clean.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi355_decode.h"
#include "mi355_decode_cpu.h"

static uint16_t* buf16(size_t n, uint16_t fill) {
  void* p = nullptr;
  if (posix_memalign(&p, 16, n * 2 ? n * 2 : 16)) abort();      // exactly n elements: one past the end is poisoned
  uint16_t* q = static_cast<uint16_t*>(p);
  for (size_t i = 0; i < n; ++i) q[i] = static_cast<uint16_t>(fill + 37 * i) & 0x3FFFu ? static_cast<uint16_t>(0x3C00u + (i % 251)) : fill;
  return q;
}
#define CHECK(call)                                                                 \
  do {                                                                              \
    const int32_t rc_ = (call);                                                     \
    if (rc_ != MI355_OK) {                                                          \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mi355_cpu_last_error());        \
      return 1;                                                                     \
    }                                                                               \
  } while (0)
#define REJECT(call)                                                                \
  do {                                                                              \
    if ((call) == MI355_OK) {                                                       \
      fprintf(stderr, "%s was accepted\n", #call);                                  \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

int main() {
  for (int dtype = 0; dtype < 2; ++dtype) {
    // ---- norms: 5 rows of 448, strided input
    {
      const int T = 5, H = 448, XS = 512;
      uint16_t *x = buf16(T * XS - (XS - H), 1), *o = buf16(T * H, 0), *w = buf16(H, 2), *r = buf16(T * H, 3);
      CHECK(mi355_cpu_rms_norm(o, x, w, T, H, XS, H, 1e-5f, dtype));
      CHECK(mi355_cpu_fused_add_rms_norm(o, x, r, w, T, H, XS, H, H, 1e-5f, dtype));
      REJECT(mi355_cpu_rms_norm(o, x, w, T, 250, XS, H, 1e-5f, dtype));
      REJECT(mi355_cpu_rms_norm(nullptr, x, w, T, H, XS, H, 1e-5f, dtype));
      free(x), free(o), free(w), free(r);
    }
    // ---- rope + KV store, decode and prefill attention: GQA 7, head_dim 64, ragged contexts
    {
      const int nh = 7, nkv = 1, d = 64, T = 9, nblk = 6, maxpos = 96, width = (nh + 2 * nkv) * d, MB = 6;
      uint16_t *qkv = buf16(T * width, 5), *cs = buf16(maxpos * d, 7), *kc = buf16(nblk * nkv * 16 * d, 0),
               *vc = buf16(nblk * nkv * d * 16, 0);
      int64_t pos[T], slots[T];
      for (int t = 0; t < T; ++t) pos[t] = (t * 13) % maxpos, slots[t] = (t == 3) ? -1 : (t * 7) % (nblk * 16);
      slots[5] = nblk * 16 + 3;                                     // beyond the pool: not stored
      pos[6] = maxpos + 50;                                         // clamped
      CHECK(mi355_cpu_rope_kv_write(qkv, width, pos, cs, maxpos, slots, kc, vc, nblk, T, nh, nkv, d, dtype));
      const int B = 4;
      int32_t bt[B * MB], ctx[B] = {1, 17, 96, 0};
      for (int i = 0; i < B * MB; ++i) bt[i] = (i * 5) % nblk;
      bt[2] = -3, bt[7] = nblk + 9;                                 // out-of-range ids are clamped
      uint16_t *q = buf16(B * nh * d, 9), *out = buf16(B * nh * d, 0);
      CHECK(mi355_cpu_paged_attn_decode(out, q, nh * d, kc, vc, nblk, bt, MB, ctx, B, nh, nkv, d, 0.125f, dtype));
      ctx[2] = 1000;                                                // beyond max_blocks * 16: clamped
      CHECK(mi355_cpu_paged_attn_decode(out, q, nh * d, kc, vc, nblk, bt, MB, ctx, B, nh, nkv, d, 0.125f, dtype));
      int32_t cu[3] = {0, 5, 9}, pctx[2] = {40, 4};
      uint16_t *pq = buf16(9 * nh * d, 11), *po = buf16(9 * nh * d, 0);
      CHECK(mi355_cpu_paged_attn_prefill(po, nh * d, pq, nh * d, kc, vc, nblk, bt, MB, cu, pctx, 2, 5, nh, nkv, d, 0.125f, dtype));
      REJECT(mi355_cpu_paged_attn_decode(out, q, nh * d, kc, vc, nblk, bt, MB, ctx, B, 34, 2, d, 0.125f, dtype));
      REJECT(mi355_cpu_paged_attn_decode(out, q, nh * d, kc, vc, nblk, bt, MB, ctx, B, nh, nkv, 96, 0.125f, dtype));
      free(qkv), free(cs), free(kc), free(vc), free(q), free(out), free(pq), free(po);
    }
    // ---- SwiGLU, linear with bias (K not a multiple of 8), argmax with a tail
    {
      const int T = 3, F = 24;
      uint16_t *x = buf16(T * 2 * F, 13), *o = buf16(T * F, 0);
      CHECK(mi355_cpu_silu_and_mul(o, x, 2 * F, T, F, dtype));
      free(x), free(o);
      const int M = 5, N = 37, K = 99;
      uint16_t *a = buf16(M * K, 15), *w = buf16(N * K, 17), *b = buf16(N, 19), *y = buf16(M * N, 0);
      CHECK(mi355_cpu_linear(y, N, a, K, w, b, M, N, K, dtype));
      CHECK(mi355_cpu_linear(y, N, a, K, w, nullptr, M, N, K, dtype));
      free(a), free(w), free(b), free(y);
      const int R = 3, V = 1003, VS = 1008;
      uint16_t* lg = buf16(R * VS - (VS - V), 21);
      int64_t ids[R];
      CHECK(mi355_cpu_argmax(ids, lg, VS, R, V, dtype));
      for (int r = 0; r < R; ++r)
        if (ids[r] < 0 || ids[r] >= V) return 2;
      free(lg);
    }
  }
  {
    const int R = 2, V = 50;
    void* p = nullptr;
    if (posix_memalign(&p, 16, sizeof(float) * (R * 52 - 2))) abort();
    float* lg = static_cast<float*>(p);
    for (int i = 0; i < R * 52 - 2; ++i) lg[i] = static_cast<float>((i * 31) % 17);
    lg[60] = NAN;
    int64_t ids[R];
    CHECK(mi355_cpu_argmax(ids, lg, 52, R, V, MI355_F32));
    if (ids[1] != 8) return 3;                                      // the NaN of row 1
    free(lg);
  }
  puts("cpu twin: sanitizer driver clean");
  return 0;
}
