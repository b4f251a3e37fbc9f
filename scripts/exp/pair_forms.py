"""EXPERIMENT (Llama-3-8B shapes, 32 rows, bf16): a K-split projection together with the kernel that consumes
its output, in the slab form the decode step runs and in the form with K split across the WAVES of a workgroup
(linear_tiled_kw_kernel) that stores one 16-bit (or one fp32-slab) result:
  o / down + add + norm:  slabs -> rms_norm_slabs   vs   16-bit store -> fused_add_rms_norm
  qkv + fused attention:  2/4 slabs summed in the attention prologue   vs   one slab from the kw kernel
us per pair from a HIP graph walking 16 weight copies (nothing is cache-warm)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vllm-backend-public_amd"))
import numpy as np
import torch
from mi355_decode import ops

dev, dt = "cuda", torch.bfloat16
M, H, F, C = 32, 4096, 14336, 16
eps = 1e-5


def graph_us(fn, n=C, reps=5):
    fn(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n):
            fn(i)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); e.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / n)
    return round(best, 2)


res = {}
normw = torch.ones(H, dtype=dt, device=dev)
resid = torch.zeros(M, H, dtype=dt, device=dev)
xo = torch.empty(M, H, dtype=dt, device=dev)
for kind, K in (("o", H), ("down", F)):
    x = torch.randn(M, K, dtype=dt, device=dev) * 0.1
    wt = [ops.tile_weights(torch.empty(H, K, dtype=dt, device=dev).normal_(0, 0.02)) for _ in range(C)]
    y = torch.empty(M, H, dtype=dt, device=dev)
    out = {}
    for tiles, groups in ((104, 4), (104, 8), (4, 4), (8, 4), (8, 8), (302, 2), (304, 2), (304, 4), (308, 2)):
        part = torch.empty(groups, M, H, dtype=torch.float32, device=dev)
        try:
            def pair(i):
                ops.linear_tiled(x, wt[i], H, partial=part, tiles_per_wg=tiles, k_groups=groups, epilogue=ops.EPI_PARTIAL)
                ops.fused_add_rms_norm_partial(part, groups, resid, normw, eps, out=xo)
            out[f"slabs {tiles}/{groups} + norm"] = graph_us(pair)
            out[f"slabs {tiles}/{groups} alone"] = graph_us(lambda i: ops.linear_tiled(
                x, wt[i], H, partial=part, tiles_per_wg=tiles, k_groups=groups, epilogue=ops.EPI_PARTIAL))
        except Exception as exc:
            out[f"slabs {tiles}/{groups}"] = repr(exc)[:80]
        resid.zero_()
    for tiles in (302, 304, 308, 402, 404):
        try:
            def pair16(i):
                ops.linear_tiled(x, wt[i], H, out=y, tiles_per_wg=tiles, k_groups=1)
                ops.fused_add_rms_norm(y, resid, normw, eps, out=xo)
            out[f"16-bit store {tiles}/1 + add + norm"] = graph_us(pair16)
            out[f"16-bit store {tiles}/1 alone"] = graph_us(lambda i: ops.linear_tiled(x, wt[i], H, out=y, tiles_per_wg=tiles, k_groups=1))
        except Exception as exc:
            out[f"16-bit store {tiles}/1"] = repr(exc)[:80]
        resid.zero_()
    res[kind] = out
    del wt
    print(json.dumps({kind: out}), flush=True)

# qkv + fused attention at context 160
nh, nkv, d, ctx_len, bs = 32, 8, 128, 160, 16
qw = (nh + 2 * nkv) * d
x = torch.randn(M, H, dtype=dt, device=dev) * 0.1
wt = [ops.tile_weights(torch.empty(qw, H, dtype=dt, device=dev).normal_(0, 0.02)) for _ in range(C)]
blocks_per = (ctx_len + bs) // bs + 1
kc = [torch.zeros(M * blocks_per, nkv, bs, d, dtype=dt, device=dev) for _ in range(C)]
vc = [torch.zeros(M * blocks_per, nkv, d, bs, dtype=dt, device=dev) for _ in range(C)]
bt = torch.arange(M * blocks_per, dtype=torch.int32, device=dev).reshape(M, blocks_per)
ctx = torch.full((M,), ctx_len, dtype=torch.int32, device=dev)
pos = torch.full((M,), ctx_len - 1, dtype=torch.int64, device=dev)
slots = (bt[:, (ctx_len - 1) // bs].to(torch.int64) * bs + (ctx_len - 1) % bs)
from mi355_decode.config import ModelConfig
from mi355_decode.model_runner import build_cos_sin_table
cos_sin = build_cos_sin_table(ModelConfig.from_pretrained("llama-3-8b"), 512, dt, dev)
attn = torch.empty(M, nh * d, dtype=dt, device=dev)
import inspect
sig = inspect.signature(ops.paged_attn_decode_fused)
out = {"attention_signature": str(sig)[:400]}
for tiles, groups in ((8, 2), (8, 4), (4, 2), (104, 4), (302, 1), (304, 1), (308, 1), (304, 2), (402, 1), (404, 1)):
    part = torch.empty(groups, M, qw, dtype=torch.float32, device=dev)
    try:
        def pair(i):
            ops.linear_tiled(x, wt[i], qw, partial=part, tiles_per_wg=tiles, k_groups=groups, epilogue=ops.EPI_PARTIAL)
            ops.paged_attn_decode_fused(None, cos_sin, slots, kc[i], vc[i], bt, ctx, nh, nkv, d, d ** -0.5, num_splits=1,
                                        out=attn, partial=part, num_slabs=groups)
        out[f"qkv {tiles}/{groups} + attention"] = graph_us(pair)
        out[f"qkv {tiles}/{groups} alone"] = graph_us(lambda i: ops.linear_tiled(
            x, wt[i], qw, partial=part, tiles_per_wg=tiles, k_groups=groups, epilogue=ops.EPI_PARTIAL))
    except Exception as exc:
        out[f"qkv {tiles}/{groups}"] = repr(exc)[:200]
qkv16 = torch.empty(M, qw, dtype=dt, device=dev)
for tiles in (302, 304, 308, 402, 404):
    try:
        def pair16(i):
            ops.linear_tiled(x, wt[i], qw, out=qkv16, tiles_per_wg=tiles, k_groups=1)
            ops.paged_attn_decode_fused(qkv16, cos_sin, slots, kc[i], vc[i], bt, ctx, nh, nkv, d, d ** -0.5, num_splits=1, out=attn)
        out[f"qkv 16-bit store {tiles}/1 + attention"] = graph_us(pair16)
    except Exception as exc:
        out[f"qkv 16-bit store {tiles}/1"] = repr(exc)[:200]
res["qkv"] = out
print(json.dumps({"qkv": out}), flush=True)
