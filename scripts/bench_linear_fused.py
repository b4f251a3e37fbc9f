"""Micro-benchmark of the fused GEMM launches against their unfused parts (in-graph timing, weights rotating)."""
import sys
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops

torch.manual_seed(0)
dev = "cuda"
M, H, F = 32, 4096, 14336


def timed(fn, ws, reps):
    fn(ws[0]); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            for w in ws:
                fn(w)
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1000 / (5 * reps * len(ws))


def weights(N, K, inter=False):
    copies = max(2, int(600e6 // (N * K * 2)) + 1)
    return [ops.tile_weights(torch.empty(N, K, dtype=torch.bfloat16, device=dev).normal_(0, 0.02), interleave_halves=inter)
            for _ in range(copies)]


parts = H // 64
sumsq = torch.rand(parts, 32, dtype=torch.float32, device=dev) * 64
counters = torch.zeros(parts, dtype=torch.int32, device=dev)
nw = torch.ones(H, dtype=torch.bfloat16, device=dev)
resid = torch.randn(M, H, dtype=torch.bfloat16, device=dev)
xh = torch.randn(M, H, dtype=torch.bfloat16, device=dev)

for name, N, K, groups in (("o", H, H, 4), ("down", H, F, 4)):
    ws = weights(N, K)
    reps = max(1, 64 // len(ws))
    x = torch.randn(M, K, dtype=torch.bfloat16, device=dev)
    part = torch.empty(groups, M, N, dtype=torch.float32, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t_p = timed(lambda w: ops.linear_tiled(x, w, N, partial=part, tiles_per_wg=104, k_groups=groups, epilogue=ops.EPI_PARTIAL), ws, reps)
    t_n = timed(lambda w: ops.fused_add_rms_norm_partial(part, groups, resid, nw, 1e-5, out), ws, reps)
    t_rs = timed(lambda w: ops.linear_tiled_fused(x, w, N, tiles_per_wg=104, k_groups=groups, epilogue=ops.EPI_REDUCE_STORE, partial=part, out=out, counters=counters), ws, reps)
    t_rr = timed(lambda w: ops.linear_tiled_fused(x, w, N, tiles_per_wg=104, k_groups=groups, epilogue=ops.EPI_REDUCE_RESID, partial=part, residual=resid, sumsq_out=sumsq, counters=counters), ws, reps)
    print(f"{name:8s}: partial {t_p:6.1f}us  + norm kernel {t_n:5.1f}us | reduce_store {t_rs:6.1f}us  reduce_resid {t_rr:6.1f}us", flush=True)

for name, N, K, tiles, groups, epi in (("gate_up", 2 * F, H, 8, 1, ops.EPI_SILU_MUL), ("qkv", 6144, H, 104, 2, ops.EPI_PARTIAL),
                                       ("lm_head", 128256, H, 8, 1, ops.EPI_STORE)):
    ws = weights(N, K, inter=(epi == ops.EPI_SILU_MUL))
    reps = max(1, 64 // len(ws))
    part = torch.empty(groups, M, N, dtype=torch.float32, device=dev)
    out = torch.empty(M, N // 2 if epi == ops.EPI_SILU_MUL else N, dtype=torch.bfloat16, device=dev)
    kw = dict(tiles_per_wg=tiles, k_groups=groups, epilogue=epi)
    if epi == ops.EPI_PARTIAL:
        kw["partial"] = part
    else:
        kw["out"] = out
    t_plain = timed(lambda w: ops.linear_tiled(xh, w, N, **kw), ws, reps)
    t_norm = timed(lambda w: ops.linear_tiled_fused(resid, w, N, norm_weight=nw, norm_sumsq=sumsq, norm_parts=parts, eps=1e-5, **kw), ws, reps)
    print(f"{name:8s}: plain {t_plain:6.1f}us | with norm prologue {t_norm:6.1f}us", flush=True)
