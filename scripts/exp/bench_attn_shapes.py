"""EXPERIMENT: decode attention alone at the shapes of the single-GPU benchmarks (Llama-3-8B batch 32,
Qwen2-7B batch 64) over context lengths: unfused (rotated q given), fused on packed rows, fused on
fp32 slabs; 1 and 2 KV splits.  Reports us per launch and the KV bytes per second."""
import sys
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops

dev, dt = "cuda", torch.bfloat16
L = 8


def timed(fn):
    for i in range(2):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(4):
            for i in range(L):
                fn(i)
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1000 / (5 * 4 * L)


def run(name, B, nh, nkv, d, slabs, ctxs):
    cmax = max(ctxs)
    bps = (cmax + 15) // 16
    nb = B * bps
    kc = [torch.randn(nb, nkv, 16, d, dtype=dt, device=dev) for _ in range(L)]
    vc = [torch.randn(nb, nkv, d, 16, dtype=dt, device=dev) for _ in range(L)]
    bt = torch.arange(nb, dtype=torch.int32, device=dev).view(B, bps)
    cs = torch.randn(4096, d, dtype=dt, device=dev)
    width = (nh + 2 * nkv) * d
    qkv = torch.randn(B, width, dtype=dt, device=dev)
    part = torch.randn(slabs, B, width, dtype=torch.float32, device=dev)
    out = torch.empty(B, nh * d, dtype=dt, device=dev)
    ws = torch.empty(ops.decode_workspace_bytes(B, nh, d, 16), dtype=torch.uint8, device=dev)
    for C in ctxs:
        ctx = torch.full((B,), C, dtype=torch.int32, device=dev)
        pos = (ctx - 1).to(torch.int64)
        slots = bt[torch.arange(B, device=dev), (pos // 16)].to(torch.int64) * 16 + pos % 16
        kvb = B * C * nkv * d * 2 * 2
        for sp in (1, 2):
            t0 = timed(lambda i: ops.paged_attn_decode(qkv, kc[i], vc[i], bt, ctx, nh, nkv, d, 0.088, num_splits=sp,
                                                       workspace=ws, out=out))
            t1 = timed(lambda i: ops.paged_attn_decode_fused(qkv, cs, slots, kc[i], vc[i], bt, ctx, nh, nkv, d, 0.088,
                                                             num_splits=sp, workspace=ws, out=out))
            t2 = timed(lambda i: ops.paged_attn_decode_fused(None, cs, slots, kc[i], vc[i], bt, ctx, nh, nkv, d, 0.088,
                                                             num_splits=sp, workspace=ws, out=out, partial=part,
                                                             num_slabs=slabs))
            print(f"{name} ctx {C:5d} splits {sp}: unfused {t0:6.2f} us ({kvb / t0 / 1e3:5.0f} GB/s)  fused {t1:6.2f}  "
                  f"fused on {slabs} slabs {t2:6.2f} ({kvb / t2 / 1e3:5.0f} GB/s)", flush=True)
    del kc, vc
    torch.cuda.empty_cache()


run("qwen2-7b b64 (28/4 heads)", 64, 28, 4, 128, 4, (1, 128, 512, 768, 1024))
run("llama-3-8b b32 (32/8 heads)", 32, 32, 8, 128, 2, (1, 128, 192, 256, 1024, 2048))
