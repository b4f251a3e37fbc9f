"""EXPERIMENT: decode attention at the shape of a Llama-3-70B rank on 8 GPUs (configs[3]): 16 sequences,
8 query heads on 1 kv head, contexts 2176, vs the number of KV splits."""
import sys
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops

dev, dt = "cuda", torch.bfloat16
B, nh, nkv, d, L = 16, 8, 1, 128, 16
C = 2176
bps = (C + 15) // 16
nb = B * bps
kc = [torch.randn(nb, nkv, 16, d, dtype=dt, device=dev) for _ in range(L)]
vc = [torch.randn(nb, nkv, d, 16, dtype=dt, device=dev) for _ in range(L)]
bt = torch.arange(nb, dtype=torch.int32, device=dev).view(B, bps)
ctx = torch.full((B,), C, dtype=torch.int32, device=dev)
pos = (ctx - 1).to(torch.int64)
slots = (bt[torch.arange(B), (pos // 16)].to(torch.int64) * 16 + pos % 16)
cs = torch.randn(4096, d, dtype=dt, device=dev)
width = (nh + 2 * nkv) * d
qkv = torch.randn(B, width, dtype=dt, device=dev)
out = torch.empty(B, nh * d, dtype=dt, device=dev)
ws = torch.empty(ops.decode_workspace_bytes(B, nh, d, 64), dtype=torch.uint8, device=dev)


def timed(fn):
    for i in range(3):
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


kvb = B * C * nkv * d * 2 * 2
for sp in (1, 2, 4, 8, 9, 12, 16, 24, 32):
    t = timed(lambda i: ops.paged_attn_decode_fused(qkv, cs, slots, kc[i], vc[i], bt, ctx, nh, nkv, d, 0.088, num_splits=sp,
                                                    workspace=ws, out=out))
    print(f"splits {sp:2d} ({B * nkv * sp:4d} workgroups): {t:6.2f} us ({kvb / t / 1e3:5.0f} GB/s)", flush=True)
