"""Decode attention: rope kernel + attention kernel vs the fused launch (16-bit rows / fp32 slabs), in-graph."""
import sys
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops

dev, dt = "cuda", torch.bfloat16
B, nh, nkv, d, L, ctx_len = 32, 32, 8, 128, 32, 192
width = (nh + 2 * nkv) * d
blocks_per_seq = 16
nb = B * blocks_per_seq
kc = [torch.randn(nb, nkv, 16, d, dtype=dt, device=dev) for _ in range(L)]
vc = [torch.randn(nb, nkv, d, 16, dtype=dt, device=dev) for _ in range(L)]
bt = torch.arange(nb, dtype=torch.int32, device=dev).view(B, blocks_per_seq)
ctx = torch.full((B,), ctx_len, dtype=torch.int32, device=dev)
pos = (ctx - 1).to(torch.int64)
slots = (bt[torch.arange(B), (pos // 16)].to(torch.int64) * 16 + pos % 16)
cs = torch.randn(4096, d, dtype=dt, device=dev)
qkv = torch.randn(B, width, dtype=dt, device=dev)
out = torch.empty(B, nh * d, dtype=dt, device=dev)
parts = {s: torch.randn(s, B, width, dtype=torch.float32, device=dev) for s in (1, 2, 4)}


def timed(fn):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(2):
            for i in range(L):
                fn(i)
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1000 / (5 * 2 * L)


def unfused(i):
    ops.rope_kv_write(qkv, pos, cs, slots, kc[i], vc[i], nh, nkv, d)
    ops.paged_attn_decode(qkv[:, : nh * d], kc[i], vc[i], bt, ctx, nh, nkv, d, 0.088, out=out)


def unfused_partial(i):
    ops.rope_kv_write(qkv, pos, cs, slots, kc[i], vc[i], nh, nkv, d, partial=parts[2], num_slabs=2)
    ops.paged_attn_decode(qkv[:, : nh * d], kc[i], vc[i], bt, ctx, nh, nkv, d, 0.088, out=out)


print(f"attention alone                 {timed(lambda i: ops.paged_attn_decode(qkv[:, : nh * d], kc[i], vc[i], bt, ctx, nh, nkv, d, 0.088, out=out)):6.2f} us")
print(f"rope alone (16-bit rows)        {timed(lambda i: ops.rope_kv_write(qkv, pos, cs, slots, kc[i], vc[i], nh, nkv, d)):6.2f} us")
print(f"rope + attention (16-bit rows)  {timed(unfused):6.2f} us")
print(f"rope + attention (2 slabs)      {timed(unfused_partial):6.2f} us")
print(f"fused, 16-bit rows              {timed(lambda i: ops.paged_attn_decode_fused(qkv, cs, slots, kc[i], vc[i], bt, ctx, nh, nkv, d, 0.088, out=out)):6.2f} us")
for s in (1, 2, 4):
    print(f"fused, {s} slab(s)                {timed(lambda i: ops.paged_attn_decode_fused(None, cs, slots, kc[i], vc[i], bt, ctx, nh, nkv, d, 0.088, out=out, partial=parts[s], num_slabs=s)):6.2f} us")
noslot = torch.full((B,), -1, dtype=torch.int64, device=dev)
print(f"fused, 2 slabs, nothing stored  {timed(lambda i: ops.paged_attn_decode_fused(None, cs, noslot, kc[i], vc[i], bt, ctx, nh, nkv, d, 0.088, out=out, partial=parts[2], num_slabs=2)):6.2f} us")
