"""Decode attention kernel time vs context length and split count (in-graph, 32 layers' caches rotating)."""
import sys
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops

dev, dt = "cuda", torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32          # sequences
nh, nkv, d, L = (int(sys.argv[2]) if len(sys.argv) > 2 else 32), 8, 128, 16
MAXCTX = 2304
blocks_per_seq = MAXCTX // 16
nb = B * blocks_per_seq
kc = [torch.randn(nb, nkv, 16, d, dtype=dt, device=dev) for _ in range(L)]
vc = [torch.randn(nb, nkv, d, 16, dtype=dt, device=dev) for _ in range(L)]
bt = torch.arange(nb, dtype=torch.int32, device=dev).view(B, blocks_per_seq)
q = torch.randn(B, nh * d, dtype=dt, device=dev)
out = torch.empty(B, nh * d, dtype=dt, device=dev)
ws = torch.empty(ops.decode_workspace_bytes(B, nh, d, 16), dtype=torch.uint8, device=dev)


def timed(fn):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
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


for c in ((1, 32, 128, 192, 256, 512, 1024, 2048) if len(sys.argv) <= 3 else tuple(int(x) for x in sys.argv[3].split(","))):
    ctx = torch.full((B,), c, dtype=torch.int32, device=dev)
    line = []
    for sp in (1, 2, 4, 8):
        if c // sp < 32 and sp > 1:
            continue
        t = timed(lambda i: ops.paged_attn_decode(q, kc[i], vc[i], bt, ctx, nh, nkv, d, 0.088, num_splits=sp, workspace=ws, out=out))
        kvb = B * c * nkv * d * 2 * 2
        line.append(f"splits {sp}: {t:6.2f}us ({kvb / t / 1e3:5.0f} GB/s)")
    print(f"ctx {c:5d}: " + "  ".join(line), flush=True)
