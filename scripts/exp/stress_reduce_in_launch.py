"""Stress of the in-launch K-split reduction (EPI_REDUCE_STORE): the same launch repeated many times,
each result compared bit for bit with the first slab-sum reference; prints how many launches differed.
MI355_DECODE_LIB=<other .so> compares builds."""
import sys
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops
dev = "cuda"
total_bad = 0
for (M, N, K, tiles, groups) in [(17, 6144, 4096, 104, 2), (32, 4096, 4096, 104, 4), (32, 4096, 14336, 104, 4), (9, 520, 2048, 104, 2)]:
    g = torch.Generator(device=dev).manual_seed(M * 7 + N)
    x = torch.randn(M, K, dtype=torch.float16, device=dev, generator=g)
    w = torch.randn(N, K, dtype=torch.float16, device=dev, generator=g) * 0.05
    wt = ops.tile_weights(w)
    part = ops.linear_tiled(x, wt, N, k_groups=groups, epilogue=ops.EPI_PARTIAL, tiles_per_wg=tiles)
    ref = part[0].clone()
    for s in range(1, groups):
        ref += part[s]
    ref = ref.to(torch.float16)
    counters = torch.zeros(-(-((N + 15) // 16) // (tiles % 100)), dtype=torch.int32, device=dev)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    bad = 0
    n = 400
    for i in range(n):
        if i % 4 == 0:
            junk.fill_(i & 255)                   # churn the caches between launches
        ops.linear_tiled_fused(x, wt, N, tiles_per_wg=tiles, k_groups=groups, epilogue=ops.EPI_REDUCE_STORE, out=out,
                               counters=counters)
        if not torch.equal(out, ref):
            bad += 1
    print(f"M={M} N={N} K={K} {tiles}/{groups}: {bad} of {n} launches differ", flush=True)
    total_bad += bad
print("TOTAL_BAD", total_bad)
