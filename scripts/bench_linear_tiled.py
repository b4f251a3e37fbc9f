"""Micro-benchmark: mi355_linear_tiled vs torch.mm (hipBLASLt) on the decode GEMM shapes, in-graph timing,
weights rotating over > 256 MiB."""
import sys
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops

torch.manual_seed(0)
dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [("gate_up", 28672, 4096), ("down", 4096, 14336), ("qkv", 6144, 4096), ("o", 4096, 4096), ("lm_head", 128256, 4096)]


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


for name, N, K in shapes:
    copies = max(2, int(600e6 // (N * K * 2)) + 1)
    ws = [torch.empty(N, K, dtype=torch.bfloat16, device=dev).normal_(0, 0.02) for _ in range(copies)]
    wts = [ops.tile_weights(w) for w in ws]
    x = torch.randn(M, K, dtype=torch.bfloat16, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    reps = max(1, 64 // copies)
    nbytes = (N * K + M * K + M * N) * 2
    line = [f"torch {timed(lambda w: torch.mm(x, w.t(), out=out), ws, reps):6.1f}us"]
    phases = K // 1024
    for tiles in (4, 8, 104, 108):
        us = timed(lambda wt: ops.linear_tiled(x, wt, N, out=out, tiles_per_wg=tiles), wts, reps)
        line.append(f"T{tiles} {us:6.1f}us/{nbytes/us/1e3:5.0f}GB/s")
    for tiles, groups in ((4, 2), (104, 2), (4, 4), (8, 2), (8, 4), (8, 7), (4, 7), (8, 14), (104, 4), (104, 8), (108, 4), (108, 7), (108, 14)):
        ksl = 512 if tiles >= 100 else 1024
        if (K // ksl) % groups:
            continue
        part = torch.empty(groups, M, N, dtype=torch.float32, device=dev)
        us = timed(lambda wt: ops.linear_tiled(x, wt, N, partial=part, tiles_per_wg=tiles, k_groups=groups, epilogue=ops.EPI_PARTIAL), wts, reps)
        line.append(f"T{tiles}G{groups} {us:6.1f}us/{nbytes/us/1e3:5.0f}GB/s")
    if name == "gate_up":
        wti = [ops.tile_weights(w, interleave_halves=True) for w in ws]
        act = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
        for tiles in (4, 8, 108):
            us = timed(lambda wt: ops.linear_tiled(x, wt, N, out=act, tiles_per_wg=tiles, epilogue=ops.EPI_SILU_MUL), wti, reps)
            line.append(f"silu-T{tiles} {us:6.1f}us/{nbytes/us/1e3:5.0f}GB/s")
    print(f"{name:8s} M={M} N={N} K={K}: " + "  ".join(line), flush=True)
