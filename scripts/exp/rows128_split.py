"""Exploration: the existing tiled kernels at 128 rows on the slices of a Llama-3-8B TP4 rank, K split over
workgroups into fp32 slabs (which the o / down / gate_up call sites of a tensor-parallel rank cannot take
today: their consumers want 16-bit), against torch.mm.  In-graph timing over rotating weight copies."""
import sys
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops
from mi355_decode._lib import Mi355Error

torch.manual_seed(0)
dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
tp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H, F, V, nh, nkv, d = 4096, 14336, 128256, 32, 8, 128
shapes = [("qkv", (nh + 2 * nkv) * d // tp, H), ("o", H, nh * d // tp), ("gate_up", 2 * F // tp, H), ("down", H, F // tp),
          ("lm_head", V // tp, H)]


def timed(fn, ws, reps):
    fn(ws[0]); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            for w in ws:
                fn(w)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record(); e.synchronize()
        best = min(best, s.elapsed_time(e) * 1000 / (reps * len(ws)))
    return best


for name, N, K in shapes:
    copies = max(2, int(400e6 // (N * K * 2)) + 1)
    ws = [torch.empty(N, K, dtype=torch.bfloat16, device=dev).normal_(0, 0.02) for _ in range(copies)]
    wts = [ops.tile_weights(w) for w in ws]
    x = torch.randn(M, K, dtype=torch.bfloat16, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    reps = max(1, 64 // copies)
    line = [f"torch {timed(lambda w: torch.mm(x, w.t(), out=out), ws, reps):6.1f}"]
    part = torch.empty(8, M, N, dtype=torch.float32, device=dev)
    for tiles in (8, 216, 212, 302, 104, 108, 208, 304, 402):
        for groups in (1, 2, 4, 8):
            ksl = ops.tiled_slice_width(M, tiles)
            if K % ksl or groups > K // ksl:
                continue
            wgs = -(-((N + 15) // 16) // ops.tiles_per_workgroup(tiles)) * groups
            if not 48 <= wgs <= 1024:
                continue
            try:
                if groups == 1:
                    us = timed(lambda wt: ops.linear_tiled(x, wt, N, out=out, tiles_per_wg=tiles), wts, reps)
                else:
                    us = timed(lambda wt: ops.linear_tiled(x, wt, N, partial=part, tiles_per_wg=tiles, k_groups=groups,
                                                           epilogue=ops.EPI_PARTIAL), wts, reps)
            except Mi355Error:
                continue
            line.append(f"T{tiles}G{groups}({wgs}) {us:5.1f}")
    print(f"{name:8s} M={M} N={N} K={K} ({N*K*2/1e6:.1f} MB): " + "  ".join(line), flush=True)
