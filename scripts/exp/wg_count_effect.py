"""Does a weight-streaming GEMM whose workgroup count is not a multiple of the 256 CUs lose time
to the idle CUs?  M = 32, K = 4096, the qkv form (8 tiles per workgroup, K split over 4 groups,
fp32 slabs) and the gate_up-like store form (8 tiles, no split), N swept so that the launch has
128 ... 512 workgroups; in-graph timing on rotating weight copies.  Time against bytes: a
straight line means only bytes + a fixed cost matter."""
import json
import sys
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops

dev, M, K = "cuda", 32, 4096
x = torch.randn(M, K, dtype=torch.bfloat16, device=dev)


def timed(fn, ws):
    fn(ws[0]); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(2):
            for w in ws:
                fn(w)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); e.synchronize()
        best = min(best, s.elapsed_time(e) * 1000 / (2 * len(ws)))
    return best


for form, groups, Ns in (("partial 8/4", 4, (4096, 5120, 6144, 7168, 8192, 9216, 10240, 12288, 16384)),
                         ("store 8/1", 1, (16384, 24576, 28672, 32768, 36864, 40960, 49152, 65536))):
    for N in Ns:
        copies = max(3, int(700e6 // (N * K * 2)) + 1)
        wts = [ops.tile_weights(torch.empty(N, K, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)) for _ in range(copies)]
        if groups > 1:
            part = torch.empty(groups, M, N, dtype=torch.float32, device=dev)
            fn = lambda w: ops.linear_tiled(x, w, N, partial=part, tiles_per_wg=8, k_groups=groups, epilogue=ops.EPI_PARTIAL)
        else:
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            fn = lambda w: ops.linear_tiled(x, w, N, out=out, tiles_per_wg=8)
        us = timed(fn, wts)
        wgs = (N // 16 + 7) // 8 * groups
        print(json.dumps({"form": form, "N": N, "workgroups": wgs, "us": round(us, 2), "weight_MB": round(N * K * 2 / 1e6, 1),
                          "GBs": round(N * K * 2 / us / 1e3)}), flush=True)
        del wts
