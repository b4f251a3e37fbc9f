"""K split across workgroups into fp32 slabs (EPI_PARTIAL) for the o / down shapes of a tensor-parallel
rank, next to the slab-consuming norm as a stand-in for a slab-sum kernel: what a 'sum the slabs, then
all-reduce' form could cost against the 16-bit-store forms (scripts/exp/bench_tp_reduce_store.py)."""
import json, sys
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops
dev = "cuda"

def timed(body, n):
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        body()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); e.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / n)
    return best

cases = [("8B TP2 down", 64, 4096, 7168), ("8B TP2 o", 64, 4096, 2048), ("8B TP4 down", 128, 4096, 3584),
         ("8B TP4 o", 128, 4096, 1024), ("70B TP8 down", 16, 8192, 3584), ("70B TP8 o", 16, 8192, 1024)]
for name, M, N, K in cases:
    copies = max(4, min(32, int(600e6 // (N * K * 2))))
    wt = [ops.tile_weights(torch.empty(N, K, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)) for _ in range(copies)]
    x = torch.randn(M, K, dtype=torch.bfloat16, device=dev)
    part = torch.empty(8, M, N, dtype=torch.float32, device=dev)
    resid = torch.randn(M, N, dtype=torch.bfloat16, device=dev)
    nw = torch.ones(N, dtype=torch.bfloat16, device=dev)
    xn = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    rec = {"case": name, "M": M, "N": N, "K": K}
    res = {}
    for cand in ops.linear_tiled_candidates(N, K, True, M):
        if cand[1] == 1:
            continue
        try:
            t_g = timed(lambda: [ops.linear_tiled(x, w, N, partial=part, tiles_per_wg=cand[0], k_groups=cand[1], epilogue=ops.EPI_PARTIAL) for w in wt], copies)
            t_p = timed(lambda: [(ops.linear_tiled(x, w, N, partial=part, tiles_per_wg=cand[0], k_groups=cand[1], epilogue=ops.EPI_PARTIAL),
                                  ops.fused_add_rms_norm_partial(part, cand[1], resid, nw, 1e-5, xn)) for w in wt], copies)
            res[f"{cand[0]}/{cand[1]}"] = [round(t_g, 2), round(t_p, 2)]
        except Exception:
            pass
    best = sorted(res.items(), key=lambda kv: kv[1][1])[:4]
    rec["best_partial_gemm_and_pair_with_norm_us"] = best
    print(json.dumps(rec), flush=True)
