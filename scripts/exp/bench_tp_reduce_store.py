"""o / down of a tensor-parallel rank need a 16-bit output (it is all-reduced).  Forms: the store
epilogue with K unsplit or split across the waves of a workgroup (what the engine times today), the
library GEMM, and K split across workgroups with the in-launch ticket reduction
(EPI_REDUCE_STORE: the last workgroup of a column group sums the slabs).  us per launch, graph replay
over distinct weight copies."""
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
    ws = [torch.empty(N, K, dtype=torch.bfloat16, device=dev).normal_(0, 0.02) for _ in range(copies)]
    wt = [ops.tile_weights(w) for w in ws]
    x = torch.randn(M, K, dtype=torch.bfloat16, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    part = torch.empty(8, M, N, dtype=torch.float32, device=dev)
    counters = torch.zeros(4096, dtype=torch.int32, device=dev)
    rec = {"case": name, "M": M, "N": N, "K": K, "MB": round(N * K * 2 / 1e6, 1)}
    rec["library"] = round(timed(lambda: [torch.mm(x, w.t(), out=out) for w in ws], copies), 2)
    store = {}
    for cand in ops.linear_tiled_candidates(N, K, False, M):
        try:
            store[f"{cand[0]}/1"] = round(timed(lambda: [ops.linear_tiled(x, w, N, out=out, tiles_per_wg=cand[0]) for w in wt], copies), 2)
        except Exception:
            pass
    rec["store_best"] = min(store.items(), key=lambda kv: kv[1]) if store else None
    red = {}
    for tiles in (8, 104):
        for g in (2, 4, 8):
            if g > K // ops.tiled_slice_width(M, tiles):
                continue
            try:
                red[f"{tiles}/{g}"] = round(timed(lambda: [ops.linear_tiled_fused(x, w, N, tiles_per_wg=tiles, k_groups=g,
                                                                                 epilogue=ops.EPI_REDUCE_STORE, out=out, partial=part,
                                                                                 counters=counters) for w in wt], copies), 2)
            except Exception as exc:
                red[f"{tiles}/{g}"] = str(exc)[:60]
    rec["reduce_store"] = red
    print(json.dumps(rec), flush=True)
