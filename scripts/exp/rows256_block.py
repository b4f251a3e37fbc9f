"""Exploration: 256 rows as ONE row block (16 row tiles per wave, x slices of 128 columns) against two blocks of 128
and four of 64, on the shapes of Llama-3-8B at batch 256 (one GPU) and of a TP8 rank."""
import json, sys, torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops
from mi355_decode._lib import Mi355Error
M = 256
shapes = [("tp1 gate_up", 28672, 4096), ("tp1 down", 4096, 14336), ("tp1 qkv", 6144, 4096), ("tp1 o", 4096, 4096), ("tp1 lm_head", 128256, 4096),
          ("tp8 gate_up", 3584, 4096), ("tp8 down", 4096, 1792), ("tp8 lm_head", 16032, 4096)]

def timed(fn, ws, reps):
    fn(ws[0]); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            for w in ws: fn(w)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); e.synchronize()
        best = min(best, s.elapsed_time(e) * 1000 / (reps * len(ws)))
    return best

for name, N, K in shapes:
    copies = max(2, int(600e6 // (N * K * 2)) + 1)
    ws = [torch.empty(N, K, dtype=torch.bfloat16, device="cuda").normal_(0, 0.02) for _ in range(copies)]
    wts = [ops.tile_weights(w) for w in ws]
    x = torch.randn(M, K, dtype=torch.bfloat16, device="cuda")
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ref = torch.mm(x, ws[0].t()).float()
    reps = max(1, 32 // copies)
    row = {"proj": name, "library_us": round(timed(lambda w: torch.mm(x, w.t(), out=out), ws, reps), 1)}
    part = torch.empty(8, M, N, dtype=torch.float32, device="cuda") if N * M * 32 < 2e9 else None
    for code in (8, 216, 214, 1008, 1216, 3008):
        for g in (1, 2, 4):
            if g > 1 and part is None: continue
            try:
                if g == 1:
                    call = lambda wt: ops.linear_tiled(x, wt, N, out=out, tiles_per_wg=code)
                    call(wts[0]); got = out.float()
                else:
                    call = lambda wt: ops.linear_tiled(x, wt, N, partial=part, tiles_per_wg=code, k_groups=g, epilogue=ops.EPI_PARTIAL)
                    call(wts[0]); got = part[:g].sum(0)
                err = (got - ref).abs().max().item()
                if not err < 0.05 * ref.abs().max().item() + 0.05:
                    row[f"{code}/{g}"] = "WRONG"; continue
                row[f"{code}/{g}"] = round(timed(call, wts, reps), 1)
            except Mi355Error:
                pass
    print(json.dumps(row), flush=True)
