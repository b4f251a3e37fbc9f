"""mi355_linear_tiled (planner geometry) vs torch.mm at M = 32 / 64 / 128 on the decode shapes, TP 1 / 2 / 4."""
import sys
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops

dev = "cuda"


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


for tp, M in ((1, 32), (2, 64), (4, 128), (1, 64), (1, 128)):
    shapes = [("gate_up", 28672, 4096, False), ("down", 4096, 14336, True), ("qkv", 6144 // tp, 4096, True),
              ("o", 4096, 4096 // tp, True), ("lm_head", 128256, 4096, False)]
    for name, N, K, split in shapes:
        copies = max(2, int(600e6 // (N * K * 2)) + 1)
        ws = [torch.empty(N, K, dtype=torch.bfloat16, device=dev).normal_(0, 0.02) for _ in range(copies)]
        wts = [ops.tile_weights(w, interleave_halves=(name == "gate_up")) for w in ws]
        x = torch.randn(M, K, dtype=torch.bfloat16, device=dev)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        reps = max(1, 64 // copies)
        nbytes = (N * K + M * K + M * N) * 2
        t_t = timed(lambda w: torch.mm(x, w.t(), out=out), ws, reps)
        tiles, groups = ops.plan_linear_tiled(N, K, split, M)
        if name == "gate_up":
            act = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
            t_k = timed(lambda w: ops.linear_tiled(x, w, N, out=act, tiles_per_wg=tiles, epilogue=ops.EPI_SILU_MUL), wts, reps)
        elif groups > 1:
            part = torch.empty(groups, M, N, dtype=torch.float32, device=dev)
            t_k = timed(lambda w: ops.linear_tiled(x, w, N, partial=part, tiles_per_wg=tiles, k_groups=groups, epilogue=ops.EPI_PARTIAL), wts, reps)
        else:
            t_k = timed(lambda w: ops.linear_tiled(x, w, N, out=out, tiles_per_wg=tiles), wts, reps)
        extra = ""
        if name == "o" and tp > 1:
            cnt = torch.zeros(N // 64, dtype=torch.int32, device=dev)
            part = torch.empty(groups, M, N, dtype=torch.float32, device=dev)
            t_r = timed(lambda w: ops.linear_tiled_fused(x, w, N, tiles_per_wg=tiles, k_groups=groups, epilogue=ops.EPI_REDUCE_STORE, partial=part, out=out, counters=cnt), wts, reps)
            extra = f"  reduce_store {t_r:6.1f}us"
        print(f"tp{tp} M={M:3d} {name:8s} {N}x{K}: torch {t_t:6.1f}us  tiled[{tiles}/{groups}] {t_k:6.1f}us ({nbytes/t_k/1e3:5.0f} GB/s){extra}", flush=True)
