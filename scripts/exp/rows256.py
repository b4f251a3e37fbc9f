"""Exploration: row blocks (129 .. 256 rows) of the tiled GEMM + the slab consumers: a correctness pass against
torch in fp32, then timings on the slices of a Llama-3-8B rank against torch.mm (in-graph, rotating weights)."""
import sys
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops
from mi355_decode._lib import Mi355Error

torch.manual_seed(0)
dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H, F, V, nh, nkv, d = 4096, 14336, 128256, 32, 8, 128

# ---- correctness
for m in (129, 192, 256):
    N, K = 1536, 1024
    w = torch.empty(N, K, dtype=torch.bfloat16, device=dev).normal_(0, 0.05)
    x = torch.randn(m, K, dtype=torch.bfloat16, device=dev)
    ref = x.float() @ w.float().t()
    wt, wti = ops.tile_weights(w), ops.tile_weights(w, interleave_halves=True)
    for tiles in (8, 208, 210, 216, 212, 302):
        y = ops.linear_tiled(x, wt, N, tiles_per_wg=tiles)
        err = (y.float() - ref).abs().max().item()
        part = ops.linear_tiled(x, wt, N, tiles_per_wg=tiles, k_groups=2, epilogue=ops.EPI_PARTIAL)
        y2 = ops.sum_slabs(part, 2, torch.empty_like(y))
        err2 = (y2.float() - ref).abs().max().item()
        act = ops.linear_tiled(x, wti, N, tiles_per_wg=tiles, epilogue=ops.EPI_SILU_MUL)
        parti = ops.linear_tiled(x, wti, N, tiles_per_wg=tiles, k_groups=4, epilogue=ops.EPI_PARTIAL)
        act2 = ops.silu_and_mul_slabs(parti, 4, torch.empty_like(act))
        g, u = ref[:, :N // 2].bfloat16().float(), ref[:, N // 2:].bfloat16().float()
        aref = (torch.nn.functional.silu(g).bfloat16().float() * u)
        e3, e4 = (act.float() - aref).abs().max().item(), (act2.float() - aref).abs().max().item()
        print(f"M={m} T{tiles}: store err {err:.3g}, slabs+sum err {err2:.3g}, swiglu err {e3:.3g}, slabs+swiglu err {e4:.3g} "
              f"(scale {ref.abs().max().item():.3g} / {aref.abs().max().item():.3g})", flush=True)

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
    for g in (2, 4, 8):
        line.append(f"sum{g} {timed(lambda w: ops.sum_slabs(part, g, out), ws[:1], 16):4.1f}")
        if name == "gate_up":
            act = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
            line.append(f"swiglu{g} {timed(lambda w: ops.silu_and_mul_slabs(part, g, act), ws[:1], 16):4.1f}")
    for tiles in (8, 208, 210, 216, 212, 302):
        for groups in (1, 2, 4, 8):
            ksl = ops.tiled_slice_width(M, tiles)
            if K % ksl or groups > K // ksl:
                continue
            wgs = -(-((N + 15) // 16) // ops.tiles_per_workgroup(tiles)) * groups * ((M + 127) // 128)
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
