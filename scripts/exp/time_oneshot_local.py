"""Exploration: what the one-shot all-reduce kernel costs a rank locally (a communicator of one rank: the kernel
delivers into its own buffer, flags, waits for its own flag, sums one slot), behind a GEMM that leaves dirty fp32
slabs in the L2 -- the state it runs in inside a step.  In-graph timing."""
import sys
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops
from mi355_decode.collectives import OneShotAllReduce

dev = "cuda"
ar = OneShotAllReduce(rank=0, world=1, max_bytes=4 << 20)
H = 4096


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); e.synchronize()
        best = min(best, s.elapsed_time(e) * 1000 / reps)
    return best


for rows in (64, 128, 256):
    t = torch.randn(rows, H, device=dev).to(torch.bfloat16)
    slabs = torch.randn(4, rows, H, device=dev)
    out = torch.empty(rows, H, dtype=torch.bfloat16, device=dev)
    w = torch.randn(4096, 2048, device=dev).to(torch.bfloat16)
    wt = ops.tile_weights(w)
    x = torch.randn(rows, 2048, device=dev).to(torch.bfloat16)
    part = torch.empty(4, rows, H, dtype=torch.float32, device=dev)
    gemm = lambda: ops.linear_tiled(x, wt, H, partial=part, tiles_per_wg=8, k_groups=4 if rows > 64 else 2, epilogue=ops.EPI_PARTIAL)
    g = timed(gemm)
    print(f"rows {rows} ({rows * H * 2 >> 10} KiB): plain {timed(lambda: ar.run(t)):5.1f} us, from 4 slabs {timed(lambda: ar.run_slabs(slabs, 4, out)):5.1f} us, "
          f"sum_slabs kernel {timed(lambda: ops.sum_slabs(slabs, 4, out)):4.1f} us; behind a slab GEMM ({g:4.1f} us): "
          f"gemm+plain {timed(lambda: (gemm(), ar.run(t))):5.1f}, gemm+slabs {timed(lambda: (gemm(), ar.run_slabs(part, 4 if rows > 64 else 2, out))):5.1f}, "
          f"gemm+sum+plain {timed(lambda: (gemm(), ops.sum_slabs(part, 4 if rows > 64 else 2, out), ar.run(out))):5.1f}", flush=True)
ar.check()
