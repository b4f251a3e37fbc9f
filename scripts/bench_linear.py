"""Micro-benchmark: mi355_linear_decode vs torch.mm (hipBLASLt) on the decode GEMM shapes.
Weights rotate over enough copies to exceed the 256 MiB Infinity Cache."""
import sys, time
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops

torch.manual_seed(0)
dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [("gate_up", 28672, 4096), ("down", 4096, 14336), ("qkv", 6144, 4096), ("o", 4096, 4096), ("lm_head", 128256, 4096)]
for name, N, K in shapes:
    copies = max(2, int(600e6 // (N * K * 2)) + 1)
    ws = [torch.empty(N, K, dtype=torch.bfloat16, device=dev).normal_(0, 0.02) for _ in range(copies)]
    x = torch.randn(M, K, dtype=torch.bfloat16, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ref = torch.mm(x, ws[0].t()).float()
    res = {}
    for label, fn in [("torch", lambda w: torch.mm(x, w.t(), out=out)),
                      ("ks1", lambda w: ops.linear_decode(x, w, out=out, k_slices=1)),
                      ("ks2", lambda w: ops.linear_decode(x, w, out=out, k_slices=2)),
                      ("ks4", lambda w: ops.linear_decode(x, w, out=out, k_slices=4)),
                      ]:
        fn(ws[0])
        torch.cuda.synchronize()
        err = (out.float() - ref).abs().max().item()
        g = torch.cuda.CUDAGraph()
        reps = max(1, 64 // copies)
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
        us = s.elapsed_time(e) * 1000 / (5 * reps * copies)
        res[label] = (us, (N * K + M * K + M * N) * 2 / us / 1e3, err)
    print(f"{name:8s} M={M} N={N} K={K} copies={copies}: " + "  ".join(f"{k} {v[0]:7.1f}us {v[1]:6.0f}GB/s err {v[2]:.3f}" for k, v in res.items()), flush=True)
