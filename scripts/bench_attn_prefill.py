#!/usr/bin/env python3
"""Prefill attention alone: time per launch and MFMA rate at the benchmark configurations' shapes.
MI355_DECODE_LIB=<other .so> python3 scripts/bench_attn_prefill.py compares builds."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-backend-public_amd"))
import torch
from mi355_decode import ops

def run(B, P, nh, nkv, d, reps=20):
    T = B * P
    dev = "cuda"
    nblk = B * ((P + 15) // 16)
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.empty(T, nh * d, dtype=torch.bfloat16, device=dev).normal_(0, 1, generator=g)
    kc = torch.empty(nblk, nkv, 16, d, dtype=torch.bfloat16, device=dev).normal_(0, 1, generator=g)
    vc = torch.empty(nblk, nkv, d, 16, dtype=torch.bfloat16, device=dev).normal_(0, 1, generator=g)
    bt = torch.arange(nblk, dtype=torch.int32, device=dev).view(B, -1)
    cu = torch.arange(0, T + 1, P, dtype=torch.int32, device=dev)
    ctx = torch.full((B,), P, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    f = lambda: ops.paged_attn_prefill(q, kc, vc, bt, cu, ctx, P, nh, nkv, d, d ** -0.5, out=out)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        f()
    e.record(); e.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    flops = 4.0 * B * (P * (P + 1) / 2) * nh * d
    return {"B": B, "P": P, "heads": f"{nh}/{nkv}", "d": d, "us": round(us, 1), "tflops": round(flops / us / 1e6, 1),
            "mfma_frac_of_2500": round(flops / us / 1e6 / 2500, 4)}

if __name__ == "__main__":
    res = [run(32, 128, 32, 8, 128), run(16, 2048, 8, 1, 128), run(64, 512, 28, 4, 128), run(4, 2048, 32, 8, 128)]
    print(json.dumps({"lib": os.environ.get("MI355_DECODE_LIB", "in-tree"), "results": res}, indent=1))
