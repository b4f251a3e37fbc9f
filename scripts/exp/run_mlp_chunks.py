"""Prefill MLP of one Llama-3-8B layer at 4096 packed tokens: gate_up GEMM -> SwiGLU -> down GEMM,
whole batch at once vs in 2 / 4 token chunks (a chunk's 59-117 MB of gate_up output may still be
in the memory-side cache when the SwiGLU kernel reads it; the weights are then streamed once per
chunk).  us per layer, events around 16 layers' worth (4 weight copies cycled)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vllm-backend-public_amd"))
import torch
from mi355_decode import ops
M, H, F = 4096, 4096, 14336
dev = "cuda"
C = 4
wgu = [torch.empty(2 * F, H, dtype=torch.bfloat16, device=dev).normal_(0, 0.02) for _ in range(C)]
wdn = [torch.empty(H, F, dtype=torch.bfloat16, device=dev).normal_(0, 0.02) for _ in range(C)]
x = torch.randn(M, H, dtype=torch.bfloat16, device=dev)
gu = torch.empty(M, 2 * F, dtype=torch.bfloat16, device=dev)
act = torch.empty(M, F, dtype=torch.bfloat16, device=dev)
dn = torch.empty(M, H, dtype=torch.bfloat16, device=dev)

def layer(i, chunks):
    step = M // chunks
    for c in range(chunks):
        r = slice(c * step, (c + 1) * step)
        torch.mm(x[r], wgu[i % C].t(), out=gu[r])
        ops.silu_and_mul(gu[r], out=act[r])
        torch.mm(act[r], wdn[i % C].t(), out=dn[r])

res = {}
for chunks in (1, 2, 4, 1, 2, 4):
    for i in range(4):
        layer(i, chunks)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(16):
        layer(i, chunks)
    e.record(); e.synchronize()
    res.setdefault(f"chunks_{chunks}_us_per_layer", []).append(round(s.elapsed_time(e) * 1e3 / 16, 1))
print(json.dumps(res))
