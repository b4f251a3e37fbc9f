"""The four prefill GEMMs of a Llama-3-8B layer at 4096 packed tokens, run in the order of a layer
(32 layers' worth of distinct weights would not fit here: 4 copies are cycled) and timed per kind
with HIP events inside the sequence -- what the library's default choice sustains in context."""
import json, sys
import torch
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
shapes = {"qkv": (6144, 4096), "o": (4096, 4096), "gate_up": (28672, 4096), "down": (4096, 14336)}
dev = "cuda"
C = 4
ws = {k: [torch.empty(n, kk, dtype=torch.bfloat16, device=dev).normal_(0, 0.02) for _ in range(C)] for k, (n, kk) in shapes.items()}
xs = {k: torch.randn(M, kk, dtype=torch.bfloat16, device=dev) for k, (n, kk) in shapes.items()}
outs = {k: torch.empty(M, n, dtype=torch.bfloat16, device=dev) for k, (n, kk) in shapes.items()}
ev = {k: [] for k in shapes}
def layer(i, record):
    for k in shapes:
        if record:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
        torch.mm(xs[k], ws[k][i % C].t(), out=outs[k])
        if record:
            e.record(); ev[k].append((s, e))
for i in range(8):
    layer(i, False)
torch.cuda.synchronize()
for i in range(64):
    layer(i, True)
torch.cuda.synchronize()
res = {k: round(sum(s.elapsed_time(e) for s, e in v) * 1e3 / len(v), 1) for k, v in ev.items()}
res["layer_us"] = round(sum(res.values()), 1)
res["tflops"] = {k: round(2.0 * M * n * kk / res[k] / 1e6, 1) for k, (n, kk) in shapes.items()}
print(json.dumps(res))
