"""Time of the random-sampling launch (argmax + draw) for 32 rows of a 128256-token vocabulary,
with a cut (top-k 50, top-p 0.9: radix select + sort + inverse CDF) and without one (whole-row
Gumbel-max), inside a HIP graph."""
import json
import sys
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops

B, V = 32, 128256
lg = (torch.randn(B, V, device="cuda") * 3).to(torch.bfloat16)
T = torch.full((B,), 0.8, device="cuda")
P = {"cut": (torch.full((B,), 50, dtype=torch.int32, device="cuda"), torch.full((B,), 0.9, device="cuda")),
     "nucleus_only": (torch.full((B,), -1, dtype=torch.int32, device="cuda"), torch.full((B,), 0.9, device="cuda")),
     "whole_vocabulary": (torch.full((B,), -1, dtype=torch.int32, device="cuda"), torch.ones(B, device="cuda"))}
seeds = torch.arange(B, device="cuda")
steps = torch.zeros(B, dtype=torch.int64, device="cuda")
out = torch.empty(B, dtype=torch.int64, device="cuda")
rec = {}
for name, (k, p) in P.items():
    def body():
        for _ in range(20):
            ops.sample(lg, T, k, p, seeds, steps, out=out, cut_rows=name != "whole_vocabulary")
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        body()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); e.synchronize()
        best = min(best, s.elapsed_time(e) * 1000 / 20)
    rec[name + "_us"] = round(best, 2)
print(json.dumps(rec))
