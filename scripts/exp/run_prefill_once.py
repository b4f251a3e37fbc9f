"""A few launches of the prefill attention kernel at one shape, for counter passes:
python3 scripts/exp/run_prefill_once.py B P heads kv_heads [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vllm-backend-public_amd"))
import torch
from mi355_decode import ops

B, P, nh, nkv = (int(v) for v in sys.argv[1:5])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 4
d, dev = 128, "cuda"
T = B * P
nblk = B * ((P + 15) // 16)
g = torch.Generator(device=dev).manual_seed(0)
q = torch.empty(T, nh * d, dtype=torch.bfloat16, device=dev).normal_(0, 1, generator=g)
kc = torch.empty(nblk, nkv, 16, d, dtype=torch.bfloat16, device=dev).normal_(0, 1, generator=g)
vc = torch.empty(nblk, nkv, d, 16, dtype=torch.bfloat16, device=dev).normal_(0, 1, generator=g)
bt = torch.arange(nblk, dtype=torch.int32, device=dev).view(B, -1)
cu = torch.arange(0, T + 1, P, dtype=torch.int32, device=dev)
ctx = torch.full((B,), P, dtype=torch.int32, device=dev)
out = torch.empty_like(q)
for _ in range(n):
    ops.paged_attn_prefill(q, kc, vc, bt, cu, ctx, P, nh, nkv, d, d ** -0.5, out=out)
torch.cuda.synchronize()
print("done")
