"""Phase times of the cut path of the sampling kernel (row 0's workgroup), from a build of
sample.hip with wall-clock stamps and a device printf (MI355_DECODE_LIB=<that build>)."""
import sys
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops

B, V = 32, 128256
lg = (torch.randn(B, V, device="cuda") * 3).to(torch.bfloat16)
T = torch.full((B,), 0.8, device="cuda")
seeds = torch.arange(B, device="cuda")
steps = torch.zeros(B, dtype=torch.int64, device="cuda")
for k, p in ((50, 0.9), (-1, 0.9), (1024, 1.0)):
    kk = torch.full((B,), k, dtype=torch.int32, device="cuda")
    pp = torch.full((B,), p, device="cuda")
    for _ in range(3):
        ops.sample(lg, T, kk, pp, seeds, steps)
        torch.cuda.synchronize()
