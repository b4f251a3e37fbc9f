"""A few launches of the decode GEMM at 256 rows in row blocks of 128 / 64 / 32 (for a rocprofv3 FETCH_SIZE pass:
do the row blocks of a column block read the weights from HBM once?).  Llama-3-8B gate_up on one GPU, 235 MB."""
import sys, torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops
M, N, K = 256, 28672, 4096
torch.manual_seed(0)
x = torch.randn(M, K, dtype=torch.bfloat16, device="cuda")
ws = [ops.tile_weights(torch.empty(N, K, dtype=torch.bfloat16, device="cuda").normal_(0, 0.02)) for _ in range(3)]
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for code in (214, 1214, 2008):
    for w in ws * 2:
        ops.linear_tiled(x, w, N, out=out, tiles_per_wg=code)
torch.cuda.synchronize()
