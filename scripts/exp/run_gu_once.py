"""A few launches of the Qwen2-7B gate_up GEMM + SwiGLU epilogue (for rocprofv3 passes):
run_gu_once.py <rows> <tiles_per_wg> <fp8|bf16>"""
import sys
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops

M, tiles, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
N, K = 37888, 3584
torch.manual_seed(0)
x = torch.randn(M, K, dtype=torch.bfloat16, device="cuda")
out = torch.empty(M, N // 2, dtype=torch.bfloat16, device="cuda")
w = torch.empty(N, K, dtype=torch.bfloat16, device="cuda").normal_(0, 0.02)
if kind == "fp8":
    q, sc = ops.tile_weights_fp8(w, interleave_halves=True)
    fn = lambda: ops.linear_tiled_w8(x, q, sc, N, out=out, tiles_per_wg=tiles, epilogue=ops.EPI_SILU_MUL)
else:
    wt = ops.tile_weights(w, interleave_halves=True)
    fn = lambda: ops.linear_tiled(x, wt, N, out=out, tiles_per_wg=tiles, epilogue=ops.EPI_SILU_MUL)
for _ in range(6):
    fn()
torch.cuda.synchronize()
