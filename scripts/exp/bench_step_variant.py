"""EXPERIMENT: the headline decode step (Llama-3-8B, batch 32, contexts ~158) timed with whichever library
MI355_DECODE_LIB names: ms per step from graph replays of the runner's captured step."""
import sys
import numpy as np
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import weights as W
from mi355_decode.config import ModelConfig
from mi355_decode.model_runner import ModelRunner

cfg = ModelConfig.from_pretrained("llama-3-8b")
B, ctx, max_len = 32, 158, 320
bps = (max_len + 15) // 16
ws = W.synthetic(cfg, torch.bfloat16, "cuda", seed=1)
r = ModelRunner(cfg, ws, torch.bfloat16, "cuda", num_blocks=B * bps, max_num_seqs=B, max_model_len=max_len, tiled_linears=True)
n64, n32 = r.host_inputs()
n64[:B] = np.arange(B) + 1000
bt = n32[B:].reshape(B, r.max_blocks)
for b in range(B):
    bt[b, :bps] = np.arange(b * bps, (b + 1) * bps)
    p = ctx - 1
    n64[B + b], n64[2 * B + b], n32[b] = p, bt[b, p // 16] * 16 + p % 16, p + 1
r.run_decode(B); r.run_decode(B)
torch.cuda.synchronize()
g = r.graphs[r.bucket_for(B)]
best = 1e9
for _ in range(5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        g.replay()
    e.record(); e.synchronize()
    best = min(best, s.elapsed_time(e) / 20)
print(f"step {best:.4f} ms")
