"""Whole-step time of the headline decode step (batch 32, context 158) for given GEMM plans:
python3 scripts/exp/step_fixed_plans.py  (plans: qkv 8/4, o 104/4, gu 8/1, lm_head 8/1; down 104/8 and 104/4)"""
import json, sys
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
bucket = r.bucket_for(B)
def step_ms(plans):
    r._tuned_plans[bucket] = dict(plans)
    r._modes[bucket] = {k: True for k in r.KINDS}
    r.graphs.pop(bucket, None)
    r.run_decode(B); r.run_decode(B)
    torch.cuda.synchronize()
    g = r.graphs[bucket]
    best = 1e9
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            g.replay()
        e.record(); e.synchronize()
        best = min(best, s.elapsed_time(e) / 10)
    return round(best, 4)
base = {"qkv": (8, 4), "o": (104, 4), "gu": (8, 1), "lm_head": (8, 1)}
print(json.dumps({"down_104_8": step_ms({**base, "down": (104, 8)}), "down_104_4": step_ms({**base, "down": (104, 4)}),
                  "down_104_8_again": step_ms({**base, "down": (104, 8)})}))
