"""Whole-step time of the headline decode step (Llama-3-8B, batch 32, contexts ~158) for each
(tiles_per_wg, k_groups) candidate of each projection, the others held at the current best:
the in-step time a candidate is worth, which the tuner's isolated GEMM + consumer timing
approximates.  One coordinate-descent pass; prints every candidate."""
import json, sys
import numpy as np
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode import ops, weights as W
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
    try:
        r.run_decode(B); r.run_decode(B)
    except Exception as exc:                      # no kernel of this form
        return None
    torch.cuda.synchronize()
    g = r.graphs[bucket]
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            g.replay()
        e.record(); e.synchronize()
        best = min(best, s.elapsed_time(e) / 10)
    return best


cur = dict(r._tiled_plan(bucket))
base = step_ms(cur)
print(json.dumps({"start": {k: list(v) for k, v in cur.items()}, "step_ms": round(base, 4)}), flush=True)
import os
kinds = os.environ.get("SWEEP_KINDS", "down,o,qkv,gu,lm_head").split(",")
max_qkv = int(os.environ.get("SWEEP_QKV_MAX_SLABS", r.QKV_MAX_SLABS))
for kind in kinds:
    n, kk, split = r.tiled_shapes[kind]
    rows = []
    for cand in ops.linear_tiled_candidates(n, kk, split, B, swiglu=(kind == "gu")):
        if kind == "qkv" and cand[1] > max_qkv:
            continue
        t = step_ms({**cur, kind: cand})
        rows.append((list(cand), None if t is None else round(t, 4)))
    ok = [x for x in rows if x[1] is not None]
    bestc = min(ok, key=lambda x: x[1])
    if bestc[1] < base * 0.997:
        cur[kind], base = tuple(bestc[0]), bestc[1]
    print(json.dumps({"projection": kind, "candidates": rows, "kept": list(cur[kind]), "step_ms": round(base, 4)}), flush=True)
print(json.dumps({"final": {k: list(v) for k, v in cur.items()}, "step_ms": round(base, 4)}), flush=True)
