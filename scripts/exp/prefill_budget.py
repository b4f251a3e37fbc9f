"""EXPERIMENT: a burst of 32 x 128-token prompts handed to the engine at once, prefilled under different token
budgets per step (max_num_batched_tokens): when each request gets its first token (p50 / max over the 32) and
when the whole burst has been prefilled."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vllm-backend-public_amd"))
import statistics
import torch
from mi355_decode.config import EngineArgs, ModelConfig
from mi355_decode.engine import SamplingParams
from mi355_decode.llm import build_engine
sys.path.insert(0, ROOT)
from bench import make_prompts

cfg = ModelConfig.from_pretrained("llama-3-8b")
B, P, G = 32, 128, 128
for budget in (8192, 2048, 1024, 512, 8192, 2048, 1024, 512):
    args = EngineArgs(model="llama-3-8b", dtype="bfloat16", max_num_seqs=B, max_model_len=P + G, max_num_batched_tokens=budget,
                      tensor_parallel_size=1, load_format="dummy", gpu_memory_utilization=0.5, seed=0)
    engine = build_engine(args, model_config=cfg)
    params = SamplingParams(max_tokens=G, temperature=0.0, ignore_eos=True)
    res = []
    for rnd in range(4):
        prompts = make_prompts(cfg.vocab_size, B, P, 1234 + rnd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i, p in enumerate(prompts):
            engine.add_request(f"r{rnd}-{i}", p, params, arrival_time=t0)
        seqs = list(engine.seqs.values())
        steps = 0
        while any(not s.output_ids for s in seqs):
            engine.step(); steps += 1
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        tt = [(s.first_token_time - t0) * 1e3 for s in seqs]
        if rnd:
            res.append((statistics.median(tt), max(tt), (t1 - t0) * 1e3, steps))
        for rid in list(engine.seqs):
            engine.abort_request(rid)
    res.sort()
    m = res[len(res) // 2]
    print(json.dumps({"budget_tokens": budget, "prefill_steps": m[3], "ttft_ms_p50_over_requests": round(m[0], 2),
                      "ttft_ms_max": round(m[1], 2), "burst_prefilled_ms": round(m[2], 2)}), flush=True)
    del engine
    torch.cuda.empty_cache()
