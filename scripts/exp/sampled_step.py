"""Engine decode step (Llama-3-8B geometry, batch 32, synthetic weights) with greedy rows, with
whole-vocabulary random sampling (temperature only) and with top-p / top-k cuts: what random
sampling adds to a step end to end (kernel + parameter upload + the id hand-over)."""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vllm-backend-public_amd"))
import torch
from mi355_decode.config import EngineArgs, ModelConfig
from mi355_decode.llm import build_engine
from mi355_decode.sampling_params import SamplingParams

cfg = ModelConfig.from_pretrained("llama-3-8b")
args = EngineArgs(model="llama-3-8b", dtype="bfloat16", max_num_seqs=32, max_model_len=256, max_num_batched_tokens=8192,
                  load_format="dummy", gpu_memory_utilization=0.9, seed=0)
eng = build_engine(args, model_config=cfg)
g = torch.Generator().manual_seed(1)
rec = {}
for name, kw in (("greedy", dict(temperature=0.0)), ("temperature_only", dict(temperature=0.8, seed=7)),
                 ("top_p_0.9", dict(temperature=0.8, top_p=0.9, seed=7)),
                 ("top_k_50_top_p_0.9", dict(temperature=0.8, top_k=50, top_p=0.9, seed=7)), ("greedy_again", dict(temperature=0.0))):
    prompts = torch.randint(1000, 100000, (32, 128), generator=g).tolist()
    for i, p in enumerate(prompts):
        eng.add_request(f"{name}-{i}", p, SamplingParams(max_tokens=127, ignore_eos=True, **kw))
    for _ in range(20):
        eng.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(80):
        eng.step()
    torch.cuda.synchronize()
    rec[name + "_ms_per_step"] = round((time.perf_counter() - t0) / 80 * 1e3, 4)
    while eng.has_unfinished():
        eng.step()
print(json.dumps(rec))
