#!/usr/bin/env python3
"""Engine decode step against a bare replay of the same HIP graph (Llama-3-8B, batch 32): what the
host, the input copies and the token read-back add on top of the GPU work.  Measured: 3.556 vs
3.539 ms -- the host is off the critical path."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-backend-public_amd"))
import torch, numpy as np
from mi355_decode.config import EngineArgs, ModelConfig
from mi355_decode.llm import build_engine
from mi355_decode.sampling_params import SamplingParams
cfg = ModelConfig.from_pretrained("llama-3-8b")
args = EngineArgs(model="llama-3-8b", dtype="bfloat16", max_num_seqs=32, max_model_len=256, max_num_batched_tokens=8192,
                  load_format="dummy", gpu_memory_utilization=0.9, seed=0)
eng = build_engine(args, model_config=cfg)
g = torch.Generator().manual_seed(1)
prompts = torch.randint(1000, 100000, (32, 128), generator=g).tolist()
for i, p in enumerate(prompts):
    eng.add_request(str(i), p, SamplingParams(max_tokens=127, temperature=0.0, ignore_eos=True))
for _ in range(20):
    eng.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(60):
    eng.step()
torch.cuda.synchronize()
step_ms = (time.perf_counter() - t0) / 60 * 1e3
r = eng.runner
gr = r.graphs[32]
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(60):
    gr.replay()
e.record(); e.synchronize()
print(f"engine step {step_ms:.4f} ms; bare graph replay {s.elapsed_time(e)/60:.4f} ms")
