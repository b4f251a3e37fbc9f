"""EXPERIMENT: TTFT of the 32 x 128-token burst with the library's default GEMM algorithms, then -- same
process, same engine -- after ModelRunner.tune_prefill_gemms picked algorithms by timing."""
import json, os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vllm-backend-public_amd")); sys.path.insert(0, ROOT)
import torch
from mi355_decode.config import EngineArgs, ModelConfig
from mi355_decode.engine import SamplingParams
from mi355_decode.llm import build_engine
from bench import make_prompts
cfg = ModelConfig.from_pretrained("llama-3-8b")
B, P, G = 32, 128, 128
args = EngineArgs(model="llama-3-8b", dtype="bfloat16", max_num_seqs=B, max_model_len=P + G, max_num_batched_tokens=8192,
                  tensor_parallel_size=1, load_format="dummy", gpu_memory_utilization=0.5, seed=0)
engine = build_engine(args, model_config=cfg)
params = SamplingParams(max_tokens=G, temperature=0.0, ignore_eos=True)
def ttft(n=7):
    out = []
    for rnd in range(n):
        prompts = make_prompts(cfg.vocab_size, B, P, 1234 + rnd)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i, p in enumerate(prompts):
            engine.add_request(f"r{time.time()}-{i}", p, params)
        seqs = list(engine.seqs.values())
        while any(not s.output_ids for s in seqs):
            engine.step()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) * 1e3)
        for rid in list(engine.seqs):
            engine.abort_request(rid)
    return round(statistics.median(out[1:]), 2), [round(x, 2) for x in out]
print(json.dumps({"default": ttft()}), flush=True)
t0 = time.time()
for ms, it in ((30, 20), (200, 100)):
    tuned = engine.runner.tune_prefill_gemms((B * P,), max_ms=ms, max_iters=it)
    print(json.dumps({"tuning_s": round(time.time() - t0, 1), "max_ms": ms, "tuned": tuned}), flush=True)
    print(json.dumps({"tuned": ttft()}), flush=True)
    import torch.cuda.tunable as tn
    tn.enable(False)
    print(json.dumps({"tunableop_off_again": ttft()}), flush=True)
    break
