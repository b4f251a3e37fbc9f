"""EXPERIMENT: time of the 32 x 128-token prefill step after the GPU has been idle for a while (the served path
prefills a burst ~10 ms after the last kernel of the previous run), and with a keep-busy kernel loop during the wait."""
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
a = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
big = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
big_o = torch.empty_like(big)
def ttft(idle_ms, busy, n=6):
    out = []
    for rnd in range(n):
        prompts = make_prompts(cfg.vocab_size, B, P, 1234 + rnd)
        torch.cuda.synchronize()
        t_end = time.perf_counter() + idle_ms * 1e-3
        while time.perf_counter() < t_end:
            if busy == "heavy":                                   # ~100 us of matrix work per launch, at most two in flight
                ev = torch.cuda.Event(); torch.mm(big, big, out=big_o); torch.mm(big, big, out=big_o); ev.record(); ev.synchronize()
            elif busy == "cpu":
                pass                                               # spin on the host only
            elif busy:
                torch.mm(a, a); torch.cuda.synchronize()
            else:
                time.sleep(0.0005)
        t0 = time.perf_counter()
        for i, p in enumerate(prompts):
            engine.add_request(f"r{time.time()}-{i}", p, params)
        seqs = list(engine.seqs.values())
        while any(not s.output_ids for s in seqs):
            engine.step()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) * 1e3)
        for rid in list(engine.seqs):
            engine.abort_request(rid)
    return round(statistics.median(out[1:]), 2)
for idle in (0, 10, 20):
    print(json.dumps({"idle_ms": idle, "ttft_ms_idle": ttft(idle, False), "ttft_ms_light_gemm_loop": ttft(idle, True),
                      "ttft_ms_heavy_gemm_loop": ttft(idle, "heavy"), "ttft_ms_host_spin_only": ttft(idle, "cpu")}), flush=True)
