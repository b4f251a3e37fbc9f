"""EXPERIMENT: TTFT of the 32 x 128-token burst with ONE of the four prefill GEMM shapes on the algorithm PyTorch
TunableOp picks for it and the other three on the library's defaults -- each shape in a fresh process (argv[1]:
qkv / o / gate_up / down / none), since TunableOp's table cannot forget an entry."""
import json, os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vllm-backend-public_amd")); sys.path.insert(0, ROOT)
import torch
from mi355_decode.config import EngineArgs, ModelConfig
from mi355_decode.engine import SamplingParams
from mi355_decode.llm import build_engine
from bench import make_prompts
which = sys.argv[1]
cfg = ModelConfig.from_pretrained("llama-3-8b")
B, P, G = 32, 128, 128
shapes = {"qkv": (6144, 4096), "o": (4096, 4096), "gate_up": (28672, 4096), "down": (4096, 14336)}
picked = None
if which != "none":
    import torch.cuda.tunable as tn
    n, k = shapes[which]
    w = torch.empty(n, k, dtype=torch.bfloat16, device="cuda").normal_(0, 0.02)
    x = torch.randn(B * P, k, dtype=torch.bfloat16, device="cuda")
    out = torch.empty(B * P, n, dtype=torch.bfloat16, device="cuda")
    tn.enable(True); tn.tuning_enable(True); tn.set_max_tuning_duration(30); tn.set_max_tuning_iterations(20)
    tn.set_filename("/tmp/tunable_%s.csv" % which)
    torch.mm(x, w.t(), out=out); torch.cuda.synchronize()
    tn.tuning_enable(False)
    picked = [r[2] for r in tn.get_results()]
    del w, x, out
args = EngineArgs(model="llama-3-8b", dtype="bfloat16", max_num_seqs=B, max_model_len=P + G, max_num_batched_tokens=8192,
                  tensor_parallel_size=1, load_format="dummy", gpu_memory_utilization=0.5, seed=0)
engine = build_engine(args, model_config=cfg)
params = SamplingParams(max_tokens=G, temperature=0.0, ignore_eos=True)
out = []
for rnd in range(9):
    prompts = make_prompts(cfg.vocab_size, B, P, 1234 + rnd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i, p in enumerate(prompts):
        engine.add_request(f"r{rnd}-{i}", p, params)
    seqs = list(engine.seqs.values())
    while any(not s.output_ids for s in seqs):
        engine.step()
    torch.cuda.synchronize(); out.append((time.perf_counter() - t0) * 1e3)
    for rid in list(engine.seqs):
        engine.abort_request(rid)
print(json.dumps({"tuned_shape": which, "picked": picked, "ttft_ms_median": round(statistics.median(out[2:]), 2),
                  "ttft_ms_min": round(min(out[2:]), 2)}), flush=True)
