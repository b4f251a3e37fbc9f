"""Host side of one prefill step of the headline workload (32 x 128 tokens): cProfile of
add_request x 32 + engine.step() up to the point the step's kernels are enqueued and the first
tokens are read back."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vllm-backend-public_amd"))
import torch
from mi355_decode.config import EngineArgs, ModelConfig
from mi355_decode.llm import build_engine
from mi355_decode.sampling_params import SamplingParams

cfg = ModelConfig.from_pretrained("llama-3-8b")
B, P, G = 32, 128, 128
args = EngineArgs(model="llama-3-8b", dtype="bfloat16", max_num_seqs=B, max_model_len=P + G, max_num_batched_tokens=8192,
                  load_format="dummy", gpu_memory_utilization=0.9, seed=0)
eng = build_engine(args, model_config=cfg)
params = SamplingParams(max_tokens=G, temperature=0.0, ignore_eos=True)
g = torch.Generator().manual_seed(1)
def batch(tag):
    prompts = torch.randint(1000, 100000, (B, P), generator=g).tolist()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i, p in enumerate(prompts):
        eng.add_request(f"{tag}-{i}", p, params)
    t1 = time.perf_counter()
    while any(not s.output_ids for s in list(eng.sched.waiting) + eng.sched.running):
        eng.step()
    t2 = time.perf_counter()
    for rid in list(eng.seqs):
        eng.abort_request(rid)
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3
for k in range(3):
    print("add_request x32 %.3f ms, prefill step %.3f ms" % batch(f"w{k}"))
pr = cProfile.Profile()
pr.enable()
r = batch("p")
pr.disable()
print("profiled: add_request x32 %.3f ms, prefill step %.3f ms" % r)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:5000])
