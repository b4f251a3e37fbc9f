"""Host-side profile of the decode step loop (cProfile), bench-like workload."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, "vllm-backend-public_amd"); sys.path.insert(0, ".")
import torch
from mi355_decode.config import EngineArgs, ModelConfig
from mi355_decode.llm import build_engine
from mi355_decode.sampling_params import SamplingParams
import bench
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = ModelConfig.from_pretrained("llama-3-8b")
eng = build_engine(EngineArgs(model="llama-3-8b", dtype="bfloat16", max_num_seqs=32, max_model_len=256, load_format="dummy",
                              num_layers_override=layers), model_config=cfg)
for i, p in enumerate(bench.make_prompts(cfg.vocab_size, 32, 128, 1)):
    eng.add_request(str(i), p, SamplingParams(max_tokens=128, temperature=0.0, ignore_eos=True))
eng.step()
for _ in range(10): eng.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(100): eng.step()
pr.disable()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) * 10)
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
