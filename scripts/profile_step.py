"""Where the time of one decode step goes: bare graph replays back to back, replays with the
per-step copies around them, and the engine's own step loop (Llama-3-8B geometry, batch 32)."""
import sys
import time
import torch
sys.path.insert(0, "vllm-backend-public_amd")
from mi355_decode.config import EngineArgs, ModelConfig
from mi355_decode.llm import build_engine
from mi355_decode.sampling_params import SamplingParams

B, P, G = 32, 128, 128
cfg = ModelConfig.from_pretrained("llama-3-8b")
eng = build_engine(EngineArgs(model="llama-3-8b", dtype="bfloat16", max_num_seqs=B, max_model_len=P + G,
                              max_num_batched_tokens=8192, load_format="dummy", gpu_memory_utilization=0.9, seed=0),
                   model_config=cfg)
g = torch.Generator().manual_seed(1)
for i in range(B):
    eng.add_request(str(i), torch.randint(1000, 100000, (P,), generator=g).tolist(),
                    SamplingParams(max_tokens=G, temperature=0.0, ignore_eos=True))
for _ in range(40):
    eng.step()
r = eng.runner
torch.cuda.synchronize()
graph = r.graphs[r.bucket_for(B)]
N = 100


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3


print(f"bare graph replay            {wall(graph.replay):.4f} ms/step")


def with_copies():
    r.in64.copy_(r.h64s[0], non_blocking=True)
    r.in32.copy_(r.h32s[0], non_blocking=True)
    graph.replay()
    r.h_nexts[0][:B].copy_(r.next_ids[:B], non_blocking=True)
    r.done[0].record()


print(f"replay + 2 H2D + D2H + event {wall(with_copies):.4f} ms/step")


def with_copies_ahead():
    r.in64[B:].copy_(r.h64s[0][B:], non_blocking=True)
    r.ids[:B].copy_(r.next_ids[:B], non_blocking=True)
    r.in32.copy_(r.h32s[0], non_blocking=True)
    graph.replay()
    r.h_nexts[0][:B].copy_(r.next_ids[:B], non_blocking=True)
    r.done[0].record()


print(f"... + d2d ids (run-ahead)    {wall(with_copies_ahead):.4f} ms/step")
t = wall(lambda: eng.step())
print(f"engine.step loop             {t:.4f} ms/step  ({eng.stats})")
