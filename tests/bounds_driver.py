"""Run by tests/test_bounds_gpu.py in a process of its own, with the bounds-checking build of the
kernel library loaded (MI355_DECODE_LIB): (1) a deliberately wrong call is caught and named
without faulting the GPU; (2) the decode step of every benchmark configuration's rank shape, as
a HIP graph and eagerly, on the tiled and on the library GEMM path, and a prefill + sampled
generation through the engine refuse no access.  Prints one JSON line."""
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vllm-backend-public_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from mi355_decode import bounds  # noqa: E402
from mi355_decode._lib import DTYPE_BF16, lib  # noqa: E402


def negative_control():
    """rms_norm told that x has 6 rows when the tensor holds 4: rows 4 and 5 leave the allocation."""
    H = 4096
    x = torch.randn(4, H, device="cuda").bfloat16()
    out = torch.empty(6, H, device="cuda", dtype=torch.bfloat16)
    w = torch.ones(H, device="cuda", dtype=torch.bfloat16)
    bounds.sync_allocations()
    rc = lib.mi355_rms_norm(out.data_ptr(), x.data_ptr(), w.data_ptr(), 6, H, H, H, 1e-5, DTYPE_BF16, None)
    assert rc == 0, rc
    rep = bounds.report()
    assert rep is not None and rep["file"].endswith("rms_norm.hip") and rep["base"] == x.data_ptr(), rep
    assert rep["address"] >= x.data_ptr() + x.numel() * 2, rep
    # 2 rows x 512 vectors, each loaded once by the generic kernel's cached form
    assert rep["refused"] == 2 * (H // 8), rep
    assert bounds.report() is None                      # the record was cleared
    # the rows inside the tensor were computed as usual
    ref = x.float() * torch.rsqrt((x.float() ** 2).mean(-1, keepdim=True) + 1e-5)
    assert torch.allclose(out[:4].float(), ref.bfloat16().float(), atol=2e-2)
    return rep


def rank_steps():
    from mi355_decode import weights as W
    from mi355_decode.config import ModelConfig
    from mi355_decode.model_runner import ModelRunner
    from tests.test_configs_gpu import CASES
    done = []
    for name, model, tp, rank, batch, ctx, max_len, shard_mlp, fp8 in CASES:
        cfg = ModelConfig.from_pretrained(model)
        blocks_per_seq = (max_len + 15) // 16
        num_blocks = batch * blocks_per_seq
        rng = np.random.Generator(np.random.PCG64(9))
        ids = rng.integers(0, cfg.vocab_size, batch)
        pos = np.asarray([ctx - 1 - (b % 7) for b in range(batch)], dtype=np.int64)
        tables = np.arange(num_blocks, dtype=np.int32).reshape(batch, blocks_per_seq)
        slots = tables[np.arange(batch), pos // 16].astype(np.int64) * 16 + pos % 16
        for tiled in (True, False):
            t0 = time.time()
            ws = W.synthetic(cfg, torch.bfloat16, "cuda", seed=5, tp_rank=rank, tp_size=tp, num_layers=2, std=0.03,
                             shard_mlp=shard_mlp, shard_vocab=shard_mlp)
            bounds.clear()                                  # weight tiling runs unchecked: allocations are in flux
            r = ModelRunner(cfg, ws, torch.bfloat16, "cuda", num_blocks=num_blocks, max_num_seqs=batch,
                            max_model_len=max_len, tp_rank=rank, tp_size=tp, tp_group=None, tiled_linears=tiled,
                            weight_fp8=fp8)
            r._allreduce = lambda t: None
            r._allreduce_max = lambda t: None
            n64, n32 = r.host_inputs()
            B = r.max_num_seqs
            n64[:B], n64[B:2 * B], n64[2 * B:] = ids, pos, slots
            n32[:B] = pos + 1
            n32[B:].reshape(B, r.max_blocks)[:, :blocks_per_seq] = tables
            r.run_decode(B)                                 # tunes, captures, allocates
            assert r.use_graphs and r.bucket_for(B) in r.graphs
            n = bounds.sync_allocations()
            r.run_decode(B)                                 # graph replay, every access checked
            rep = bounds.report()
            assert rep is None, (name, tiled, "graph", rep)
            r.use_graphs = False
            r.run_decode(B)                                 # the same step launched eagerly
            rep = bounds.report()
            assert rep is None, (name, tiled, "eager", rep)
            done.append(f"{name}/{'tiled' if tiled else 'library'}:{n} ranges:{time.time() - t0:.1f}s")
            del r, ws
            gc.collect()
            torch.cuda.empty_cache()
    return done


def engine_generation():
    """Prefill (several chunk shapes), decode, greedy and sampled rows, log-probabilities."""
    from mi355_decode.config import EngineArgs, ModelConfig
    from mi355_decode.llm import build_engine
    from mi355_decode.sampling_params import SamplingParams
    from oracle.tiny_models import make_prompts, make_weights
    out = []
    for name in ("llama_d128", "qwen2_d64", "opt_d64"):
        arch, cfg, w = make_weights(name)
        bounds.clear()
        eng = build_engine(EngineArgs(model=name, dtype="bfloat16", max_num_seqs=8, max_model_len=256,
                                      gpu_memory_utilization=0.0),
                           state_dict={k: torch.from_numpy(v) for k, v in w.items()},
                           model_config=ModelConfig.from_dict({**cfg, "architectures": [arch]}))

        def run(tag):
            prompts = make_prompts(name, 6, [5, 16, 17, 33, 64, 100])
            for i, p in enumerate(prompts):
                sp = SamplingParams(max_tokens=12, temperature=0.0 if i % 2 == 0 else 0.8, top_k=-1 if i < 3 else 40,
                                    top_p=1.0 if i < 4 else 0.9, seed=7 + i, ignore_eos=True,
                                    logprobs=1 if i == 1 else None)
                eng.add_request(f"{tag}{i}", [int(t) for t in p], sp)
            n = 0
            while eng.has_unfinished():
                n += sum(len(o.new_token_ids) for o in eng.step())
            return n

        run("warm")
        n = bounds.sync_allocations(granularity="segment")    # the engine allocates per-step temporaries
        toks = run("checked")
        # tensors from segments the allocator opened during the run are not in the list: counted, not refused
        rep = bounds.report()
        assert rep is None or rep["refused"] == 0, (name, rep)
        out.append(f"{name}:{toks} tokens:{n} segments:{rep['unknown_base'] if rep else 0} accesses with an unlisted base")
        del eng
        gc.collect()
        torch.cuda.empty_cache()
    return out


def sampler_rows():
    """The sampling launch on exactly sized logits (no slack behind the last row), every kind of
    row in one call: greedy, whole-vocabulary draw, top-k / top-p cuts, an all-tied row (the list
    overflows: radix select), a mostly-NaN row; vocabularies with and without a vector tail."""
    from mi355_decode import ops
    done = []
    g = torch.Generator(device="cpu").manual_seed(3)
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        for B, V in ((9, 128256), (5, 1000), (33, 50272), (4, 72)):
            lg = (torch.randn(B, V, generator=g) * 3).to(dt)
            lg[1] = 0.0
            lg[2, 5:] = float("nan")
            lgd = lg.cuda()
            temp = torch.tensor([0.0, 0.7, 1.0, 1.3] * 9)[:B].cuda()
            top_k = torch.tensor([-1, 40, 1024, -1, 7] * 8, dtype=torch.int32)[:B].cuda()
            top_p = torch.tensor([1.0, 0.9, 1.0, 0.5, 1.0, 1.0, 0.95] * 6)[:B].cuda()
            seeds = torch.arange(B, dtype=torch.int64).cuda()
            steps = torch.arange(B, dtype=torch.int64).cuda() * 3
            out = torch.empty(B, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            bounds.sync_allocations()
            ops.sample(lgd, temp, top_k, top_p, seeds, steps, out=out)
            torch.cuda.synchronize()
            rep = bounds.report()
            assert rep is None, (str(dt), B, V, rep)
            assert int(out.min()) >= 0 and int(out.max()) < V
            done.append(f"{str(dt).split('.')[-1]}:{B}x{V}")
            bounds.clear()
    return done


if __name__ == "__main__":
    assert bounds.enabled(), "not the bounds-checking build: set MI355_DECODE_LIB"
    res = {"negative_control": negative_control()}
    res["sampler"] = sampler_rows()
    res["rank_steps"] = rank_steps()
    res["engine"] = engine_generation()
    print(json.dumps(res))
