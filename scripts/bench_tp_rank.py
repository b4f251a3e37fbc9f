#!/usr/bin/env python3
"""One GPU: the decode step of ONE tensor-parallel rank at the shapes of N-GPU runs (collectives
replaced by no-ops), MLP + lm_head sharded vs replicated, tiled vs library GEMMs vs the timing-based choice.  It prices the compute
side of a rank's step; the all-reduces come on top on a real node.

    python scripts/bench_tp_rank.py [--model llama-3-8b] [--tps 2,4,8] [--batch-per-gpu 32]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-backend-public_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mi355_decode import weights as W  # noqa: E402
from mi355_decode.config import ModelConfig  # noqa: E402
from mi355_decode.model_runner import ModelRunner  # noqa: E402


def step_ms(cfg, tp, batch, ctx, shard_mlp, tiled, layers, oneshot=False):
    max_len = ctx + 64
    bps = (max_len + 15) // 16
    ws = W.synthetic(cfg, torch.bfloat16, "cuda", seed=1, tp_rank=tp - 1, tp_size=tp, num_layers=layers,
                     shard_mlp=shard_mlp, shard_vocab=shard_mlp)
    r = ModelRunner(cfg, ws, torch.bfloat16, "cuda", num_blocks=batch * bps, max_num_seqs=batch, max_model_len=max_len,
                    tp_rank=tp - 1, tp_size=tp, tp_group=None, tiled_linears=tiled)
    if oneshot:
        # the rank's own share of the one-shot all-reduce: a communicator of ONE rank (the kernel delivers the
        # message into its own buffer, waits for its own flag and "sums" one slot) -- launch, slab sum and local
        # traffic are real, the peers' traffic and arrival skew are not
        from mi355_decode.collectives import OneShotAllReduce
        r.oneshot = OneShotAllReduce(rank=0, world=1, max_bytes=max(1 << 20, batch * cfg.hidden_size * 2))
    else:
        r._allreduce = lambda t: None
    r._allreduce_max = lambda t: None
    n64, n32 = r.host_inputs()
    B = batch
    n64[:B] = np.arange(B) + 1000
    bt = n32[B:].reshape(B, r.max_blocks)
    for b in range(B):
        bt[b, :bps] = np.arange(b * bps, (b + 1) * bps)
        p = ctx - 1
        n64[B + b], n64[2 * B + b], n32[b] = p, bt[b, p // 16] * 16 + p % 16, p + 1
    r.run_decode(B)
    r.run_decode(B)
    torch.cuda.synchronize()
    g = r.graphs[r.bucket_for(B)]
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        g.replay()
    e.record()
    e.synchronize()
    ms = s.elapsed_time(e) / 20
    used = "".join(k[0] if v else "-" for k, v in r._mode(B).items())      # q o g d l: projections on tiled kernels
    tuned = {k: v for (_, k), v in getattr(r, "tune_us", {}).items()}
    plans = {k: list(v) for k, v in r._tiled_plan(r.bucket_for(B)).items()} if r.tiled is not None else {}
    del r, ws
    torch.cuda.empty_cache()
    return ms, used, tuned, plans


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--tps", default="2,4,8")
    ap.add_argument("--batch-per-gpu", type=int, default=32)
    ap.add_argument("--ctx", type=int, default=192)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--oneshot", action="store_true",
                    help="keep the all-reduce launches: a one-rank communicator of the one-shot kernel (slab sums formed "
                         "inside it) instead of no-ops")
    ap.add_argument("--only", default=None,
                    help="one variant instead of all six, e.g. sharded-timed, replicated-library (for a profiler run)")
    a = ap.parse_args()
    cfg = ModelConfig.from_pretrained(a.model)
    L = a.layers or cfg.num_hidden_layers
    for tp in (int(t) for t in a.tps.split(",")):
        batch = a.batch_per_gpu * tp
        for shard in (False, True):
            for tiled in (True, False, None):
                name = f"{'sharded' if shard else 'replicated'}-{ {True: 'tiled', False: 'library', None: 'timed'}[tiled]}"
                if a.only and a.only != name:
                    continue
                ms, used, tuned, plans = step_ms(cfg, tp, batch, a.ctx, shard, tiled, L, a.oneshot)
                if tiled is True and used == "-----":
                    continue                              # no tiled kernels for this row count: same as library
                print(json.dumps({"model": a.model, "tp": tp, "batch": batch, "layers": L, "mlp_and_vocab_sharded": shard,
                                  "allreduce": "one-rank one-shot kernel" if a.oneshot else "no-op",
                                  "gemms": {True: "tiled", False: "library", None: "timed choice"}[tiled],
                                  "tiled_projections": used, "rank_step_ms": round(ms, 4),
                                  "tok_s_without_collectives": round(batch / ms * 1e3, 1),
                                  **({"tiled_plans": plans} if used != "-----" else {}),
                                  **({"tuned_us_tiled_vs_library": tuned} if tuned else {})}), flush=True)


if __name__ == "__main__":
    main()
